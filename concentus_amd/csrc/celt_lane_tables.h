// celt_lane_tables.h -- LDS copies of the small CELT tables for the lane-per-frame builds.
// Include BEFORE celt_enc.h (and after defining CA_LANE_FRAME); call fill_lds_tables() first thing in the kernel.
#pragma once
#include "device_tables.h"

// The back phase looks its small tables up at per-lane addresses (pulse cache, PVQ U(n,k), band edges ...):
// from global memory every look-up is an L2 round trip on the critical path of a lone wavefront. The
// workgroup copies them into LDS once and the sources below see the LDS copies under the tables' names.
#define CA_LDS_TABLES(X) \
    X(uint32_t, CLT_tell_frac_correction, 8) \
    X(uint32_t, CLT_pvq_u_data, 1272) \
    X(int16_t, CLT_eband5ms, 22) \
    X(int16_t, CLT_pred_coef, 4) \
    X(int16_t, CLT_logN400, 21) \
    X(uint16_t, CLT_pvq_u_row, 15) \
    X(int16_t, CLT_intensity_thresholds, 21) \
    X(int16_t, CLT_intensity_histeresis, 21) \
    X(int16_t, CLT_exp2_table8, 8) \
    X(int16_t, CLT_cache_index50, 105) \
    X(int16_t, CLT_beta_coef, 4) \
    X(uint8_t, CLT_band_allocation, 231) \
    X(int8_t, CLT_tf_select_table, 32) \
    X(uint8_t, CLT_log2_frac_table, 24) \
    X(uint8_t, CLT_trim_icdf, 11) \
    X(uint8_t, CLT_spread_icdf, 4) \
    X(uint8_t, CLT_small_energy_icdf, 3) \
    X(uint8_t, CLT_ordery_table, 30) \
    X(uint8_t, CLT_e_prob_model, 336) \
    X(int8_t, CLT_eMeans, 25) \
    X(uint8_t, CLT_cache_caps50, 168) \
    X(uint8_t, CLT_cache_bits50, 392)
namespace ca {
struct LdsTables {
#define X(T, NAME, N) T NAME##_[N];
    CA_LDS_TABLES(X)
#undef X
};
__shared__ LdsTables g_lds_tables;
// PVQ search state of the 64 frames of the workgroup, [element][lane]: y and |x| (int16, 48 each), iy (int32, 48)
__shared__ int16_t g_lds_pvq16[2 * 48 * 64];
__shared__ int32_t g_lds_pvq32[48 * 64];
__shared__ int16_t g_lds_xs[48 * 64];
__device__ __forceinline__ void fill_lds_tables()
{
#define X(T, NAME, N) for (int k = threadIdx.x; k < N; k += blockDim.x) g_lds_tables.NAME##_[k] = NAME[k];
    CA_LDS_TABLES(X)
#undef X
    __syncthreads();
}
}  // namespace ca
#define CLT_tell_frac_correction g_lds_tables.CLT_tell_frac_correction_
#define CLT_pvq_u_data g_lds_tables.CLT_pvq_u_data_
#define CLT_eband5ms g_lds_tables.CLT_eband5ms_
#define CLT_pred_coef g_lds_tables.CLT_pred_coef_
#define CLT_logN400 g_lds_tables.CLT_logN400_
#define CLT_pvq_u_row g_lds_tables.CLT_pvq_u_row_
#define CLT_intensity_thresholds g_lds_tables.CLT_intensity_thresholds_
#define CLT_intensity_histeresis g_lds_tables.CLT_intensity_histeresis_
#define CLT_exp2_table8 g_lds_tables.CLT_exp2_table8_
#define CLT_cache_index50 g_lds_tables.CLT_cache_index50_
#define CLT_beta_coef g_lds_tables.CLT_beta_coef_
#define CLT_band_allocation g_lds_tables.CLT_band_allocation_
#define CLT_tf_select_table g_lds_tables.CLT_tf_select_table_
#define CLT_log2_frac_table g_lds_tables.CLT_log2_frac_table_
#define CLT_trim_icdf g_lds_tables.CLT_trim_icdf_
#define CLT_spread_icdf g_lds_tables.CLT_spread_icdf_
#define CLT_small_energy_icdf g_lds_tables.CLT_small_energy_icdf_
#define CLT_ordery_table g_lds_tables.CLT_ordery_table_
#define CLT_e_prob_model g_lds_tables.CLT_e_prob_model_
#define CLT_eMeans g_lds_tables.CLT_eMeans_
#define CLT_cache_caps50 g_lds_tables.CLT_cache_caps50_
#define CLT_cache_bits50 g_lds_tables.CLT_cache_bits50_
