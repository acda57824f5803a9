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
    // derived per (LM + 1, band), built by fill_lds_tables: the length of the pulse-cache row (cache[0]) and its last entry
    // (cache[cache[0]], rate.h's "maximum bits of an unsplit band") -- one look-up instead of two / three dependent ones
    uint8_t cache_len_[105];
    uint8_t cache_max_[105];
};
#if defined(CA_HOST_EMU)
#define CA_SHARED static            // the lane build on a CPU (tests/emu/celt_lane_emu.cpp): the workgroup's LDS is a plain array
#else
#define CA_SHARED __shared__
#endif
CA_SHARED LdsTables g_lds_tables;
// Per-lane scratch of the 64 frames of the workgroup, [element][lane] (30 KB in one block), 16-bit slots throughout. Every lane
// owns ONE column, and the two wavefronts of a workgroup (OPUSGPU_LANE_FRAMES=32) run independently, so the block may only
// ever be addressed with ONE element size -- a 32-bit view of a region would land in other lanes' columns. The decoder's
// lane kernel (celt_dec.h) lays it out as
//   slots   0..175  band scratch: a leaf while it is scaled and un-rotated, a band during resynthesis / (de)interleave
//                   (LANE_SCRATCH_N: the widest band, 176 bins, fits since round 3 -- at 144 the last band of every
//                   transient frame was re-arranged in place in HBM, two bytes per access)       g_lds_pvq16
//   slots 176..239  the pulse vector iy of a leaf of up to LANE_IY16_N = 64 bins (16-bit: |iy| <= K < 2^15)   g_lds_iy16
// The encoder's back kernel (celt_back_lane_kernel.hip) defines CA_LANE_SLOTS 264 and lays the column out itself
// (celt_enc_front.h LS_*, celt_enc_lane.h M_*).
enum { LANE_SCRATCH_N = 176, LANE_IY16_N = 64 };
#if !defined(CA_LANE_SLOTS)
#define CA_LANE_SLOTS (LANE_SCRATCH_N + LANE_IY16_N)
#endif
CA_SHARED __attribute__((aligned(16))) int16_t g_lds_scratch[CA_LANE_SLOTS * 64];
#define g_lds_pvq16 (ca::g_lds_scratch)
#define g_lds_iy16 (ca::g_lds_scratch + ca::LANE_SCRATCH_N * 64)
#if defined(CA_HOST_EMU)
static inline void fill_lds_tables()
{
#define X(T, NAME, N) for (int k = 0; k < N; k++) g_lds_tables.NAME##_[k] = NAME[k];
    CA_LDS_TABLES(X)
#undef X
    for (int k = 0; k < 105; k++) {
        const int ix = CLT_cache_index50[k];
        g_lds_tables.cache_len_[k] = ix >= 0 ? CLT_cache_bits50[ix] : 0;
        g_lds_tables.cache_max_[k] = ix >= 0 ? CLT_cache_bits50[ix + CLT_cache_bits50[ix]] : 0;
    }
}
#else
__device__ __forceinline__ void fill_lds_tables()
{
#define X(T, NAME, N) for (int k = threadIdx.x; k < N; k += blockDim.x) g_lds_tables.NAME##_[k] = NAME[k];
    CA_LDS_TABLES(X)
#undef X
    for (int k = threadIdx.x; k < 105; k += blockDim.x) {
        const int ix = CLT_cache_index50[k];
        g_lds_tables.cache_len_[k] = ix >= 0 ? CLT_cache_bits50[ix] : 0;
        g_lds_tables.cache_max_[k] = ix >= 0 ? CLT_cache_bits50[ix + CLT_cache_bits50[ix]] : 0;
    }
    __syncthreads();
}
#endif
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
