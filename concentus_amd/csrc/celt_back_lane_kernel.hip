// celt_back_lane_kernel.hip -- back phase (FrameMid -> packet) with ONE LANE per frame.
//
// The back phase is a serial chain of ~160 small steps per frame; with one
// wavefront per frame most of its vector instructions carry 1-16 useful lanes. This build compiles the
// same sources with LANES == 1 (wave.h, CA_LANE_FRAME): 64 frames share a wavefront, every instruction
// does work for all of them, the per-frame working set lives in private memory.
#define CA_LANE_FRAME 1
#include <stdlib.h>
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
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

__global__ __launch_bounds__(64) void celt_back_lane_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                            FrameMid *mid, u8 *out, int out_stride,
                                                            int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes, int lanes_per_wave)
{
    // small batches: fewer frames per wavefront so that several wavefronts share a SIMD and hide each
    // other's memory latency (the kernel is latency-bound long before it is issue-bound)
    fill_lds_tables();
    const int n = blockIdx.x * lanes_per_wave + threadIdx.x;
    if ((int)threadIdx.x >= lanes_per_wave || n >= nframes) return;
    BackLds F;
    opusgpu_celt_state *st = states ? states + n : nullptr;
    FrameResult r = celt_encode_back(F, cfg, mid + n, st, out + (size_t)n * out_stride);
    out_len[n] = r.bytes;
    out_rng[n] = r.final_range;
}

}  // namespace ca

extern "C" void opusgpu_launch_back_lane(const opusgpu_celt_config *cfg, void *states, const void *mid, unsigned char *out,
                                         int out_stride, int32_t *out_len, uint32_t *out_rng, int n, hipStream_t s)
{
    static const int env_lpw = getenv("OPUSGPU_LPW") ? atoi(getenv("OPUSGPU_LPW")) : 0;
    int lpw = env_lpw > 0 ? env_lpw : 64;
    hipLaunchKernelGGL(ca::celt_back_lane_kernel, dim3((n + lpw - 1) / lpw), dim3(64), 0, s, *cfg, (opusgpu_celt_state *)states,
                       (ca::FrameMid *)mid, out, out_stride, out_len, out_rng, n, lpw);
}
