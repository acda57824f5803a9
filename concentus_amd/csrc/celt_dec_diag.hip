// celt_dec_diag.hip -- DIAGNOSTIC build of the decoder's lane kernel with s_memtime stage stamps
// (-DCA_STAGE_TIMING). Never used for reported throughput; cycle totals per wavefront go to their own buffer.
#define CA_LANE_FRAME 1
#define CA_STAGE_TIMING 1
#include "celt_lane_tables.h"
#include "celt_dec.h"
#include "opusgpu_internal.h"

namespace ca {

enum { NSTAGES = 32 };

__global__ __launch_bounds__(64) void celt_decode_lane_diag_kernel(opusgpu_celt_dec_state *states, const u8 *__restrict__ packets,
                                                                   int packet_stride, const int *__restrict__ len,
                                                                   int *__restrict__ ret, u32 *__restrict__ rng, int n,
                                                                   unsigned long long *stamps)
{
    fill_lds_tables();
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n) return;
    DecWork F;
    F.lds_iy16 = (CA_AS_LDS i16 *)(g_lds_iy16 + threadIdx.x);
    F.lds_pvq16 = (CA_AS_LDS i16 *)(g_lds_pvq16 + threadIdx.x);
    unsigned long long acc[NSTAGES];
    for (int i = 0; i < NSTAGES; i++) acc[i] = 0;
    StageClock clk;
    clk.acc = acc;
    clk.last = __builtin_amdgcn_s_memtime();
    F.diag = &clk;
    DecResult r = celt_decode_front(F, states + k, packets + (size_t)k * packet_stride, len[k]);
    ret[k] = r.samples;
    rng[k] = r.final_range;
    if (threadIdx.x == 0 && blockIdx.x < 4096)
        for (int i = 0; i < NSTAGES; i++) stamps[(size_t)blockIdx.x * NSTAGES + i] += acc[i];
}

}  // namespace ca

// Diagnostic: stage 1 of opusgpu_decode_batch alone (the states advance as usual, no PCM is produced), with
// per-stage cycle stamps of each wavefront. d_stamps: zero-initialised u64 [4096][32].
extern "C" int opusgpu_decode_lane_diag(void *d_states, const unsigned char *d_packets, int packet_stride, const int32_t *d_len,
                                        int32_t *d_ret, uint32_t *d_rng, int n_streams, unsigned long long *d_stamps, void *stream)
{
    if (!d_states || !d_packets || !d_len || !d_ret || !d_rng || !d_stamps || n_streams <= 0) return OPUSGPU_BAD_ARG;
    hipLaunchKernelGGL(ca::celt_decode_lane_diag_kernel, dim3((n_streams + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       (opusgpu_celt_dec_state *)d_states, d_packets, packet_stride, d_len, d_ret, d_rng, n_streams, d_stamps);
    return opusgpu_check_launch();
}
