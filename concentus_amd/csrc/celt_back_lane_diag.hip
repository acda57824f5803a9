// celt_back_lane_diag.hip -- DIAGNOSTIC build of the lane-per-frame back kernel with s_memtime stage
// stamps (-DCA_STAGE_TIMING). Never used for reported throughput; the per-stage cycle totals of each
// wavefront go to a buffer of their own and feed no output.
#define CA_LANE_FRAME 1
#define CA_LANE_SLOTS 264             // 16-bit slots of a lane's LDS column (celt_enc_front.h LS_SLOTS, celt_enc_lane.h)
#define CA_STAGE_TIMING 1
#include "celt_lane_tables.h"
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

enum { NSTAGES = 32 };

__global__ __launch_bounds__(64) void celt_back_lane_diag_kernel(opusgpu_celt_config cfg, FrameMid *mid, u8 *out, int out_stride,
                                                                 int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes,
                                                                 unsigned long long *stamps)
{
    fill_lds_tables();
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= nframes) return;
    BackLds F;
    F.col = (CA_AS_LDS i16 *)(g_lds_scratch + threadIdx.x);
    unsigned long long acc[NSTAGES];
    for (int k = 0; k < NSTAGES; k++) acc[k] = 0;
    StageClock clk;
    clk.acc = acc;
    clk.last = __builtin_amdgcn_s_memtime();
    FrameResult r = celt_encode_back(F, cfg, mid + n, (opusgpu_celt_state *)nullptr, out + (size_t)n * out_stride, &clk);
    out_len[n] = r.bytes;
    out_rng[n] = r.final_range;
    if (threadIdx.x == 0 && blockIdx.x < 4096)
        for (int k = 0; k < NSTAGES; k++) stamps[(size_t)blockIdx.x * NSTAGES + k] += acc[k];
}

}  // namespace ca

// Runs ONLY the back phase, lane-per-frame, on FrameMid records already in d_workspace (e.g. left there by
// opusgpu_encode_batch with the same n_frames). d_stamps: zero-initialised u64 [4096][32], one row per wavefront.
extern "C" int opusgpu_back_lane_diag(const opusgpu_celt_config *cfg, void *d_workspace, unsigned char *d_out, int out_stride,
                                      int32_t *d_out_len, uint32_t *d_out_rng, int n_frames, unsigned long long *d_stamps,
                                      void *stream)
{
    if (!cfg || !d_workspace || !d_out || !d_out_len || !d_out_rng || !d_stamps || n_frames <= 0) return OPUSGPU_BAD_ARG;
    hipLaunchKernelGGL(ca::celt_back_lane_diag_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, (hipStream_t)stream, *cfg,
                       (ca::FrameMid *)d_workspace, d_out, out_stride, d_out_len, d_out_rng, n_frames, d_stamps);
    return opusgpu_check_launch();
}
