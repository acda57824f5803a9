// celt_enc_kernel_diag.hip -- DIAGNOSTIC build of the two frame kernels with s_memtime stage stamps
// (-DCA_STAGE_TIMING). Never used for reported throughput: the stamps serialise the stages. The
// per-stage cycle totals go to a buffer of their own and feed no output.
#define CA_STAGE_TIMING 1
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

enum { NSTAGES = 32 };

__global__ __launch_bounds__(64, 2) void celt_front_diag_kernel(opusgpu_celt_config cfg, const i16 *__restrict__ pcm,
                                                                FrameMid *__restrict__ mid, int nframes, unsigned long long *stamps)
{
    __shared__ FrontLds F;
    unsigned long long acc[NSTAGES];
    for (int k = 0; k < NSTAGES; k++) acc[k] = 0;
    StageClock clk;
    clk.acc = acc;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        clk.last = __builtin_amdgcn_s_memtime();
        celt_encode_front(F, cfg, (const opusgpu_celt_state *)nullptr, (opusgpu_celt_state *)nullptr,
                          pcm + (size_t)n * FRAME * cfg.channels, mid + n, &clk);
        wave_sync();
    }
    if (lane() == 0)
        for (int k = 0; k < NSTAGES; k++) stamps[(size_t)blockIdx.x * NSTAGES + k] = acc[k];
}

__global__ __launch_bounds__(64, 4) void celt_back_diag_kernel(opusgpu_celt_config cfg, const FrameMid *__restrict__ mid,
                                                               u8 *__restrict__ out, int out_stride, int *__restrict__ out_len,
                                                               u32 *__restrict__ out_rng, int nframes, unsigned long long *stamps)
{
    __shared__ BackLds F;
    unsigned long long acc[NSTAGES];
    for (int k = 0; k < NSTAGES; k++) acc[k] = 0;
    StageClock clk;
    clk.acc = acc;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        clk.last = __builtin_amdgcn_s_memtime();
        FrameResult r = celt_encode_back(F, cfg, mid + n, (opusgpu_celt_state *)nullptr, out + (size_t)n * out_stride, &clk);
        if (lane() == 0) { out_len[n] = r.bytes; out_rng[n] = r.final_range; }
        wave_sync();
    }
    if (lane() == 0)
        for (int k = 0; k < NSTAGES; k++) stamps[(size_t)blockIdx.x * NSTAGES + k] += acc[k];
}

}  // namespace ca

using namespace ca;

// d_stamps: device, zero-initialised u64 [4096][32]; independent frames only; n_frames <= workspace capacity
extern "C" int opusgpu_encode_batch_diag(const opusgpu_celt_config *cfg, const int16_t *d_pcm, unsigned char *d_out,
                                         int out_stride, int32_t *d_out_len, uint32_t *d_out_rng, int n_frames,
                                         void *d_workspace, size_t workspace_bytes, unsigned long long *d_stamps, void *stream)
{
    if (!cfg || !d_pcm || !d_out || !d_out_len || !d_out_rng || !d_stamps || !d_workspace || n_frames <= 0) return OPUSGPU_BAD_ARG;
    if (workspace_bytes < (size_t)n_frames * sizeof(FrameMid)) return OPUSGPU_BUFFER_TOO_SMALL;
    const int cus = opusgpu_num_cus();
    int g1 = n_frames < cus * 6 ? n_frames : cus * 6, g2 = n_frames < cus * 16 ? n_frames : cus * 16;
    if (g1 > 4096) g1 = 4096;
    if (g2 > 4096) g2 = 4096;
    hipLaunchKernelGGL(celt_front_diag_kernel, dim3(g1), dim3(64), 0, (hipStream_t)stream, *cfg, d_pcm, (FrameMid *)d_workspace,
                       n_frames, d_stamps);
    hipLaunchKernelGGL(celt_back_diag_kernel, dim3(g2), dim3(64), 0, (hipStream_t)stream, *cfg, (const FrameMid *)d_workspace,
                       d_out, out_stride, d_out_len, d_out_rng, n_frames, d_stamps);
    return opusgpu_check_launch();
}
