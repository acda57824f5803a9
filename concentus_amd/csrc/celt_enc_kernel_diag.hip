// celt_enc_kernel_diag.hip -- DIAGNOSTIC build of the frame kernel with s_memtime stage stamps
// (-DCA_STAGE_TIMING). Never used for reported throughput: the stamps serialise the stages. The
// per-stage cycle shares it produces are written to a buffer of their own and feed no output.
#define CA_STAGE_TIMING 1
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

enum { NSTAGES = 16 };

__global__ __launch_bounds__(64, 2) void celt_encode_diag_kernel(opusgpu_celt_config cfg, const i16 *__restrict__ pcm,
                                                                 u8 *__restrict__ out, int out_stride, int *__restrict__ out_len,
                                                                 u32 *__restrict__ out_rng, int nframes, unsigned long long *stamps)
{
    __shared__ FrameLds F;
    unsigned long long acc[NSTAGES];
    for (int k = 0; k < NSTAGES; k++) acc[k] = 0;
    StageClock clk;
    clk.acc = acc;
    for (int n = blockIdx.x; n < nframes; n += gridDim.x) {
        clk.last = __builtin_amdgcn_s_memtime();
        FrameResult r = celt_encode_frame(F, cfg, nullptr, nullptr, pcm + (size_t)n * FRAME * cfg.channels,
                                          out + (size_t)n * out_stride, &clk);
        if (lane() == 0) { out_len[n] = r.bytes; out_rng[n] = r.final_range; }
        wave_sync();
    }
    if (lane() == 0)
        for (int k = 0; k < NSTAGES; k++) stamps[(size_t)blockIdx.x * NSTAGES + k] = acc[k];
}

}  // namespace ca

using namespace ca;

// stamps: device, u64 [grid][16]; returns the grid size used (or a negative error)
extern "C" int opusgpu_encode_batch_diag(const opusgpu_celt_config *cfg, const int16_t *d_pcm, unsigned char *d_out,
                                         int out_stride, int32_t *d_out_len, uint32_t *d_out_rng, int n_frames,
                                         unsigned long long *d_stamps, int max_grid, void *stream)
{
    if (!cfg || !d_pcm || !d_out || !d_out_len || !d_out_rng || !d_stamps || n_frames <= 0) return OPUSGPU_BAD_ARG;
    int cap = opusgpu_num_cus() * 6;
    if (cap > max_grid) cap = max_grid;
    int grid = n_frames < cap ? n_frames : cap;
    hipLaunchKernelGGL(celt_encode_diag_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, *cfg, d_pcm, d_out, out_stride,
                       d_out_len, d_out_rng, n_frames, d_stamps);
    int rc = opusgpu_check_launch();
    return rc < 0 ? rc : grid;
}
