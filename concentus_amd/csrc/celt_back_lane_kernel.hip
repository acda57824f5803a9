// celt_back_lane_kernel.hip -- back phase (FrameMid -> packet) with ONE LANE per frame.
//
// The back phase is a serial chain of ~160 small steps per frame; with one
// wavefront per frame most of its vector instructions carry 1-16 useful lanes. This build compiles the
// same sources with LANES == 1 (wave.h, CA_LANE_FRAME): 64 frames share a wavefront, every instruction
// does work for all of them, the per-frame working set lives in private memory.
#define CA_LANE_FRAME 1
#include "celt_lane_tables.h"
#include "celt_enc.h"
#include "opusgpu_internal.h"

namespace ca {

__global__ __launch_bounds__(64) void celt_back_lane_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states,
                                                            FrameMid *mid, u8 *out, int out_stride,
                                                            int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    fill_lds_tables();
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= nframes) return;
    BackLds F;
    F.lds_pvq16 = g_lds_pvq16 + threadIdx.x;
    F.lds_pvq32 = g_lds_pvq32 + threadIdx.x;
    F.lds_xs = g_lds_xs + threadIdx.x;
    opusgpu_celt_state *st = states ? states + n : nullptr;
    FrameResult r = celt_encode_back(F, cfg, mid + n, st, out + (size_t)n * out_stride);
    out_len[n] = r.bytes;
    out_rng[n] = r.final_range;
}

}  // namespace ca

extern "C" void opusgpu_launch_back_lane(const opusgpu_celt_config *cfg, void *states, const void *mid, unsigned char *out,
                                         int out_stride, int32_t *out_len, uint32_t *out_rng, int n, hipStream_t s)
{
    hipLaunchKernelGGL(ca::celt_back_lane_kernel, dim3((n + 63) / 64), dim3(64), 0, s, *cfg, (opusgpu_celt_state *)states,
                       (ca::FrameMid *)mid, out, out_stride, out_len, out_rng, n);
}
