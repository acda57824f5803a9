// celt_stage_kernels.hip -- lane-per-(frame, channel) kernels for the strictly serial stages of the front
// phase (see celt_stage_lane.h): 64 independent recurrences per wavefront instead of 2.
#include "celt_enc.h"
#include "celt_stage_lane.h"
#include "opusgpu_internal.h"

namespace ca {

static_assert(offsetof(FrameMid, X) % 16 == 0 && sizeof(FrameMid) % 16 == 0, "FrameMid::X is moved 16 bytes at a time");

// dc_reject: pcm [n][960][2] int16 -> mid[f].X planar [2][960] int16, filter memory -> mid[f].hp_mem
__global__ __launch_bounds__(256) void celt_dc_reject_kernel(const opusgpu_celt_state *states, const i16 *__restrict__ pcm,
                                                             FrameMid *__restrict__ mid, int nframes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = t >> 1, c = t & 1;
    if (f >= nframes) return;
    i32 hp[2] = {0, 0};
    if (states) { hp[0] = states[f].hp_mem[2 * c]; hp[1] = states[f].hp_mem[2 * c + 1]; }
    stage_dc_reject_channel(pcm + (size_t)f * FRAME * 2, c, hp, mid[f].X + c * FRAME);
    mid[f].hp_mem[2 * c] = hp[0];
    mid[f].hp_mem[2 * c + 1] = hp[1];
}

// transient metric per channel: in_ws [n][2][1080] int32 -> mid[f].trans_unmask[c]; mid[f].X is scratch here
__global__ __launch_bounds__(256) void celt_transient_kernel(FrameMid *__restrict__ mid, const i32 *__restrict__ in_ws, int nframes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = t >> 1, c = t & 1;
    if (f >= nframes) return;
    mid[f].trans_unmask[c] = stage_transient_channel(in_ws + ((size_t)f * 2 + c) * (FRAME + OVL), mid[f].X + c * FRAME);
}

}  // namespace ca

extern "C" void opusgpu_launch_dc_reject(const void *states, const int16_t *pcm, void *mid, int n, hipStream_t s)
{
    hipLaunchKernelGGL(ca::celt_dc_reject_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, s, (const opusgpu_celt_state *)states, pcm,
                       (ca::FrameMid *)mid, n);
}

extern "C" void opusgpu_launch_transient(void *mid, const int32_t *in_ws, int n, hipStream_t s)
{
    hipLaunchKernelGGL(ca::celt_transient_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, s, (ca::FrameMid *)mid, in_ws, n);
}
