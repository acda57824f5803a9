// celt_back_lane_kernel.hip -- back phase (FrameMid -> packet) with ONE LANE per frame.
//
// The back phase is a serial chain of ~160 small steps per frame; with one
// wavefront per frame most of its vector instructions carry 1-16 useful lanes. This build compiles the
// sources with LANES == 1 (wave.h, CA_LANE_FRAME): 64 frames share a wavefront, every instruction does work
// for all of them. The per-frame working set is NOT in private memory (round 3, celt_enc_lane.h): a lane owns
// a column of 264 16-bit slots of the workgroup's LDS, its rows in HBM and registers.
#define CA_LANE_FRAME 1
#define CA_LANE_SLOTS 264             // 16-bit slots of a lane's LDS column (celt_enc_front.h LS_SLOTS, celt_enc_lane.h)
#include "celt_lane_tables.h"
#include "celt_enc.h"
#include "opusgpu_internal.h"
#include <stdlib.h>

namespace ca {

// A = frames per wavefront. A == 64 fills every lane; A == 32 leaves the upper lanes of each wavefront idle and
// puts 2 wavefronts into the 64-frame workgroup instead (same LDS footprint, [element][64] slots shared by the
// workgroup's waves). While the per-frame working set was private memory (rounds 1-2) the half-filled waves, two per SIMD,
// won: they hid each other's scratch latency. With the working set in the LDS column the full wave wins (the default, below):
// its instruction count is hardly larger than a half-filled one's, so the SIMD issues half as much.
template <int A>
__device__ __forceinline__ void back_lane_body(const opusgpu_celt_config &cfg, opusgpu_celt_state *states, FrameMid *mid, u8 *out,
                                               int out_stride, int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    fill_lds_tables();
    const int l = threadIdx.x & 63;
    if (l >= A) return;
    const int slot = (threadIdx.x >> 6) * A + l;
    const int n = blockIdx.x * 64 + slot;
    if (n >= nframes) return;
    BackLds F;
    F.col = (CA_AS_LDS i16 *)(g_lds_scratch + slot);
    opusgpu_celt_state *st = states ? states + n : nullptr;
    FrameResult r = celt_encode_back(F, cfg, mid + n, st, out + (size_t)n * out_stride);
    out_len[n] = r.bytes;
    out_rng[n] = r.final_range;
}

template <int A> __global__ void celt_back_lane_kernel(opusgpu_celt_config cfg, opusgpu_celt_state *states, FrameMid *mid, u8 *out,
                                                       int out_stride, int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes);

// one full wavefront per SIMD (the LDS columns of 4 x 64 frames fill a CU): the compiler is told so -- it may spend the whole
// register file on keeping loads in flight instead of saving registers for wavefronts that cannot come
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void celt_back_lane_kernel<64>(opusgpu_celt_config cfg, opusgpu_celt_state *states, FrameMid *mid, u8 *out, int out_stride,
                               int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    back_lane_body<64>(cfg, states, mid, out, out_stride, out_len, out_rng, nframes);
}

template <>
__global__ __launch_bounds__(128)
void celt_back_lane_kernel<32>(opusgpu_celt_config cfg, opusgpu_celt_state *states, FrameMid *mid, u8 *out, int out_stride,
                               int *__restrict__ out_len, u32 *__restrict__ out_rng, int nframes)
{
    back_lane_body<32>(cfg, states, mid, out, out_stride, out_len, out_rng, nframes);
}

}  // namespace ca

// frames per wavefront of the lane kernels: OPUSGPU_LANE_FRAMES = 64 | 32 overrides the per-kernel default
extern "C" int opusgpu_lane_frames(int dflt)
{
    // measured on MI355X at 65 536 frames. Round 2 (4.3 KB of private working set per frame): back kernel 5.91 ms (64) / 5.53 ms
    // (32) / 9.87 ms (16). Round 3, working set in the LDS column: 3.24 ms (64) / 3.31 ms (32); narrower wavefronts (three of
    // 21-22 frames at 168 VGPRs: 5.14 ms, four of 16 at 128 VGPRs: 4.50 ms) run chains that are hardly shorter, and there are
    // more of them to issue -- their kernels are gone. Read per call so tests can compare the mappings.
    const char *e = getenv("OPUSGPU_LANE_FRAMES");
    const int v = e ? atoi(e) : dflt;
    return (v == 32 || v == 64) ? v : dflt;
}

// default of the encoder's back kernel: one full wavefront per SIMD. Its vector-ALU instruction count is that of a half-filled
// one (645 k against 605 k per wavefront: the divergence of the partition walks saturates), so a SIMD issues half as many,
// and what the second wavefront hid is hidden by loads issued ahead in the kernel's own instruction stream.
enum { BACK_LANE_FRAMES_DEFAULT = 64 };

extern "C" void opusgpu_launch_back_lane(const opusgpu_celt_config *cfg, void *states, const void *mid, unsigned char *out,
                                         int out_stride, int32_t *out_len, uint32_t *out_rng, int n, hipStream_t s)
{
    const int a = opusgpu_lane_frames(BACK_LANE_FRAMES_DEFAULT);
    const dim3 grid((n + 63) / 64), block(64 * (64 / a));
#define CA_LAUNCH(A) hipLaunchKernelGGL(ca::celt_back_lane_kernel<A>, grid, block, 0, s, *cfg, (opusgpu_celt_state *)states, \
                                        (ca::FrameMid *)mid, out, out_stride, out_len, out_rng, n)
    if (a == 32) CA_LAUNCH(32);
    else CA_LAUNCH(64);
#undef CA_LAUNCH
}
