// mdct_kernels.hip -- batched CELT MDCT forward / backward for gfx950 (BASELINE config #2).
//
// One 64-lane wavefront per (frame, channel) transform, one transform per one-wave workgroup (nothing is staged per
// workgroup, the dispatcher balances). Data path per transform: the fold / pre-rotation reads the input row straight
// from HBM, the butterfly stages run in the wavefront's LDS scratch (480 complex points, bank-swizzled: mdct_dev.h),
// the post-rotation writes the coefficients straight to HBM; the 5 KB of tables are read through L1.
//
// Replaces clt_mdct_forward_c / clt_mdct_backward_c (opus-fix/celt/mdct.c:121,263) as driven by
// compute_mdcts (celt/celt_encoder.c:418-461) and celt_synthesis (celt/celt_decoder.c:323-346).
#include "mdct_dev.h"
#include "opusgpu_internal.h"
#include <stdlib.h>

namespace ca {

// LDS holds only the FFT scratch of the transform (480 complex points, 3 840 B): the fold reads the input samples and the
// post-rotation writes the coefficients straight from / to HBM. Both walk their rows with 4-byte accesses at an 8-byte
// stride from both ends (mdct.c:155-204, :237-257), so a row's cache lines are touched twice a few hundred nanoseconds
// apart and the second touch hits L2; what that buys is LDS: 3.8 KB instead of 8.2 KB (12 KB backward) per wavefront, i.e.
// all the wavefronts a CU can hold (limited by registers, not LDS) instead of 16, and the chain of one transform -- fold,
// five butterfly stages, post-rotation, each an LDS round trip -- is hidden by the others.
struct __align__(16) MdctFftLds {
    __align__(16) int2 f2[480];
};

// __launch_bounds__(64, 8): the register budget of eight wavefronts per SIMD = all 32 wavefront slots of a CU, so that the 8 192
// transforms of config #2 (32 per CU) are resident in ONE round; the kernels need 53 / 57 VGPRs (mdct_dev.h: every stage derives its
// addresses from an opaque copy of the lane index, which keeps the compiler from hoisting them across stages). Tried on top of this
// and dropped, all measured at 4 096 and 65 536 frames (gpurun_out/r03_c, r03_f): keeping 2 or 4 trips of the pre-rotation's loads in
// flight (no change), a start-up stagger of the wavefront layers of a one-round launch (monotonically slower).
template <int SHIFT>
__global__ __launch_bounds__(64, 8) void mdct_forward_kernel(const i32 *__restrict__ sig, i32 *__restrict__ freq,
                                                               int ntransforms)
{
    constexpr int B = 1 << SHIFT;
    __shared__ MdctFftLds S;
    const int lane = threadIdx.x;
    const MdctTab T = mdct_global_tab<SHIFT>();   // 5 KB of tables: L1-resident
    for (int t = blockIdx.x; t < ntransforms; t += gridDim.x) {
        mdct_forward_wave<SHIFT, B>(sig + (size_t)t * 1080, S.f2, freq + (size_t)t * 960, 1, T, lane);
        wave_sync();
    }
}

template <int SHIFT>
__global__ __launch_bounds__(64, 8) void mdct_backward_kernel(const i32 *__restrict__ freq, i32 *sig, int ntransforms)
{
    constexpr int B = 1 << SHIFT;
    __shared__ MdctFftLds S;
    const int lane = threadIdx.x;
    const MdctTab T = mdct_global_tab<SHIFT>();
    for (int t = blockIdx.x; t < ntransforms; t += gridDim.x) {
        // out[0, 60) (the previous frame's tail) is live on entry, [60, 1020) is produced, [1020, 1080) untouched (mdct.c:345-361)
        mdct_backward_wave<SHIFT, B>(freq + (size_t)t * 960, 1, S.f2, sig + (size_t)t * 1080, T, lane);
        wave_sync();
    }
}

struct __align__(16) MdctFwdLds {
    __align__(16) i32 buf[1080];      // input samples; re-used as the 960 output coefficients
    __align__(16) int2 f2[480];
};
struct __align__(16) MdctBwdLds {
    __align__(16) i32 coef[960];
    __align__(16) i32 out[1080];
    __align__(16) int2 f2[480];
};

// Single transform with an arbitrary output/input stride, for the per-call RTCD-style hook.
template <int SHIFT>
__global__ __launch_bounds__(64) void mdct_forward_single_kernel(const i32 *in, i32 *out, int stride)
{
    __shared__ MdctFwdLds S;
    const int lane = threadIdx.x;
    constexpr int N2 = 960 >> SHIFT;
    const MdctTab T = mdct_global_tab<SHIFT>();   // 5 KB of tables: L1-resident; a copy per one-wave workgroup in LDS would halve the wavefronts a CU holds
    for (int i = lane; i < N2 + 120; i += 64) S.buf[i] = in[i];
    wave_sync();
    mdct_forward_wave<SHIFT, 1>(S.buf, S.f2, out, stride, T, lane);
}

template <int SHIFT>
__global__ __launch_bounds__(64) void mdct_backward_single_kernel(const i32 *in, i32 *out, int stride)
{
    __shared__ MdctBwdLds S;
    const int lane = threadIdx.x;
    constexpr int N2 = 960 >> SHIFT;
    const MdctTab T = mdct_global_tab<SHIFT>();   // 5 KB of tables: L1-resident; a copy per one-wave workgroup in LDS would halve the wavefronts a CU holds
    for (int i = lane; i < N2; i += 64) S.coef[i] = in[i * stride];
    for (int i = lane; i < 120; i += 64) S.out[i] = out[i];
    wave_sync();
    mdct_backward_wave<SHIFT, 1>(S.coef, 1, S.f2, S.out, T, lane);
    for (int i = lane; i < N2 + 60; i += 64) out[i] = S.out[i];
}

// opus_fft_c (opus-fix/celt/kiss_fft.c:580-599): scaled, bit-reversing, out-of-place forward FFT of 480 >> SHIFT complex
// points, one wavefront per transform; fin / fout hold [re, im] int32 pairs.
struct __align__(16) FftLds {
    __align__(16) int2 x[480];
};

template <int SHIFT>
__global__ __launch_bounds__(64) void fft_kernel(const int2 *__restrict__ fin, int2 *__restrict__ fout, int ntransforms)
{
    constexpr int NFFT = 480 >> SHIFT, SCALE_SHIFT = (8 - SHIFT) - 1;    // kiss_fft_state.scale = 17476, scale_shift = 8 - shift
    __shared__ FftLds S;
    const int lane = threadIdx.x;
    const MdctTab T = mdct_global_tab<SHIFT>();   // 5 KB of tables: L1-resident; a copy per one-wave workgroup in LDS would halve the wavefronts a CU holds
    wave_sync();
    for (int t = blockIdx.x; t < ntransforms; t += gridDim.x) {
        const int2 *src = fin + (size_t)t * NFFT;
        for (int i = lane; i < NFFT; i += 64) {
            const int2 v = src[i];
            fft_put<SHIFT, 1>(S.x, T.bitrev[i], cpx{mul16_32_q16(17476, v.x) >> SCALE_SHIFT, mul16_32_q16(17476, v.y) >> SCALE_SHIFT});
        }
        wave_sync();
        fft_wave<SHIFT, 1>(S.x, T.tw, lane);
        int2 *dst = fout + (size_t)t * NFFT;
        for (int i = lane; i < NFFT; i += 64) { const cpx v = fft_get<SHIFT, 1>(S.x, i); dst[i] = make_int2(v.r, v.i); }
        wave_sync();
    }
}

static int grid_for(int ntransforms, int waves_per_cu)
{
    int cap = opusgpu_num_cus() * waves_per_cu;
    return ntransforms < cap ? ntransforms : cap;
}

}  // namespace ca

using namespace ca;

extern "C" int opusgpu_mdct_forward_batch(const int32_t *d_sig, int32_t *d_freq, int n_frames, int channels,
                                          int shift, void *stream)
{
    if (n_frames < 0 || channels < 1 || channels > 2) return OPUSGPU_BAD_ARG;
    if (shift < 0 || shift > 3) return OPUSGPU_BAD_ARG;
    if (shift != 0 && shift != 3) return OPUSGPU_UNIMPLEMENTED;    // 20 ms frames: long or 8 short blocks
    int nt = n_frames * channels;
    if (nt == 0) return OPUSGPU_OK;
    if (!d_sig || !d_freq) return OPUSGPU_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int grid = nt;        // one transform per one-wave workgroup: nothing is staged per workgroup, the dispatcher balances
    if (shift == 0) hipLaunchKernelGGL(mdct_forward_kernel<0>, dim3(grid), dim3(64), 0, s, d_sig, d_freq, nt);
    else            hipLaunchKernelGGL(mdct_forward_kernel<3>, dim3(grid), dim3(64), 0, s, d_sig, d_freq, nt);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_mdct_backward_batch(const int32_t *d_freq, int32_t *d_sig, int n_frames, int channels,
                                           int shift, void *stream)
{
    if (n_frames < 0 || channels < 1 || channels > 2) return OPUSGPU_BAD_ARG;
    if (shift < 0 || shift > 3) return OPUSGPU_BAD_ARG;
    if (shift != 0 && shift != 3) return OPUSGPU_UNIMPLEMENTED;
    int nt = n_frames * channels;
    if (nt == 0) return OPUSGPU_OK;
    if (!d_sig || !d_freq) return OPUSGPU_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int grid = nt;
    if (shift == 0) hipLaunchKernelGGL(mdct_backward_kernel<0>, dim3(grid), dim3(64), 0, s, d_freq, d_sig, nt);
    else            hipLaunchKernelGGL(mdct_backward_kernel<3>, dim3(grid), dim3(64), 0, s, d_freq, d_sig, nt);
    return opusgpu_check_launch();
}

extern "C" int opusgpu_fft_batch(const int32_t *d_fin, int32_t *d_fout, int n_transforms, int shift, void *stream)
{
    if (n_transforms < 0 || shift < 0 || shift > 3) return OPUSGPU_BAD_ARG;
    if (n_transforms == 0) return OPUSGPU_OK;
    if (!d_fin || !d_fout || d_fin == d_fout) return OPUSGPU_BAD_ARG;          // out of place, as opus_fft_c asserts
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for(n_transforms, 12);
    const int2 *fin = (const int2 *)d_fin;
    int2 *fout = (int2 *)d_fout;
    switch (shift) {
    case 0: hipLaunchKernelGGL(fft_kernel<0>, dim3(grid), dim3(64), 0, s, fin, fout, n_transforms); break;
    case 1: hipLaunchKernelGGL(fft_kernel<1>, dim3(grid), dim3(64), 0, s, fin, fout, n_transforms); break;
    case 2: hipLaunchKernelGGL(fft_kernel<2>, dim3(grid), dim3(64), 0, s, fin, fout, n_transforms); break;
    default: hipLaunchKernelGGL(fft_kernel<3>, dim3(grid), dim3(64), 0, s, fin, fout, n_transforms); break;
    }
    return opusgpu_check_launch();
}

// opus_fft(cfg, fin, fout) with the reference's argument list (macro at celt/kiss_fft.h:135-178 -> opus_fft_c): host
// pointers, one transform, synchronous. `cfg` must be one of the four states of the static 48 kHz mode
// (nfft 480 / 240 / 120 / 60, scale 17476, scale_shift 8 - shift); the head of kiss_fft_state is validated, its
// tables are not read. Plumbing / parity only.
struct ref_kiss_fft_state_head { int nfft; int16_t scale; int scale_shift; int shift; };

extern "C" void opusgpu_opus_fft(const void *cfg, const void *fin, void *fout)
{
    const ref_kiss_fft_state_head *h = (const ref_kiss_fft_state_head *)cfg;
    int shift = -1;
    if (h && fin && fout && fin != fout)
        for (int k = 0; k < 4; k++)
            if (h->nfft == (480 >> k) && h->scale == 17476 && h->scale_shift == 8 - k) shift = k;
    if (shift < 0) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    const size_t bytes = (size_t)(480 >> shift) * 8;
    int32_t *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, bytes) != hipSuccess || hipMalloc(&d_out, bytes) != hipSuccess) {
        opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL);
        if (d_in) (void)hipFree(d_in);
        return;
    }
    int rc = opusgpu_copy(d_in, fin, bytes, hipMemcpyHostToDevice);
    if (rc == OPUSGPU_OK) rc = opusgpu_fft_batch(d_in, d_out, 1, shift, nullptr);
    if (rc == OPUSGPU_OK) rc = opusgpu_copy(fout, d_out, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    opusgpu_set_last_error(rc);
}

// ---- per-call hooks with the reference's own signature (host pointers) ---------------------------
// Same argument list as clt_mdct_forward_c / clt_mdct_backward_c (celt/mdct.h:77-110). `l`, `window`
// and `overlap` must describe the static mode (n = 1920, window120, 120); they are validated, not read
// on the device. Latency-dominated: for plumbing/parity only, throughput comes from the batch API.
struct ref_mdct_lookup_head { int n; int maxshift; };

static int hook_args_ok(const void *l, const void *window, int overlap, int shift, int stride)
{
    if (!l || !window || overlap != 120 || shift < 0 || shift > 3 || stride < 1) return 0;
    const ref_mdct_lookup_head *h = (const ref_mdct_lookup_head *)l;
    return h->n == 1920 && h->maxshift == 3;
}

extern "C" void opusgpu_clt_mdct_forward(const void *l, int32_t *in, int32_t *out, const int16_t *window,
                                         int overlap, int shift, int stride, int arch)
{
    (void)arch;
    if (!hook_args_ok(l, window, overlap, shift, stride)) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    const int n2 = 960 >> shift;
    const size_t in_bytes = (size_t)(n2 + 120) * 4, out_elems = (size_t)(n2 - 1) * stride + 1;
    int32_t *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, in_bytes) != hipSuccess || hipMalloc(&d_out, out_elems * 4) != hipSuccess) {
        opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL);
        if (d_in) (void)hipFree(d_in);
        return;
    }
    int rc = opusgpu_copy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (rc == OPUSGPU_OK) rc = opusgpu_copy(d_out, out, out_elems * 4, hipMemcpyHostToDevice);   // keep the gaps when stride > 1
    if (rc == OPUSGPU_OK) switch (shift) {
    case 0: hipLaunchKernelGGL(mdct_forward_single_kernel<0>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    case 1: hipLaunchKernelGGL(mdct_forward_single_kernel<1>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    case 2: hipLaunchKernelGGL(mdct_forward_single_kernel<2>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    default: hipLaunchKernelGGL(mdct_forward_single_kernel<3>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    }
    if (rc == OPUSGPU_OK) rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK) rc = opusgpu_copy(out, d_out, out_elems * 4, hipMemcpyDeviceToHost);
    opusgpu_set_last_error(rc);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
}

extern "C" void opusgpu_clt_mdct_backward(const void *l, int32_t *in, int32_t *out, const int16_t *window,
                                          int overlap, int shift, int stride, int arch)
{
    (void)arch;
    if (!hook_args_ok(l, window, overlap, shift, stride)) { opusgpu_set_last_error(OPUSGPU_BAD_ARG); return; }
    const int n2 = 960 >> shift;
    const size_t in_elems = (size_t)(n2 - 1) * stride + 1, out_bytes = (size_t)(n2 + 120) * 4;
    int32_t *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, in_elems * 4) != hipSuccess || hipMalloc(&d_out, out_bytes) != hipSuccess) {
        opusgpu_set_last_error(OPUSGPU_ALLOC_FAIL);
        if (d_in) (void)hipFree(d_in);
        return;
    }
    int rc = opusgpu_copy(d_in, in, in_elems * 4, hipMemcpyHostToDevice);
    if (rc == OPUSGPU_OK) rc = opusgpu_copy(d_out, out, out_bytes, hipMemcpyHostToDevice);
    if (rc == OPUSGPU_OK) switch (shift) {
    case 0: hipLaunchKernelGGL(mdct_backward_single_kernel<0>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    case 1: hipLaunchKernelGGL(mdct_backward_single_kernel<1>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    case 2: hipLaunchKernelGGL(mdct_backward_single_kernel<2>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    default: hipLaunchKernelGGL(mdct_backward_single_kernel<3>, dim3(1), dim3(64), 0, 0, d_in, d_out, stride); break;
    }
    if (rc == OPUSGPU_OK) rc = opusgpu_check_launch();
    if (rc == OPUSGPU_OK) rc = opusgpu_copy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
    opusgpu_set_last_error(rc);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
}
