// pitch_hooks.hip -- celt_pitch_xcorr() with the reference's own argument list, as a per-call hook (host pointers).
//
// opus-fix/celt/pitch.c:214-258 (celt_pitch_xcorr), the CELT_PITCH_XCORR_IMPL[] table entry of celt/pitch.h:186-204:
//   xcorr[i] = sum_j x[j] * y[i + j]   (MAC16_16 in wrapping 32-bit arithmetic),  i < max_pitch,  j < len
//   returns max(1, max_i xcorr[i])
// Inside the frame encoder the same sums are computed by the front kernel on LDS-resident buffers (celt_enc_front.h);
// this entry point exists so the reference's RTCD slot has a target, for plumbing and parity. One lane per lag, x
// staged in LDS, the maximum by a wave reduction and one atomic per wavefront.
#include "fixmath.h"
#include "opusgpu_internal.h"

namespace ca {

enum { XCORR_MAX_LEN = 2048 };

__global__ __launch_bounds__(256) void pitch_xcorr_kernel(const i16 *__restrict__ x, const i16 *__restrict__ y,
                                                          i32 *__restrict__ xcorr, int len, int max_pitch, i32 *maxcorr)
{
    __shared__ i16 xs[XCORR_MAX_LEN];
    for (int j = threadIdx.x; j < len; j += blockDim.x) xs[j] = x[j];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    i32 sum = (i32)0x80000000;
    if (i < max_pitch) {
        u32 acc = 0;
        for (int j = 0; j < len; j++) acc += (u32)__mul24((i32)xs[j], (i32)y[i + j]);
        sum = (i32)acc;
        xcorr[i] = sum;
    }
    const i32 m = wave_max(sum);
    if ((threadIdx.x & 63) == 0) atomicMax(maxcorr, m);
}

}  // namespace ca

extern "C" int32_t opusgpu_celt_pitch_xcorr(const int16_t *x, const int16_t *y, int32_t *xcorr, int len, int max_pitch, int arch)
{
    (void)arch;
    if (!x || !y || !xcorr || len < 1 || len > ca::XCORR_MAX_LEN || max_pitch < 1) {
        opusgpu_set_last_error(OPUSGPU_BAD_ARG);
        return 0;
    }
    const size_t xb = (size_t)len * 2, yb = (size_t)(len + max_pitch - 1) * 2, cb = (size_t)max_pitch * 4;
    int16_t *d_x = nullptr, *d_y = nullptr;
    int32_t *d_c = nullptr, *d_m = nullptr, h_m = 1;
    int rc = OPUSGPU_OK;
    if (hipMalloc(&d_x, xb) != hipSuccess || hipMalloc(&d_y, yb) != hipSuccess || hipMalloc(&d_c, cb) != hipSuccess ||
        hipMalloc(&d_m, 4) != hipSuccess)
        rc = OPUSGPU_ALLOC_FAIL;
    if (rc == OPUSGPU_OK) {
        rc = opusgpu_copy(d_x, x, xb, hipMemcpyHostToDevice);
        if (rc == OPUSGPU_OK) rc = opusgpu_copy(d_y, y, yb, hipMemcpyHostToDevice);
        if (rc == OPUSGPU_OK) rc = opusgpu_copy(d_m, &h_m, 4, hipMemcpyHostToDevice);         // maxcorr starts at 1 (pitch.c:224)
        if (rc == OPUSGPU_OK) {
            hipLaunchKernelGGL(ca::pitch_xcorr_kernel, dim3((max_pitch + 255) / 256), dim3(256), 0, 0, d_x, d_y, d_c, len, max_pitch, d_m);
            rc = opusgpu_check_launch();
        }
        if (rc == OPUSGPU_OK && (hipMemcpy(xcorr, d_c, cb, hipMemcpyDeviceToHost) != hipSuccess ||
                                 hipMemcpy(&h_m, d_m, 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = OPUSGPU_INTERNAL_ERROR;
    }
    if (d_x) (void)hipFree(d_x);
    if (d_y) (void)hipFree(d_y);
    if (d_c) (void)hipFree(d_c);
    if (d_m) (void)hipFree(d_m);
    opusgpu_set_last_error(rc);
    return rc == OPUSGPU_OK ? h_m : 0;
}
