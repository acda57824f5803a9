/* opusgpu.h -- C-ABI of the MI355X (gfx950) batched Opus frame path.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only. Two families of entry points:
 *
 *  (1) BATCH entry points (`*_batch`): device pointers (HBM-resident buffers) + a hipStream_t passed
 *      as void*; asynchronous on that stream. These carry the throughput.
 *  (2) PER-CALL hooks with exactly the reference's own signatures and HOST pointers, so they can be
 *      plugged where the reference reaches its kernels through the `arch`-indexed RTCD macros
 *      (opus-fix/celt/cpu_support.h:34-68, pattern instance celt/arm/arm_celt_map.c:35-120).
 *      Synchronous; for plumbing and parity, not speed.
 *
 * Return values of the int-returning functions follow opus-fix/include/opus_defines.h:46-60.
 * INTEGRATION.md shows the reference-side bindings.
 */
#ifndef OPUSGPU_H
#define OPUSGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPUSGPU_OK               0   /* OPUS_OK */
#define OPUSGPU_BAD_ARG         -1   /* OPUS_BAD_ARG */
#define OPUSGPU_BUFFER_TOO_SMALL -2  /* OPUS_BUFFER_TOO_SMALL */
#define OPUSGPU_INTERNAL_ERROR  -3   /* OPUS_INTERNAL_ERROR (HIP launch/runtime failure) */
#define OPUSGPU_INVALID_PACKET  -4   /* OPUS_INVALID_PACKET */
#define OPUSGPU_UNIMPLEMENTED   -5   /* OPUS_UNIMPLEMENTED */
#define OPUSGPU_INVALID_STATE   -6   /* OPUS_INVALID_STATE */
#define OPUSGPU_ALLOC_FAIL      -7   /* OPUS_ALLOC_FAIL */

/* ---- runtime ------------------------------------------------------------------------------------ */
/* Version string of this library ("opusgpu <semver> gfx950"); cf. opus_get_version_string(). */
const char *opusgpu_get_version_string(void);
/* Human-readable text for an error code; cf. opus_strerror() (opus-fix/celt/celt.c). */
const char *opusgpu_strerror(int error);
/* Error recorded by the last void-returning hook on this thread (OPUSGPU_OK if none). */
int opusgpu_get_last_error(void);
/* Number of compute units of the current HIP device (256 on MI355X). */
int opusgpu_num_cus(void);

/* ---- CELT MDCT, batched (BASELINE config #2) -----------------------------------------------------
 * Replaces clt_mdct_forward_c / clt_mdct_backward_c (opus-fix/celt/mdct.c:121-259, :263-363) as they
 * are driven per frame by compute_mdcts (celt/celt_encoder.c:418-461) and celt_synthesis
 * (celt/celt_decoder.c:323-346), for the static mode 48000/960 (overlap 120, window120), 20 ms frames.
 *
 *   d_sig   int32 [n_frames][channels][1080]  time-domain celt_sig (Q12): 120 overlap + 960 new samples
 *   d_freq  int32 [n_frames][channels][960]   MDCT coefficients
 *   shift   0: one long block (N=1920);  3: eight short blocks (N=240, hop 120), coefficients
 *           interleaved with stride 8 exactly as compute_mdcts/celt_synthesis lay them out.
 *
 * forward:  reads d_sig, writes d_freq (d_sig is NOT trashed, unlike the reference, mdct.h:64).
 * backward: reads d_freq and d_sig[..][0..60) (previous tail), writes d_sig[..][0..1020);
 *           d_sig[..][1020..1080) is left untouched, as in the reference.
 */
int opusgpu_mdct_forward_batch(const int32_t *d_sig, int32_t *d_freq, int n_frames, int channels,
                               int shift, void *hip_stream);
int opusgpu_mdct_backward_batch(const int32_t *d_freq, int32_t *d_sig, int n_frames, int channels,
                                int shift, void *hip_stream);

/* ---- CELT MDCT, per-call hooks (host pointers) ----------------------------------------------------
 * Signature = CLT_MDCT_FORWARD_IMPL / CLT_MDCT_BACKWARD_IMPL table entries (opus-fix/celt/mdct.h:77-110):
 *   void f(const mdct_lookup *l, kiss_fft_scalar *in, kiss_fft_scalar *out,
 *          const opus_val16 *window, int overlap, int shift, int stride, int arch)
 * `l` is passed as const void* (it must be &mode48000_960_120.mdct: n == 1920, maxshift == 3). */
void opusgpu_clt_mdct_forward(const void *l, int32_t *in, int32_t *out, const int16_t *window,
                              int overlap, int shift, int stride, int arch);
void opusgpu_clt_mdct_backward(const void *l, int32_t *in, int32_t *out, const int16_t *window,
                               int overlap, int shift, int stride, int arch);

#ifdef __cplusplus
}
#endif
#endif /* OPUSGPU_H */
