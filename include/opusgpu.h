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
#include <stddef.h>
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

/* Optional per-kernel timing, the hook bench.py's roofline leg uses: while enabled, every kernel launch
 * of opusgpu_encode_batch / opusgpu_decode_batch is bracketed by HIP events recorded on the launch stream.
 * opusgpu_kernel_timing_read waits for the recorded launches, writes the summed duration (ms) and the
 * launch count per kernel id into ms_sum[0..n_kernels) / launches[0..n_kernels), and forgets them.
 * No counterpart in the reference (its timing is wall clock in src/opus_demo.c:750-800). */
#define OPUSGPU_KERNEL_CELT_FRONT 0
#define OPUSGPU_KERNEL_CELT_BACK  1
#define OPUSGPU_KERNEL_CELT_BACK_LANE 2
#define OPUSGPU_KERNEL_CELT_DC_REJECT 3
#define OPUSGPU_KERNEL_CELT_FRONT1    4
#define OPUSGPU_KERNEL_CELT_TRANSIENT 5
#define OPUSGPU_KERNEL_CELT_FRONT2    6
#define OPUSGPU_KERNEL_DEC_LANE       7
#define OPUSGPU_KERNEL_DEC_SYNTH      8
#define OPUSGPU_KERNEL_DEC_POST       9
#define OPUSGPU_KERNEL_COUNT      10
int opusgpu_kernel_timing_enable(int on);
int opusgpu_kernel_timing_read(double *ms_sum, int *launches, int n_kernels);

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

/* ---- kiss_fft forward transform (opus_fft_c, opus-fix/celt/kiss_fft.c:580-599) ---------------------------
 * Batched: n_transforms independent FFTs of 480 >> shift complex points, d_fin / d_fout int32 [n][nfft][2] (re, im)
 * device pointers, out of place; scaling and bit reversal as opus_fft_c with the static mode's states.
 * Per call: the argument list of the opus_fft macro (celt/kiss_fft.h:135-178), host pointers; `cfg` must be one of
 * mode48000_960_120's four kiss_fft_state objects (validated by nfft / scale / scale_shift). */
int opusgpu_fft_batch(const int32_t *d_fin, int32_t *d_fout, int n_transforms, int shift, void *hip_stream);
void opusgpu_opus_fft(const void *cfg, const void *fin, void *fout);

/* ---- celt_pitch_xcorr, per call (opus-fix/celt/pitch.c:214-258; CELT_PITCH_XCORR_IMPL[] entry, celt/pitch.h:186-204) --
 * The reference's argument list, host pointers: xcorr[i] = sum_j x[j]*y[i+j] for i < max_pitch, j < len (y holds
 * len + max_pitch - 1 samples); returns max(1, max xcorr), 0 with opusgpu_get_last_error() set on failure. len <= 2048. */
int32_t opusgpu_celt_pitch_xcorr(const int16_t *x, const int16_t *y, int32_t *xcorr, int len, int max_pitch, int arch);

/* ---- Opus CELT-only frame encode, batched (BASELINE config #3) -------------------------------------
 * Replaces opus_encode() (opus-fix/src/opus_encoder.c:2007-2025 -> opus_encode_native :938 ->
 * celt_encode_with_ec, opus-fix/celt/celt_encoder.c:1379) for encoders created as
 *   opus_encoder_create(48000, channels, OPUS_APPLICATION_RESTRICTED_LOWDELAY)
 * and configured with the ctl sequence of opus_demo (src/opus_demo.c:531-543), 20 ms frames (960 samples).
 * Output packets (TOC byte + CELT payload) and OPUS_GET_FINAL_RANGE values are bit-exact with the
 * FIXED_POINT reference.
 *
 * opusgpu_celt_config mirrors the ctl-settable encoder configuration; opusgpu_celt_state is the
 * pointer-free per-stream state (what the reference keeps inside OpusEncoder/CELTEncoder between
 * frames, celt_encoder.c:82-128), one record per stream in device memory.
 */
typedef struct opusgpu_celt_config {
    int32_t channels;          /* 2; (1 and stereo->mono downmix are not implemented) */
    int32_t bitrate;           /* OPUS_SET_BITRATE, bits/s (>= 32000: stereo + fullband region) */
    int32_t vbr;               /* OPUS_SET_VBR */
    int32_t constrained_vbr;   /* OPUS_SET_VBR_CONSTRAINT */
    int32_t complexity;        /* OPUS_SET_COMPLEXITY 0..10 */
    int32_t lsb_depth;         /* OPUS_SET_LSB_DEPTH, 8..24; the int16 entry points use min(16, lsb_depth) as opus_encode does */
    int32_t loss_rate;         /* OPUS_SET_PACKET_LOSS_PERC */
    int32_t max_data_bytes;    /* opus_encode()'s max_data_bytes (opus_demo passes 1500) */
} opusgpu_celt_config;

/* sizeof(opusgpu_celt_state): bytes per stream record. */
int opusgpu_celt_state_size(void);
/* Initialise n_streams records to the state of a freshly created encoder (OPUS_RESET_STATE). */
int opusgpu_celt_state_init(void *d_states, int n_streams, void *hip_stream);

/* Encode one 20 ms frame for each of n_frames streams.
 *   d_states   device, n_frames records (frame n advances stream n), or NULL: every frame is encoded as
 *              the first frame of its own fresh stream ("independent frames") and no state is kept.
 *   d_pcm      device, int16 [n_frames][960][channels] interleaved (as opus_encode's pcm).
 *   d_out      device, packet slab [n_frames][out_stride] bytes; out_stride % 4 == 0 and
 *              out_stride >= min(max_data_bytes, 1276) rounded up to 4.
 *   d_out_len  device, int32 [n_frames]: packet length, or a negative OPUSGPU_* code for that frame
 *              (opus_encode's return value).
 *   d_out_rng  device, uint32 [n_frames]: OPUS_GET_FINAL_RANGE after the frame.
 *   d_workspace  device scratch for the hand-off between the two kernels of the path; size it with
 *              opusgpu_encode_workspace_bytes(n). A smaller workspace is legal: the batch is then processed
 *              in chunks of workspace_bytes / opusgpu_encode_workspace_bytes(1) frames.
 * Asynchronous on hip_stream (no allocation, no synchronisation inside: graph-capturable). Returns
 * OPUSGPU_UNIMPLEMENTED for configurations outside the stereo/fullband CELT-only operating region. */
size_t opusgpu_encode_workspace_bytes(int n_frames);
int opusgpu_encode_batch(const opusgpu_celt_config *cfg, void *d_states, const int16_t *d_pcm,
                         unsigned char *d_out, int out_stride, int32_t *d_out_len, uint32_t *d_out_rng,
                         int n_frames, void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* ---- batched opus_decode(), CELT-only ------------------------------------------------------------------
 * Replaces, for N streams at once, opus_decode() (opus-fix/src/opus_decoder.c:758, include/opus.h:462) ->
 * opus_decode_native -> opus_decode_frame -> celt_decode_with_ec (celt/celt_decoder.c:713) for packets with
 * TOC 0xFC (CELT-only, fullband, 20 ms, stereo, one frame). One call decodes the next packet of every stream:
 *   d_states   device, n x opusgpu_celt_dec_state_size() bytes, initialised once with
 *              opusgpu_celt_dec_state_init (== opus_decoder_create(48000, 2)); advanced by one frame per call.
 *   d_packets  device, packet i at d_packets + i * packet_stride, d_len[i] bytes (as opus_decode's data, len).
 *   d_pcm      device, int16 [n][960][2] interleaved (as opus_decode's pcm with frame_size 960), 16-byte aligned
 *              (it is written 16 bytes at a time; OPUSGPU_BAD_ARG otherwise).
 *   d_ret      device, int32 [n]: 960, or a negative OPUSGPU_* code for that stream (opus_decode's return value):
 *              OPUSGPU_UNIMPLEMENTED for other TOCs and for 1-byte packets (loss concealment / DTX).
 *   d_rng      device, uint32 [n]: OPUS_GET_FINAL_RANGE after the packet (equals the encoder's).
 * Asynchronous on hip_stream. */
int opusgpu_celt_dec_state_size(void);
int opusgpu_celt_dec_state_init(void *d_states, int n_streams, void *hip_stream);
int opusgpu_decode_batch(void *d_states, const unsigned char *d_packets, int packet_stride, const int32_t *d_len,
                         int16_t *d_pcm, int32_t *d_ret, uint32_t *d_rng, int n_streams, void *hip_stream);

/* ---- libopus single-stream encoder API as a batch of one (plumbing; BASELINE config #1) --------------
 * Same verbs, argument meaning and return codes as opus_encoder_create / opus_encoder_ctl / opus_encode /
 * opus_encoder_destroy (opus-fix/include/opus.h:164-263; src/opus_encoder.c:482, :2031, :2007, :2491), host
 * pointers, for 48 kHz stereo OPUS_APPLICATION_RESTRICTED_LOWDELAY and frame_size 960 (anything else that is
 * legal in libopus returns OPUS_UNIMPLEMENTED). ctl requests: the ones src/opus_demo.c:531-543 issues,
 * OPUS_GET_FINAL_RANGE (4031) and OPUS_RESET_STATE (4028). Each call is one opusgpu_encode_batch of a single
 * frame between two small copies: latency-bound by construction. */
typedef struct OpusGpuEncoder OpusGpuEncoder;
OpusGpuEncoder *opusgpu_encoder_create(int32_t Fs, int channels, int application, int *error);
int opusgpu_encoder_ctl(OpusGpuEncoder *st, int request, ...);
int32_t opusgpu_encode(OpusGpuEncoder *st, const int16_t *pcm, int frame_size, unsigned char *data, int32_t max_data_bytes);
void opusgpu_encoder_destroy(OpusGpuEncoder *st);
/* Decoder side, likewise: opus_decoder_create / opus_decode / opus_decoder_ctl / opus_decoder_destroy
 * (opus-fix/include/opus.h:438-505; src/opus_decoder.c:121, :758, :788, :911) for 48 kHz stereo and CELT-only
 * 20 ms stereo packets; ctl: OPUS_GET_FINAL_RANGE, OPUS_RESET_STATE. data == NULL / decode_fec (loss concealment)
 * return OPUS_UNIMPLEMENTED. */
typedef struct OpusGpuDecoder OpusGpuDecoder;
OpusGpuDecoder *opusgpu_decoder_create(int32_t Fs, int channels, int *error);
int opusgpu_decode(OpusGpuDecoder *st, const unsigned char *data, int32_t len, int16_t *pcm, int frame_size, int decode_fec);
int opusgpu_decoder_ctl(OpusGpuDecoder *st, int request, ...);
void opusgpu_decoder_destroy(OpusGpuDecoder *st);

#ifdef __cplusplus
}
#endif
#endif /* OPUSGPU_H */
