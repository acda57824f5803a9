/* oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Batch/threaded driver around the UNMODIFIED compiled reference (oracle/_ref/libopus_ref.so). It
 * contains no codec arithmetic of its own: it only loops the reference's entry points over many
 * frames (optionally on several threads; the library is re-entrant) so that
 *   - tests can generate reference outputs for whole batches in one call, and
 *   - bench.py can time "the reference's own CPU path" beside the GPU (cpu_baseline.kind="reference").
 * Struct prefixes mirror opus-fix/celt/modes.h:52-76 and celt/mdct.h:49-54 (x86-64 layout); the
 * reference headers are not included.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int n; int maxshift; const void *kfft[4]; const int16_t *trig; } ref_mdct_lookup;
typedef struct {
    int32_t Fs; int overlap; int nbEBands; int effEBands; int16_t preemph[4]; const int16_t *eBands;
    int maxLM; int nbShortMdcts; int shortMdctSize; int nbAllocVectors; const unsigned char *allocVectors;
    const int16_t *logN; const int16_t *window; ref_mdct_lookup mdct;
} ref_mode_prefix;

extern const ref_mode_prefix *opus_custom_mode_create(int32_t Fs, int frame_size, int *error);
extern void clt_mdct_forward_c(const ref_mdct_lookup *l, int32_t *in, int32_t *out, const int16_t *window,
                               int overlap, int shift, int stride, int arch);
extern void clt_mdct_backward_c(const ref_mdct_lookup *l, int32_t *in, int32_t *out, const int16_t *window,
                                int overlap, int shift, int stride, int arch);

typedef struct {
    const int32_t *sig_in; int32_t *freq; int32_t *sig_out; long first, count; int shift; int dir;
} mdct_job;

static void *mdct_worker(void *arg)
{
    mdct_job *j = (mdct_job *)arg;
    int err = 0;
    const ref_mode_prefix *m = opus_custom_mode_create(48000, 960, &err);
    int B = 1 << j->shift, n2 = 960 >> j->shift;
    int32_t tmp[1080];
    for (long t = j->first; t < j->first + j->count; t++) {
        if (j->dir & 1) {
            for (int b = 0; b < B; b++) {
                memcpy(tmp, j->sig_in + t * 1080 + b * n2, (size_t)(n2 + 120) * 4);   /* forward trashes in */
                clt_mdct_forward_c(&m->mdct, tmp, j->freq + t * 960 + b, m->window, 120, j->shift, B, 0);
            }
        }
        if (j->dir & 2) {
            for (int b = 0; b < B; b++)
                clt_mdct_backward_c(&m->mdct, j->freq + t * 960 + b, j->sig_out + t * 1080 + b * n2, m->window,
                                    120, j->shift, B, 0);
        }
    }
    return NULL;
}

/* dir bit0: forward sig_in -> freq; bit1: backward freq -> sig_out (in place on sig_out).
 * ntransforms = frames*channels, split contiguously over `threads`. */
void refdrv_mdct_batch(const int32_t *sig_in, int32_t *freq, int32_t *sig_out, long ntransforms, int shift,
                       int dir, int threads)
{
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    mdct_job *jobs = (mdct_job *)malloc(sizeof(mdct_job) * threads);
    long per = (ntransforms + threads - 1) / threads;
    int used = 0;
    for (int i = 0; i < threads; i++) {
        long first = (long)i * per;
        if (first >= ntransforms) break;
        long cnt = ntransforms - first < per ? ntransforms - first : per;
        jobs[i] = (mdct_job){sig_in, freq, sig_out, first, cnt, shift, dir};
        pthread_create(&th[i], NULL, mdct_worker, &jobs[i]);
        used++;
    }
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}

/* ---- opus_encode() batches --------------------------------------------------------------------------
 * Drives the reference's public API exactly as opus_demo does (src/opus_demo.c:519-543):
 * opus_encoder_create(48000, ch, OPUS_APPLICATION_RESTRICTED_LOWDELAY) + the ctl sequence, then
 * opus_encode(enc, pcm, 960, data, max_data_bytes) and OPUS_GET_FINAL_RANGE per frame.
 * Frames are stream-major: frame f of stream s is at index s*frames_per_stream + f; a new encoder is
 * created per stream, so frames_per_stream == 1 gives SURVEY 8d's "independent first frames".
 * Request codes: include/opus_defines.h:130-167. */
typedef struct OpusEncoder OpusEncoder;
extern OpusEncoder *opus_encoder_create(int32_t Fs, int channels, int application, int *error);
extern int opus_encoder_ctl(OpusEncoder *st, int request, ...);
extern int32_t opus_encode(OpusEncoder *st, const int16_t *pcm, int frame_size, unsigned char *data, int32_t max_data_bytes);
extern void opus_encoder_destroy(OpusEncoder *st);

typedef struct {
    int32_t channels, bitrate, vbr, constrained_vbr, complexity, lsb_depth, loss_rate, max_data_bytes;
} refdrv_config;   /* same layout as opusgpu_celt_config */

typedef struct {
    const refdrv_config *cfg; const int16_t *pcm; unsigned char *out; int out_stride; int *out_len; uint32_t *out_rng;
    long first_stream, nstreams; int frames_per_stream;
} enc_job;

static void *enc_worker(void *arg)
{
    enc_job *j = (enc_job *)arg;
    const refdrv_config *c = j->cfg;
    for (long s = j->first_stream; s < j->first_stream + j->nstreams; s++) {
        int err = 0;
        OpusEncoder *enc = opus_encoder_create(48000, c->channels, 2051, &err);
        if (!enc) continue;
        opus_encoder_ctl(enc, 4002, c->bitrate);          /* OPUS_SET_BITRATE */
        opus_encoder_ctl(enc, 4008, -1000);               /* OPUS_SET_BANDWIDTH(OPUS_AUTO) */
        opus_encoder_ctl(enc, 4006, c->vbr);              /* OPUS_SET_VBR */
        opus_encoder_ctl(enc, 4020, c->constrained_vbr);  /* OPUS_SET_VBR_CONSTRAINT */
        opus_encoder_ctl(enc, 4010, c->complexity);       /* OPUS_SET_COMPLEXITY */
        opus_encoder_ctl(enc, 4012, 0);                   /* OPUS_SET_INBAND_FEC */
        opus_encoder_ctl(enc, 4022, -1000);               /* OPUS_SET_FORCE_CHANNELS(OPUS_AUTO) */
        opus_encoder_ctl(enc, 4016, 0);                   /* OPUS_SET_DTX */
        opus_encoder_ctl(enc, 4014, c->loss_rate);        /* OPUS_SET_PACKET_LOSS_PERC */
        opus_encoder_ctl(enc, 4036, c->lsb_depth);        /* OPUS_SET_LSB_DEPTH */
        opus_encoder_ctl(enc, 4040, 5000);                /* OPUS_SET_EXPERT_FRAME_DURATION(OPUS_FRAMESIZE_ARG) */
        for (int f = 0; f < j->frames_per_stream; f++) {
            long n = s * j->frames_per_stream + f;
            j->out_len[n] = opus_encode(enc, j->pcm + n * 960 * c->channels, 960, j->out + n * (long)j->out_stride,
                                        c->max_data_bytes < j->out_stride ? c->max_data_bytes : j->out_stride);
            opus_encoder_ctl(enc, 4031, &j->out_rng[n]);  /* OPUS_GET_FINAL_RANGE */
        }
        opus_encoder_destroy(enc);
    }
    return NULL;
}

void refdrv_encode_frames(const refdrv_config *cfg, const int16_t *pcm, long nframes, int frames_per_stream,
                          unsigned char *out, int out_stride, int *out_len, uint32_t *out_rng, int threads)
{
    long nstreams = nframes / frames_per_stream;
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    enc_job *jobs = (enc_job *)malloc(sizeof(enc_job) * threads);
    long per = (nstreams + threads - 1) / threads;
    int used = 0;
    for (int i = 0; i < threads; i++) {
        long first = (long)i * per;
        if (first >= nstreams) break;
        long cnt = nstreams - first < per ? nstreams - first : per;
        jobs[i] = (enc_job){cfg, pcm, out, out_stride, out_len, out_rng, first, cnt, frames_per_stream};
        pthread_create(&th[i], NULL, enc_worker, &jobs[i]);
        used++;
    }
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}


/* ---- decode side: opus_decoder_create(48000, 2) + opus_decode(960) per packet, streams of frames_per_stream ---- */
typedef struct OpusDecoder OpusDecoder;
extern OpusDecoder *opus_decoder_create(int32_t Fs, int channels, int *error);             /* include/opus.h:438 */
extern int opus_decode(OpusDecoder *st, const unsigned char *data, int32_t len, int16_t *pcm, int frame_size, int decode_fec);
extern int opus_decoder_ctl(OpusDecoder *st, int request, ...);
extern void opus_decoder_destroy(OpusDecoder *st);
#define REF_OPUS_GET_FINAL_RANGE_REQUEST 4031
typedef struct {
    const unsigned char *pk; int stride; const int *len; int16_t *pcm; uint32_t *rng; int *ret;
    long first, count; int fps;
} dec_job;

static void *dec_worker(void *arg)
{
    dec_job *j = (dec_job *)arg;
    for (long s = j->first; s < j->first + j->count; s++) {
        int err = 0;
        OpusDecoder *d = opus_decoder_create(48000, 2, &err);
        for (int f = 0; f < j->fps; f++) {
            long k = s * j->fps + f;
            j->ret[k] = d ? opus_decode(d, j->pk + (size_t)k * j->stride, j->len[k], j->pcm + (size_t)k * 960 * 2, 960, 0) : err;
            if (d) opus_decoder_ctl(d, REF_OPUS_GET_FINAL_RANGE_REQUEST, &j->rng[k]);
        }
        if (d) opus_decoder_destroy(d);
    }
    return NULL;
}

void refdrv_decode_frames(const unsigned char *packets, int stride, const int *len, long nframes, int frames_per_stream,
                          int16_t *pcm, uint32_t *rng, int *ret, int threads)
{
    long nstreams = nframes / frames_per_stream;
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    dec_job *jobs = (dec_job *)malloc(sizeof(dec_job) * threads);
    long per = (nstreams + threads - 1) / threads;
    int used = 0;
    for (int i = 0; i < threads; i++) {
        long first = (long)i * per;
        if (first >= nstreams) break;
        long cnt = nstreams - first < per ? nstreams - first : per;
        jobs[i] = (dec_job){packets, stride, len, pcm, rng, ret, first, cnt, frames_per_stream};
        pthread_create(&th[i], NULL, dec_worker, &jobs[i]);
        used++;
    }
    for (int i = 0; i < used; i++) pthread_join(th[i], NULL);
    free(th);
    free(jobs);
}

/* ---- SILK: opus_encode() as a mono VOIP encoder (the SILK path), `threads` encoders in parallel, each coding the same `nframes`
 * frames of `frame` samples `loops` times over; returns the bytes produced (so that nothing is optimised away). Used by bench.py
 * as the CPU baseline of the SILK streams workload: the reference's whole encoder, timed by the caller. ---- */
typedef struct { const int16_t *pcm; long nframes; int frame, fs, bitrate, vbr, complexity, loops; long bytes; } silk_job;

static void *silk_worker(void *arg)
{
    silk_job *j = (silk_job *)arg;
    int err = 0;
    unsigned char out[1500];
    OpusEncoder *enc = opus_encoder_create(j->fs, 1, 2048, &err);         /* OPUS_APPLICATION_VOIP */
    if (!enc) return NULL;
    opus_encoder_ctl(enc, 4002, j->bitrate);
    opus_encoder_ctl(enc, 4006, j->vbr);
    opus_encoder_ctl(enc, 4020, 0);
    opus_encoder_ctl(enc, 4010, j->complexity);
    opus_encoder_ctl(enc, 4012, 0);
    opus_encoder_ctl(enc, 4016, 0);
    opus_encoder_ctl(enc, 4014, 0);
    opus_encoder_ctl(enc, 4036, 16);
    for (int l = 0; l < j->loops; l++)
        for (long f = 0; f < j->nframes; f++) {
            int n = opus_encode(enc, j->pcm + f * j->frame, j->frame, out, 1500);
            if (n > 0) j->bytes += n;
        }
    opus_encoder_destroy(enc);
    return NULL;
}

long refdrv_silk_encode_loop(const int16_t *pcm, long nframes, int frame, int fs, int bitrate, int vbr, int complexity, int loops, int threads)
{
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    silk_job *jobs = (silk_job *)malloc(sizeof(silk_job) * threads);
    long total = 0;
    for (int i = 0; i < threads; i++) {
        jobs[i] = (silk_job){pcm, nframes, frame, fs, bitrate, vbr, complexity, loops, 0};
        pthread_create(&th[i], NULL, silk_worker, &jobs[i]);
    }
    for (int i = 0; i < threads; i++) { pthread_join(th[i], NULL); total += jobs[i].bytes; }
    free(th);
    free(jobs);
    return total;
}
