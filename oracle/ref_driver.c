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
