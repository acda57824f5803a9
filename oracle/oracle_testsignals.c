/* oracle/oracle_testsignals.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Restatement of the synthetic "music" the reference's own encoder test feeds opus_encode()
 * (opus-fix/tests/test_opus_encode.c:59-88 generate_music, driven by the multiply-with-carry generator of
 * tests/test_opus_common.h:55-62 seeded with Rz = Rw = 13371337, test_opus_encode.c:552-554): a bytebeat melody
 * (the (j*((j>>12)^((j>>10|j>>12)&26&j>>7)))&128 square wave, j advancing every 6th sample) plus noise, through a
 * DC-blocking differentiator/leaky integrator and a two-tap low-pass, per channel; interleaved stereo int16.
 * SURVEY.md 8d names it as config #3's band-limited input variant. Pinned against the reference's own function by
 * tests/test_oracle_signals.py (oracle/_ref/librefgen.so compiles the reference's test source in place). */
#include <stdint.h>

typedef struct { uint32_t z, w; } orc_mwc;

static uint32_t mwc_next(orc_mwc *g)
{
    g->z = 36969u * (g->z & 65535u) + (g->z >> 16);
    g->w = 18000u * (g->w & 65535u) + (g->w >> 16);
    return (g->z << 16) + g->w;
}

static int16_t clip16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v)); }

/* buf: len stereo sample pairs. seed: both generator words (13371337 in the reference's test); skip: generator draws
 * consumed before the first sample (the test's banner line draws one, test_opus_encode.c:558). */
void orc_generate_music(int16_t *buf, int32_t len, uint32_t seed, int skip)
{
    orc_mwc g = {seed, seed};
    int32_t a1 = 0, b1 = 0, a2 = 0, b2 = 0, c1 = 0, c2 = 0, d1 = 0, d2 = 0, j = 0;
    while (skip-- > 0) (void)mwc_next(&g);
    for (int32_t i = 0; i < len; i++) {
        uint32_t r;
        int32_t v1, v2;
        v1 = v2 = (int32_t)(((uint32_t)((j * ((j >> 12) ^ ((j >> 10 | j >> 12) & 26 & j >> 7))) & 128) + 128) << 15);
        r = mwc_next(&g); v1 += (int32_t)(r & 65535u); v1 -= (int32_t)(r >> 16);
        r = mwc_next(&g); v2 += (int32_t)(r & 65535u); v2 -= (int32_t)(r >> 16);
        b1 = v1 - a1 + ((b1 * 61 + 32) >> 6); a1 = v1;
        b2 = v2 - a2 + ((b2 * 61 + 32) >> 6); a2 = v2;
        c1 = (30 * (c1 + b1 + d1) + 32) >> 6; d1 = b1;
        c2 = (30 * (c2 + b2 + d2) + 32) >> 6; d2 = b2;
        buf[i * 2] = clip16((c1 + 128) >> 8);
        buf[i * 2 + 1] = clip16((c2 + 128) >> 8);
        if (i % 6 == 0) j++;
    }
}
