/* oracle/oracle_mdct.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped, never timed as
 * the product).
 *
 * Restates, for the single static mode 48000/960 (overlap 120, window120, FIXED_POINT):
 *   opus_fft_impl + kf_bfly2/4/3/5   opus-fix/celt/kiss_fft.c:532-578, :51-109, :111-177, :185-241, :245-322
 *   clt_mdct_forward_c               opus-fix/celt/mdct.c:121-259
 *   clt_mdct_backward_c              opus-fix/celt/mdct.c:263-363
 * The per-element arithmetic (operand order of every add/sub/S_MUL) follows the reference so the
 * results are bit-identical; loops are restated over explicit indices instead of walking pointers.
 * Pinned against the compiled reference in tests/test_oracle_mdct.py.
 */
#include <string.h>
#include "oracle_arith.h"
#include "oracle_tables.h"

typedef struct { i32 r, i; } cpx;

#define SMUL(x, t) mul16_32_q15((i16)(t), (x))   /* _kiss_fft_guts.h:57  S_MUL(a,b)=MULT16_32_Q15(b,a) */

static inline cpx c_mul(cpx a, i16 tr, i16 ti)       /* _kiss_fft_guts.h:59 C_MUL */
{
    cpx m;
    m.r = SMUL(a.r, tr) - SMUL(a.i, ti);
    m.i = SMUL(a.r, ti) + SMUL(a.i, tr);
    return m;
}
static inline cpx c_add(cpx a, cpx b) { cpx m = { a.r + b.r, a.i + b.i }; return m; }
static inline cpx c_sub(cpx a, cpx b) { cpx m = { a.r - b.r, a.i - b.i }; return m; }

/* Execution-order stage plans (radix, m) for nfft = 480 >> shift; derived from the factor lists
 * static_modes_fixed.h:432-500 read back to front as opus_fft_impl does (kiss_fft.c:554-577). */
static const int PLAN[4][5][2] = {
    { {4, 1}, {2, 4}, {4, 8}, {3, 32}, {5, 96} },
    { {4, 1}, {4, 4}, {3, 16}, {5, 48}, {0, 0} },
    { {4, 1}, {2, 4}, {3, 8}, {5, 24}, {0, 0} },
    { {4, 1}, {3, 4}, {5, 12}, {0, 0}, {0, 0} },
};

static void radix4_first(cpx *x, int groups)          /* kiss_fft.c:123-145 (m==1, unit twiddles) */
{
    for (int g = 0; g < groups; g++, x += 4) {
        cpx s0 = c_sub(x[0], x[2]);
        x[0] = c_add(x[0], x[2]);
        cpx s1 = c_add(x[1], x[3]);
        x[2] = c_sub(x[0], s1);
        x[0] = c_add(x[0], s1);
        s1 = c_sub(x[1], x[3]);
        x[1].r = s0.r + s1.i;  x[1].i = s0.i - s1.r;
        x[3].r = s0.r - s1.i;  x[3].i = s0.i + s1.r;
    }
}

static void radix2_m4(cpx *x, int groups)             /* kiss_fft.c:72-107 (m==4) */
{
    const i16 tw = 23170;                              /* QCONST16(0.7071067812f,15) */
    for (int g = 0; g < groups; g++, x += 8) {
        cpx t;
        t = x[4];
        x[4] = c_sub(x[0], t);  x[0] = c_add(x[0], t);
        t.r = SMUL(x[5].r + x[5].i, tw);
        t.i = SMUL(x[5].i - x[5].r, tw);
        x[5] = c_sub(x[1], t);  x[1] = c_add(x[1], t);
        t.r = x[6].i;  t.i = -x[6].r;
        x[6] = c_sub(x[2], t);  x[2] = c_add(x[2], t);
        t.r = SMUL(x[7].i - x[7].r, tw);
        t.i = SMUL(-x[7].i - x[7].r, tw);
        x[7] = c_sub(x[3], t);  x[3] = c_add(x[3], t);
    }
}

static void radix4(cpx *x, int m, int groups, int twstride)   /* kiss_fft.c:146-176 */
{
    for (int g = 0; g < groups; g++) {
        cpx *f = x + g * 4 * m;
        for (int j = 0; j < m; j++, f++) {
            const i16 *t1 = CLT_fft_twiddles480 + 2 * (j * twstride);
            const i16 *t2 = CLT_fft_twiddles480 + 2 * (2 * j * twstride);
            const i16 *t3 = CLT_fft_twiddles480 + 2 * (3 * j * twstride);
            cpx a = c_mul(f[m], t1[0], t1[1]);
            cpx b = c_mul(f[2 * m], t2[0], t2[1]);
            cpx c = c_mul(f[3 * m], t3[0], t3[1]);
            cpx d5 = c_sub(f[0], b);
            f[0] = c_add(f[0], b);
            cpx s3 = c_add(a, c);
            cpx s4 = c_sub(a, c);
            f[2 * m] = c_sub(f[0], s3);
            f[0] = c_add(f[0], s3);
            f[m].r = d5.r + s4.i;      f[m].i = d5.i - s4.r;
            f[3 * m].r = d5.r - s4.i;  f[3 * m].i = d5.i + s4.r;
        }
    }
}

static void radix3(cpx *x, int m, int groups, int twstride)   /* kiss_fft.c:185-241 */
{
    const i16 epi3_i = -28378;
    for (int g = 0; g < groups; g++) {
        cpx *f = x + g * 3 * m;
        for (int j = 0; j < m; j++, f++) {
            const i16 *t1 = CLT_fft_twiddles480 + 2 * (j * twstride);
            const i16 *t2 = CLT_fft_twiddles480 + 2 * (2 * j * twstride);
            cpx a = c_mul(f[m], t1[0], t1[1]);
            cpx b = c_mul(f[2 * m], t2[0], t2[1]);
            cpx s3 = c_add(a, b);
            cpx s0 = c_sub(a, b);
            f[m].r = f[0].r - (s3.r >> 1);
            f[m].i = f[0].i - (s3.i >> 1);
            s0.r = SMUL(s0.r, epi3_i);
            s0.i = SMUL(s0.i, epi3_i);
            f[0] = c_add(f[0], s3);
            f[2 * m].r = f[m].r + s0.i;
            f[2 * m].i = f[m].i - s0.r;
            f[m].r -= s0.i;
            f[m].i += s0.r;
        }
    }
}

static void radix5(cpx *x, int m, int groups, int twstride)   /* kiss_fft.c:245-322 */
{
    const i16 ya_r = 10126, ya_i = -31164, yb_r = -26510, yb_i = -19261;
    for (int g = 0; g < groups; g++) {
        cpx *f = x + g * 5 * m;
        for (int u = 0; u < m; u++, f++) {
            const i16 *tw = CLT_fft_twiddles480;
            cpx s0 = f[0];
            cpx s1 = c_mul(f[m],     tw[2 * (u * twstride)],     tw[2 * (u * twstride) + 1]);
            cpx s2 = c_mul(f[2 * m], tw[2 * (2 * u * twstride)], tw[2 * (2 * u * twstride) + 1]);
            cpx s3 = c_mul(f[3 * m], tw[2 * (3 * u * twstride)], tw[2 * (3 * u * twstride) + 1]);
            cpx s4 = c_mul(f[4 * m], tw[2 * (4 * u * twstride)], tw[2 * (4 * u * twstride) + 1]);
            cpx s7 = c_add(s1, s4), s10 = c_sub(s1, s4);
            cpx s8 = c_add(s2, s3), s9 = c_sub(s2, s3);
            f[0].r += s7.r + s8.r;
            f[0].i += s7.i + s8.i;
            cpx s5, s6, s11, s12;
            s5.r = s0.r + SMUL(s7.r, ya_r) + SMUL(s8.r, yb_r);
            s5.i = s0.i + SMUL(s7.i, ya_r) + SMUL(s8.i, yb_r);
            s6.r = SMUL(s10.i, ya_i) + SMUL(s9.i, yb_i);
            s6.i = -SMUL(s10.r, ya_i) - SMUL(s9.r, yb_i);
            f[m] = c_sub(s5, s6);
            f[4 * m] = c_add(s5, s6);
            s11.r = s0.r + SMUL(s7.r, yb_r) + SMUL(s8.r, ya_r);
            s11.i = s0.i + SMUL(s7.i, yb_r) + SMUL(s8.i, ya_r);
            s12.r = -SMUL(s10.i, yb_i) + SMUL(s9.i, ya_i);
            s12.i = SMUL(s10.r, yb_i) - SMUL(s9.r, ya_i);
            f[2 * m] = c_add(s11, s12);
            f[3 * m] = c_sub(s11, s12);
        }
    }
}

/* In-place FFT over bit-reversed input, nfft = 480 >> shift (kiss_fft.c:532-578). */
void orc_fft_inplace(i32 *data, int shift)
{
    cpx *x = (cpx *)data;
    int nfft = 480 >> shift;
    for (int s = 0; s < 5 && PLAN[shift][s][0]; s++) {
        int p = PLAN[shift][s][0], m = PLAN[shift][s][1];
        int groups = nfft / (p * m);
        int twstride = groups << shift;               /* fstride[i] << st->shift */
        if (p == 4 && m == 1) radix4_first(x, groups);
        else if (p == 2) radix2_m4(x, groups);
        else if (p == 4) radix4(x, m, groups, twstride);
        else if (p == 3) radix3(x, m, groups, twstride);
        else radix5(x, m, groups, twstride);
    }
}

static const i16 *bitrev_for(int shift)
{
    switch (shift) {
    case 0: return CLT_fft_bitrev480;
    case 1: return CLT_fft_bitrev240;
    case 2: return CLT_fft_bitrev120;
    default: return CLT_fft_bitrev60;
    }
}

static const i16 *trig_for(int shift)
{
    const i16 *t = CLT_mdct_trig960;
    int n = 1920;
    for (int i = 0; i < shift; i++) { n >>= 1; t += n; }   /* mdct.c:141-146 */
    return t;
}

/* opus_fft_c (kiss_fft.c:580-599): scaled, bit-reversing out-of-place FFT. */
void orc_fft(const i32 *fin, i32 *fout, int shift)
{
    int nfft = 480 >> shift;
    const i16 *br = bitrev_for(shift);
    int scale_shift = (8 - shift) - 1;
    for (int i = 0; i < nfft; i++) {
        fout[2 * br[i]]     = mul16_32_q16(17476, fin[2 * i]) >> scale_shift;
        fout[2 * br[i] + 1] = mul16_32_q16(17476, fin[2 * i + 1]) >> scale_shift;
    }
    orc_fft_inplace(fout, shift);
}

/* clt_mdct_forward_c (mdct.c:121-259). `in` holds N/2+overlap samples, `out` receives N/2
 * coefficients at out[k*stride]. N = 1920 >> shift. Unlike the reference this does not trash `in`. */
void orc_mdct_forward(const i32 *in, i32 *out, int shift, int stride)
{
    enum { OV = 120, OV2 = 60, Q = 30 };
    const int N2 = 960 >> shift, N4 = N2 >> 1;
    const i16 *w = CLT_window120;
    const i16 *trig = trig_for(shift);
    const i16 *br = bitrev_for(shift);
    const int scale_shift = (8 - shift) - 1;              /* st->scale_shift-1 (mdct.c:134) */
    i32 f2[2 * 480];

    for (int i = 0; i < N4; i++) {
        i32 re, im;
        int a = OV2 + 2 * i;            /* xp1 index */
        int b = N2 - 1 + OV2 - 2 * i;   /* xp2 index */
        if (i < Q) {                                              /* mdct.c:162-175 */
            i16 w1 = w[OV2 + 2 * i], w2 = w[OV2 - 1 - 2 * i];
            re = mul16_32_q15(w2, in[a + N2]) + mul16_32_q15(w1, in[b]);
            im = mul16_32_q15(w1, in[a]) - mul16_32_q15(w2, in[b - N2]);
        } else if (i < N4 - Q) {                                  /* mdct.c:178-189 */
            re = in[b];
            im = in[a];
        } else {                                                  /* mdct.c:190-203 */
            int k = i - (N4 - Q);
            i16 w1 = w[2 * k], w2 = w[OV - 1 - 2 * k];
            re = -mul16_32_q15(w1, in[a - N2]) + mul16_32_q15(w2, in[b]);
            im = mul16_32_q15(w2, in[a]) + mul16_32_q15(w1, in[b + N2]);
        }
        /* pre-rotation + scaling, scattered to bit-reversed order (mdct.c:206-231) */
        i16 t0 = trig[i], t1 = trig[N4 + i];
        i32 yr = SMUL(re, t0) - SMUL(im, t1);
        i32 yi = SMUL(im, t0) + SMUL(re, t1);
        f2[2 * br[i]]     = pshr32(mul16_32_q16(17476, yr), scale_shift);
        f2[2 * br[i] + 1] = pshr32(mul16_32_q16(17476, yi), scale_shift);
    }
    orc_fft_inplace(f2, shift);
    for (int i = 0; i < N4; i++) {                                /* mdct.c:237-257 */
        i32 fr = f2[2 * i], fi = f2[2 * i + 1];
        i16 t0 = trig[i], t1 = trig[N4 + i];
        out[(2 * i) * stride]          = SMUL(fi, t1) - SMUL(fr, t0);
        out[(N2 - 1 - 2 * i) * stride] = SMUL(fr, t1) + SMUL(fi, t0);
    }
}

/* clt_mdct_backward_c (mdct.c:263-363). `in` has N/2 coefficients at in[k*stride]; `out` has
 * N/2+overlap samples: out[0..overlap) is read (previous tail) and mixed, the rest overwritten. */
void orc_mdct_backward(const i32 *in, i32 *out, int shift, int stride)
{
    enum { OV = 120, OV2 = 60 };
    const int N2 = 960 >> shift, N4 = N2 >> 1;
    const i16 *w = CLT_window120;
    const i16 *trig = trig_for(shift);
    const i16 *br = bitrev_for(shift);
    i32 *y = out + OV2;

    for (int i = 0; i < N4; i++) {                                /* mdct.c:283-304 */
        i32 x1 = in[(2 * i) * stride], x2 = in[(N2 - 1 - 2 * i) * stride];
        i16 t0 = trig[i], t1 = trig[N4 + i];
        i32 yr = SMUL(x2, t0) + SMUL(x1, t1);
        i32 yi = SMUL(x1, t0) - SMUL(x2, t1);
        y[2 * br[i] + 1] = yr;
        y[2 * br[i]] = yi;
    }
    orc_fft_inplace(y, shift);
    for (int i = 0; i < (N4 + 1) >> 1; i++) {                     /* mdct.c:310-342 */
        i32 *p0 = y + 2 * i, *p1 = y + N2 - 2 - 2 * i;
        i32 re = p0[1], im = p0[0];
        i16 t0 = trig[i], t1 = trig[N4 + i];
        i32 yr = SMUL(re, t0) + SMUL(im, t1);
        i32 yi = SMUL(re, t1) - SMUL(im, t0);
        re = p1[1];  im = p1[0];
        p0[0] = yr;
        p1[1] = yi;
        t0 = trig[N4 - i - 1];  t1 = trig[N2 - i - 1];
        yr = SMUL(re, t0) + SMUL(im, t1);
        yi = SMUL(re, t1) - SMUL(im, t0);
        p1[0] = yr;
        p0[1] = yi;
    }
    for (int i = 0; i < OV2; i++) {                               /* mdct.c:345-361 */
        i32 x1 = out[OV - 1 - i], x2 = out[i];
        i16 w1 = w[i], w2 = w[OV - 1 - i];
        out[i]          = mul16_32_q15(w2, x2) - mul16_32_q15(w1, x1);
        out[OV - 1 - i] = mul16_32_q15(w1, x2) + mul16_32_q15(w2, x1);
    }
}

/* Batch drivers used by tests and by bench.py's cpu_baseline leg ("port" kind).
 * Layout = the product's: sig[frame][ch][1080] -> freq[frame][ch][960] (shift 0) or, for
 * shift 3, 8 short blocks with hop 120 written interleaved (stride 8), as compute_mdcts does
 * (celt_encoder.c:418-461). */
void orc_mdct_forward_batch(const i32 *sig, i32 *freq, int nframes, int channels, int shift)
{
    int B = 1 << shift, N2 = 960 >> shift;
    for (long fc = 0; fc < (long)nframes * channels; fc++)
        for (int b = 0; b < B; b++)
            orc_mdct_forward(sig + fc * 1080 + b * N2, freq + fc * 960 + b, shift, B);
}

/* Inverse of the above as celt_synthesis does it (celt_decoder.c:323-346): out[frame][ch][1080]
 * with out[0..120) holding the previous frame's overlap tail on entry. */
void orc_mdct_backward_batch(const i32 *freq, i32 *sig, int nframes, int channels, int shift)
{
    int B = 1 << shift, N2 = 960 >> shift;
    for (long fc = 0; fc < (long)nframes * channels; fc++)
        for (int b = 0; b < B; b++)
            orc_mdct_backward(freq + fc * 960 + b, sig + fc * 1080 + b * N2, shift, B);
}
