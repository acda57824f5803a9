// celt_dec.h -- one Opus CELT-only packet -> 960 stereo samples, by ONE LANE (or the host emulation).
//
// Restates celt_decode_with_ec (opus-fix/celt/celt_decoder.c:713-1060) and what it calls for 48 kHz, 20 ms,
// stereo, fullband, start 0 / end 21: the range decoder (rangedec.h), unquant_coarse/fine_energy and
// unquant_energy_finalise (quant_bands.c:434-539), tf_decode (celt_decoder.c:352-392), the dynalloc /
// trim / allocation side information (compute_allocation shared with the encoder), quant_all_bands with
// encode = 0 (bands.c:1337-1502: folding, noise fill, collapse masks, stereo merge), alg_unquant /
// decode_pulses (vq.c:329, cwrs.c:462-530), anti_collapse (bands.c:241), denormalise_bands (bands.c:169),
// celt_synthesis (celt_decoder.c:287), the post-filter comb_filter (celt.c:183) and deemphasis
// (celt_decoder.c:183), plus the CELT-only slice of opus_decode_frame (src/opus_decoder.c:200-600:
// TOC, range decoder set-up, final range). The decoder is a serial chain per packet, so it is written as
// scalar code for the lane-per-frame build (LANES == 1); packet loss concealment is not implemented.
#pragma once
#include "celt_enc_back.h"
#include "celt_state.h"

namespace ca {

enum { DEC_BUF = 2048, DEC_MEM = DEC_BUF + OVL };           // DECODE_BUFFER_SIZE + overlap per channel

// per-packet working set (host memory in the emulation, private/global memory on the GPU)
struct __attribute__((aligned(16))) DecWork {
    // Both rows live in the stream's state record in HBM and are typed as such (wave.h x16_t: global address space in the lane
    // build): every pointer into X / norm -- band, lowband, lowband_out, lowband scratch -- is then a GLOBAL pointer and the band
    // functions issue global_load / global_store. With norm[] in this (private) struct the same functions were handed pointers into
    // either space, i.e. generic ones: 3 200 FLAT instructions in the kernel, each waiting on both memory counters.
    x16_t *X;                      // -> decoded normalised bands X[c*960 + j] (opusgpu_celt_dec_state::mid_X)
    x16_t *norm;                   // -> folding source: norm / norm2 (bands.c:1369-1372), M*eBands[20] = 624 per channel (::mid_norm)
#if defined(CA_LANE_FRAME)
    // (address space in the pointer type, wave.h: a generic pointer reloaded from this private struct makes every access FLAT)
    CA_AS_LDS i16 *lds_iy16;       // -> this lane's column of the workgroup's LDS pulse vector ([element][lane], LANE_IY16_N bins)
    CA_AS_LDS i16 *lds_pvq16;      // -> ... of the 16-bit scratch (LANE_SCRATCH_N bins: (de)interleave, band staging)
#endif
    i32 iy[176];
    i16 tmp[176];
    i32 offsets[NB], cap[NB], pulses[NB], fine_quant[NB], fine_priority[NB], tf_res[NB];
    i32 bits1[NB], bits2[NB], thresh[NB], trim_offset[NB];
    u8 collapse_masks[2 * NB];
    void *diag;
};

// working set of the synthesis (one wavefront per stream: LDS; host memory in the emulation)
struct __attribute__((aligned(16))) SynthLds {
    i32 freq[FRAME];
    int2 f2[480];
    // denormalise_bands: per band gain / shift of the channel in hand, and the band of every group of eight bins (120 groups)
    i16 dn_gain[NB];
    i8 dn_shift[NB];
    u8 dn_band8[FRAME / 8];
};

CA_DEV u32 celt_lcg_rand(u32 seed) { return 1664525u * seed + 1013904223u; }                   // bands.c:63

// ---- energies ----------------------------------------------------------------------------------------
CA_DEV void unquant_coarse_energy_dec(i16 *oldE, int intra, RangeDec &dec, int C)               // quant_bands.c:434
{
    const u8 *prob_model = CLT_e_prob_model + (LM3 * 2 + intra) * 42;
    i32 prev[2] = {0, 0};
    i32 coef, beta;
    if (intra) { coef = 0; beta = 4915; }                                                      // beta_intra
    else { beta = CLT_beta_coef[LM3]; coef = CLT_pred_coef[LM3]; }
    const i32 budget = (i32)dec.storage * 8;
    for (int i = 0; i < NB; i++) {
        for (int c = 0; c < C; c++) {
            int qi;
            i32 tell = ec_tell(dec);
            if (budget - tell >= 15) {
                int pi = 2 * imin(i, 20);
                qi = ec_laplace_decode(dec, (u32)prob_model[pi] << 7, prob_model[pi + 1] << 6);
            } else if (budget - tell >= 2) {
                qi = ec_dec_icdf(dec, CLT_small_energy_icdf, 2);
                qi = (qi >> 1) ^ -(qi & 1);
            } else if (budget - tell >= 1) {
                qi = -ec_dec_bit_logp(dec, 1);
            } else {
                qi = -1;
            }
            i32 q = shl32(qi, 10);
            oldE[i + c * NB] = (i16)imax(-9216, oldE[i + c * NB]);                             // -QCONST16(9.f, DB_SHIFT)
            i32 tmp = add32(add32(pshr32(mul16_16(coef, oldE[i + c * NB]), 8), prev[c]), shl32(q, 7));
            tmp = imax(-3670016, tmp);                                                         // -QCONST32(28.f, DB_SHIFT+7)
            oldE[i + c * NB] = (i16)pshr32(tmp, 7);
            prev[c] = sub32(add32(prev[c], shl32(q, 7)), mul16_16(beta, pshr32(q, 8)));
        }
    }
}

CA_DEV void unquant_fine_energy_dec(i16 *oldE, const i32 *fine_quant, RangeDec &dec, int C)     // quant_bands.c:490
{
    for (int i = 0; i < NB; i++) {
        if (fine_quant[i] <= 0) continue;
        for (int c = 0; c < C; c++) {
            i32 q2 = (i32)ec_dec_bits(dec, (u32)fine_quant[i]);
            i32 offset = sub16((shl32(q2, 10) + 512) >> fine_quant[i], 512);
            oldE[i + c * NB] = (i16)(oldE[i + c * NB] + offset);
        }
    }
}

CA_DEV void unquant_energy_finalise_dec(i16 *oldE, const i32 *fine_quant, const i32 *fine_priority, int bits_left,
                                        RangeDec &dec, int C)                                  // quant_bands.c:513
{
    for (int prio = 0; prio < 2; prio++) {
        for (int i = 0; i < NB && bits_left >= C; i++) {
            if (fine_quant[i] >= MAX_FINE_BITS || fine_priority[i] != prio) continue;
            for (int c = 0; c < C; c++) {
                i32 q2 = (i32)ec_dec_bits(dec, 1);
                i32 offset = (i16)((shl16(q2, 10) - 512) >> (fine_quant[i] + 1));
                oldE[i + c * NB] = (i16)(oldE[i + c * NB] + offset);
                bits_left--;
            }
        }
    }
}

CA_DEV void tf_decode_dec(int isTransient, i32 *tf_res, RangeDec &dec)                          // celt_decoder.c:352
{
    const int LM = LM3;
    u32 budget = dec.storage * 8;
    u32 tell = (u32)ec_tell(dec);
    int logp = isTransient ? 2 : 4;
    int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= tf_select_rsv;
    int tf_changed = 0, curr = 0;
    for (int i = 0; i < NB; i++) {
        if (tell + logp <= budget) {
            curr ^= ec_dec_bit_logp(dec, (u32)logp);
            tell = (u32)ec_tell(dec);
            tf_changed |= curr;
        }
        tf_res[i] = curr;
        logp = isTransient ? 4 : 5;
    }
    int tf_select = 0;
    const i8 *tab = CLT_tf_select_table + LM * 8;
    if (tf_select_rsv && tab[4 * isTransient + 0 + tf_changed] != tab[4 * isTransient + 2 + tf_changed])
        tf_select = ec_dec_bit_logp(dec, 1);
    for (int i = 0; i < NB; i++) tf_res[i] = tab[4 * isTransient + 2 * tf_select + tf_res[i]];
}

// ---- PVQ decode ----------------------------------------------------------------------------------------
CA_DEV void exp_rotation1_ref(x16_t *X, int len, int stride, i32 c, i32 s)                        // vq.c:43-68
{
    const i32 ms = (i16)neg32(s);
    x16_t *p = X;
    for (int i = 0; i < len - stride; i++) {
        i32 x1 = p[0], x2 = p[stride];
        p[stride] = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
        *p++ = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
    }
    p = &X[len - 2 * stride - 1];
    for (int i = len - 2 * stride - 1; i >= 0; i--) {
        i32 x1 = p[0], x2 = p[stride];
        p[stride] = (i16)pshr32(mac16_16(mul16_16(c, x2), s, x1), 15);
        *p-- = (i16)pshr32(mac16_16(mul16_16(c, x1), ms, x2), 15);
    }
}

template <class P>
CA_DEV void exp_rotation_inv(P X, int len, int stride, int K, int spread)                       // vq.c:70-117, dir = -1
{
    if (2 * K >= len || spread == SPREAD_NONE) return;
    const int factor = spread == SPREAD_LIGHT ? 15 : spread == SPREAD_NORMAL ? 10 : 5;
    i32 gain = (i16)celt_div(mul16_16(32767, len), len + factor * K);
    i32 theta = (i16)mul16_16_q15(gain, gain) >> 1;
    i32 c = celt_cos_norm(theta);
    i32 s = celt_cos_norm((i16)sub16(32767, theta));
    int stride2 = 0;
    if (len >= 8 * stride) {
        stride2 = 1;
        while ((stride2 * stride2 + stride2) * stride + (stride >> 2) < len) stride2++;
    }
    // exp_rotation1 per block, as register-carried chains with look-ahead loads (the encoder's formulation,
    // celt_enc_back.h): a literal exp_rotation1 reads back what it stored one step earlier, a round trip through
    // memory per coefficient
    const int blen = (int)((u32)len / (u32)stride);
    if (stride2) exp_rotation1_chains(X, blen, stride, stride2, s, c);
    exp_rotation1_chains(X, blen, stride, 1, c, s);
}

// cwrsi (cwrs.c:462-524): index -> pulse vector, returns sum y^2. (Its two linear searches fetched four candidates per trip -- the
// rows, then the U values, two round trips for four steps instead of two per step -- measured slower, 3.07 -> 3.18 ms: a
// position holds 0.7 pulses on average, so three of the four fetches are wasted issue, and the second wavefront of the SIMD
// already hides the round trips.)
template <class PY>
CA_DEV i32 cwrsi_dec(int n, int k, u32 i, PY y)
{
    i32 yy = 0;
    int pos = 0;
    while (n > 2) {
        u32 p, q;
        int s, k0;
        i32 val;
        if (k >= n) {
            p = pvq_u(n, k + 1);
            s = -(int)(i >= p);
            i -= p & (u32)s;
            k0 = k;
            q = pvq_u(n, n);
            if (q > i) {
                k = n;
                do p = pvq_u(--k, n);
                while (p > i);
            } else {
                for (p = pvq_u(n, k); p > i; p = pvq_u(n, k)) k--;
            }
            i -= p;
            val = (k0 - k + s) ^ s;
            y[pos++] = val;
            yy = mac16_16(yy, val, val);
        } else {
            p = pvq_u(k, n);
            q = pvq_u(k + 1, n);
            if (p <= i && i < q) {
                i -= p;
                y[pos++] = 0;
            } else {
                s = -(int)(i >= q);
                i -= q & (u32)s;
                k0 = k;
                do p = pvq_u(--k, n);
                while (p > i);
                i -= p;
                val = (k0 - k + s) ^ s;
                y[pos++] = val;
                yy = mac16_16(yy, val, val);
            }
        }
        n--;
    }
    {   // n == 2
        u32 p = 2 * (u32)k + 1;
        int s = -(int)(i >= p);
        i -= p & (u32)s;
        int k0 = k;
        k = (int)((i + 1) >> 1);
        if (k) i -= 2 * (u32)k - 1;
        i32 val = (k0 - k + s) ^ s;
        y[pos++] = val;
        yy = mac16_16(yy, val, val);
        // n == 1
        s = -(int)i;
        val = (k + s) ^ s;
        y[pos] = val;
        yy = mac16_16(yy, val, val);
    }
    return yy;
}

CA_DEV void renormalise_vector_dec(x16_t *X, int N, i32 gain)                                     // vq.c:349-374
{
    i32 E = 1;
#pragma unroll 8
    for (int i = 0; i < N; i++) E = mac16_16(E, X[i], X[i]);
    int k = celt_ilog2(E) >> 1;
    i32 t = vshr32(E, 2 * (k - 7));
    i32 g = (i16)mul16_16_p15(celt_rsqrt_norm(t), gain);
    for (int i = 0; i < N; i += 8) {                 // eight loads in flight before the first store
        i32 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = i + u < N ? (i32)X[i + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i + u < N) X[i + u] = (i16)pshr32(mul16_16(g, v[u]), k + 1);
    }
}

template <class D, class PI>
CA_DEV unsigned alg_unquant_body(D &F, x16_t *X, int N, int K, int spread, int B, RangeDec &dec, i32 gain, PI iy);

template <class D>
CA_DEV unsigned alg_unquant_dec(D &F, x16_t *X, int N, int K, int spread, int B, RangeDec &dec, i32 gain)   // vq.c:329-346
{
#if defined(CA_LANE_FRAME)
    // the pulse vector of a leaf of up to 64 bins lives in this lane's LDS column, of a larger one in private memory: two
    // instantiations, each with the address space in its accessor type (and each run by the whole wavefront when its lanes
    // disagree: 16-bit counts so that the unsplit 64-bin band takes the first one too)
    if (N <= LANE_IY16_N) return alg_unquant_body(F, X, N, K, spread, B, dec, gain, lds_col(F.lds_iy16));
    return alg_unquant_body(F, X, N, K, spread, B, dec, gain, priv((i32 *)F.iy));
#else
    return alg_unquant_body(F, X, N, K, spread, B, dec, gain, (i32 *)F.iy);
#endif
}

template <class D, class PI>
CA_DEV unsigned alg_unquant_body(D &F, x16_t *X, int N, int K, int spread, int B, RangeDec &dec, i32 gain, PI iy)
{
    CA_STAMP_F(F, 8);
    u32 V = pvq_u(N, K) + pvq_u(N, K + 1);
    u32 idx = ec_dec_uint(dec, V);
    CA_STAMP_F(F, 3);
    i32 Ryy = cwrsi_dec(N, K, idx, iy);
    CA_STAMP_F(F, 4);
    {   // normalise_residual (vq.c:117-138)
        int k = celt_ilog2(Ryy) >> 1;
        i32 t = vshr32(Ryy, 2 * (k - 7));
        i32 g = (i16)mul16_16_p15(celt_rsqrt_norm(t), gain);
#if defined(CA_LANE_FRAME)
        if (N <= LANE_SCRATCH_N) {
            // the leaf is scaled and un-rotated in the per-lane LDS scratch (16-bit part; iy sits in the 32-bit part) and
            // reaches X in HBM once, sixteen bytes at a time where the leaf is aligned
            LdsCol<i16> T = lds_col(F.lds_pvq16);
#pragma unroll 8
            for (int i = 0; i < N; i++) T[i] = (i16)pshr32(mul16_16(g, iy[i]), k + 1);
            CA_STAMP_F(F, 5);
            exp_rotation_inv(T, N, B, K, spread);
            if ((N & 7) == 0 && ((uintptr_t)X & 15) == 0) {
                for (int i = 0; i < N; i += 8) {
                    u32 h[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) h[u] = (u32)(i32)T[i + u] & 0xffffu;
                    v4i w;
                    w.x = (i32)(h[0] | (h[1] << 16)); w.y = (i32)(h[2] | (h[3] << 16));
                    w.z = (i32)(h[4] | (h[5] << 16)); w.w = (i32)(h[6] | (h[7] << 16));
                    *reinterpret_cast<CA_AS_GLB v4i *>(X + i) = w;
                }
            } else {
#pragma unroll 8
                for (int i = 0; i < N; i++) X[i] = T[i];
            }
        } else
#endif
        {
#pragma unroll 8
            for (int i = 0; i < N; i++) X[i] = (i16)pshr32(mul16_16(g, iy[i]), k + 1);
            CA_STAMP_F(F, 5);
            exp_rotation_inv(X, N, B, K, spread);
        }
    }
    CA_STAMP_F(F, 6);
    if (B <= 1) return 1;                                                                     // extract_collapse_mask
    const int N0 = (int)((u32)N / (u32)B);
    unsigned mask = 0;
    for (int i = 0; i < B; i++) {
        i32 t = 0;
        for (int j = 0; j < N0; j++) t |= iy[i * N0 + j];
        mask |= (unsigned)(t != 0) << i;
    }
    return mask;
}

struct DecBandCtx { int i, intensity, spread, tf_change; i32 remaining_bits; u32 seed; };

struct DecSplit { int inv, imid, iside, delta, itheta, qalloc; };

// compute_theta (bands.c:645-817) with encode = 0
CA_DEV DecSplit compute_theta_dec(RangeDec &dec, DecBandCtx &ctx, int N, int *b, int B, int B0, int LM, int stereo, int *fill)
{
    DecSplit sc;
    const int i = ctx.i;
    int inv = 0, itheta = 0;
    int pulse_cap = CLT_logN400[i] + LM * (1 << BITRES);
    int offset = (pulse_cap >> 1) - (stereo && N == 2 ? QTHETA_OFFSET_TWOPHASE : QTHETA_OFFSET);
    int qn = compute_qn(N, *b, offset, pulse_cap, stereo);
    if (stereo && i >= ctx.intensity) qn = 1;
    i32 tell = (i32)ec_tell_frac(dec);
    if (qn != 1) {
        if (stereo && N > 2) {
            const int p0 = 3, x0 = qn / 2, ft = p0 * (x0 + 1) + x0;
            int fs = (int)ec_decode(dec, (u32)ft), x;
            if (fs < (x0 + 1) * p0) x = fs / p0;
            else x = x0 + 1 + (fs - (x0 + 1) * p0);
            ec_dec_update(dec, (u32)(x <= x0 ? p0 * x : (x - 1 - x0) + (x0 + 1) * p0),
                          (u32)(x <= x0 ? p0 * (x + 1) : (x - x0) + (x0 + 1) * p0), (u32)ft);
            itheta = x;
        } else if (B0 > 1 || stereo) {
            itheta = (int)ec_dec_uint(dec, (u32)(qn + 1));
        } else {
            int fs, fl, ft = ((qn >> 1) + 1) * ((qn >> 1) + 1);
            int fm = (int)ec_decode(dec, (u32)ft);
            if (fm < ((qn >> 1) * ((qn >> 1) + 1) >> 1)) {
                itheta = (int)((isqrt32(8 * (u32)fm + 1) - 1) >> 1);
                fs = itheta + 1;
                fl = itheta * (itheta + 1) >> 1;
            } else {
                itheta = (int)((2 * (u32)(qn + 1) - isqrt32(8 * (u32)(ft - fm - 1) + 1)) >> 1);
                fs = qn + 1 - itheta;
                fl = ft - ((qn + 1 - itheta) * (qn + 2 - itheta) >> 1);
            }
            ec_dec_update(dec, (u32)fl, (u32)(fl + fs), (u32)ft);
        }
        itheta = (int)((u32)(itheta * 16384) / (u32)qn);
    } else if (stereo) {
        if (*b > 2 << BITRES && ctx.remaining_bits > 2 << BITRES) inv = ec_dec_bit_logp(dec, 2);
        else inv = 0;
        itheta = 0;
    }
    int qalloc = (int)((i32)ec_tell_frac(dec) - tell);
    *b -= qalloc;
    int imid, iside, delta;
    if (itheta == 0) { imid = 32767; iside = 0; *fill &= (1 << B) - 1; delta = -16384; }
    else if (itheta == 16384) { imid = 0; iside = 32767; *fill &= ((1 << B) - 1) << B; delta = 16384; }
    else {
        imid = bitexact_cos((i16)itheta);
        iside = bitexact_cos((i16)(16384 - itheta));
        delta = frac_mul16((N - 1) << 7, bitexact_log2tan(iside, imid));
    }
    sc.inv = inv; sc.imid = imid; sc.iside = iside; sc.delta = delta; sc.itheta = itheta; sc.qalloc = qalloc;
    return sc;
}

// The folding source of a band ("lowband") needs the same TF recombination / time-division / de-interleaving the
// band itself gets before it can be copied under a pulse-less partition (bands.c:1083-1134). That is three passes
// over the band, and at ordinary rates no partition folds, so the passes are recorded here and only run when the
// first partition asks for the data (the result is the same: the source is read-only until then).
struct LowbandPrep {
    x16_t *src;          // norm + effective_lowband (nullptr: nothing to fold from)
    x16_t *scratch;      // where the transformed copy goes (nullptr: transform in place, last band)
    int N, B, N_B, tf_change, recombine, longBlocks, need_copy;
    int ready;
    x16_t *ptr;          // valid once ready
};
template <class D> CA_DEV void lowband_prepare(D &F, LowbandPrep &lp);

// quant_partition (bands.c:864-1042), encode = 0. The reference recurses; here the second child of a split
// is parked on a small stack (as in the encoder), each node carrying its own lowband, gain, fill and the
// shift its collapse mask enters the parent's mask with.
template <class D>
CA_DEV unsigned quant_partition_dec(D &F, RangeDec &dec, DecBandCtx &ctx, x16_t *X, int N, int b, int B, LowbandPrep &lp,
                                    int LM, i32 gain, int fill)
{
    struct Parked { x16_t *X; int lowband; int b, N, B, LM, first_bits, allow, fill, shift; i32 remaining, gain; };
    Parked st[5];
    int sp = 0, shift = 0;
    int lowband = lp.src ? 0 : -1;                  // offset into the (lazily prepared) folding source, -1: none
    unsigned cm_total = 0;
    for (;;) {
        while (LM != -1 && N > 2 && b > pulse_cache_max(ctx.i, LM) + 12) {
            const int B0 = B;
            N >>= 1;
            x16_t *Y = X + N;
            LM -= 1;
            if (B == 1) fill = (fill & 1) | (fill << 1);
            B = (B + 1) >> 1;
            DecSplit sc = compute_theta_dec(dec, ctx, N, &b, B, B0, LM, 0, &fill);
            int delta = sc.delta;
            const int itheta = sc.itheta;
            const i32 mid = sc.imid, side = sc.iside;
            if (B0 > 1 && (itheta & 0x3fff)) {
                if (itheta > 8192) delta -= delta >> (4 - LM);
                else delta = imin(0, delta + (N << BITRES >> (5 - LM)));
            }
            const int mbits = imax(0, imin(b, (b - delta) / 2));
            const int sbits = b - mbits;
            ctx.remaining_bits -= sc.qalloc;
            const int lowband2 = lowband >= 0 ? lowband + N : -1;
            const i32 gmid = (i16)mul16_16_p15(gain, mid), gside = (i16)mul16_16_p15(gain, side);
            const int mid_first = mbits >= sbits;
            Parked &p = st[sp++];
            p.N = N; p.B = B; p.LM = LM; p.remaining = ctx.remaining_bits;
            if (mid_first) {
                p.X = Y; p.lowband = lowband2; p.b = sbits; p.first_bits = mbits; p.allow = itheta != 0;
                p.gain = gside; p.fill = fill >> B; p.shift = shift + (B0 >> 1);
                b = mbits; gain = gmid;                     // X, lowband, fill, shift unchanged
            } else {
                p.X = X; p.lowband = lowband; p.b = mbits; p.first_bits = sbits; p.allow = itheta != 16384;
                p.gain = gmid; p.fill = fill; p.shift = shift;
                X = Y; lowband = lowband2; b = sbits; gain = gside; fill = fill >> B; shift = shift + (B0 >> 1);
            }
        }
        // leaf (bands.c:983-1039)
        unsigned cm = 0;
        int q = bits2pulses(ctx.i, LM, b);
        int curr_bits = pulses2bits(ctx.i, LM, q);
        ctx.remaining_bits -= curr_bits;
        while (ctx.remaining_bits < 0 && q > 0) {
            ctx.remaining_bits += curr_bits;
            q--;
            curr_bits = pulses2bits(ctx.i, LM, q);
            ctx.remaining_bits -= curr_bits;
        }
        if (q != 0) {
            cm = alg_unquant_dec(F, X, N, get_pulses(q), ctx.spread, B, dec, gain);
        } else {
            const unsigned cm_mask = (1u << B) - 1;
            fill &= (int)cm_mask;
            if (!fill) {
                for (int j = 0; j < N; j++) X[j] = 0;
            } else {
                if (lowband < 0) {
                    for (int j = 0; j < N; j++) {
                        ctx.seed = celt_lcg_rand(ctx.seed);
                        X[j] = (i16)((i32)ctx.seed >> 20);
                    }
                    cm = cm_mask;
                } else {
                    if (!lp.ready) lowband_prepare(F, lp);
                    x16_t *__restrict__ xo = X;
                    const x16_t *__restrict__ lb = lp.ptr + lowband;
#pragma unroll 8
                    for (int j = 0; j < N; j++) {
                        ctx.seed = celt_lcg_rand(ctx.seed);
                        i32 t = (ctx.seed & 0x8000) ? 4 : -4;                                 // QCONST16(1.0f/256, 10)
                        xo[j] = (i16)(lb[j] + t);
                    }
                    cm = (unsigned)fill;
                }
                renormalise_vector_dec(X, N, gain);
            }
        }
        cm_total |= cm << shift;
        if (sp == 0) break;
        const Parked &p = st[--sp];
        X = p.X; lowband = p.lowband; b = p.b; N = p.N; B = p.B; LM = p.LM; gain = p.gain; fill = p.fill; shift = p.shift;
        i32 rebalance = p.first_bits - (p.remaining - ctx.remaining_bits);
        if (rebalance > 3 << BITRES && p.allow) b += rebalance - (3 << BITRES);
    }
    return cm_total;
}

template <class P>
CA_DEV void haar1_ref(P X, int N0, int stride)                                                  // bands.c:580-594
{
    N0 >>= 1;
    {   // all butterflies are independent: one index space, eight per round trip (celt_enc_mid.h haar1_group)
        const int npairs = N0 * stride;
        if ((stride & (stride - 1)) == 0 && (npairs & 3) == 0) {
            int ls = 0;
            while ((1 << ls) < stride) ls++;
            int p0 = 0;
            for (; p0 + 8 <= npairs; p0 += 8) haar1_group<8>(X, p0, ls, stride);
            if (p0 < npairs) haar1_group<4>(X, p0, ls, stride);
            return;
        }
    }
    for (int i = 0; i < stride; i++) {
        int j = 0;
        for (; j + 4 <= N0; j += 4) {                // four butterflies loaded before the first store
            i32 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = X[stride * 2 * (j + u) + i]; b[u] = X[stride * (2 * (j + u) + 1) + i]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                i32 t1 = mul16_16(23170, a[u]), t2 = mul16_16(23170, b[u]);
                X[stride * 2 * (j + u) + i] = (i16)pshr32(add32(t1, t2), 15);
                X[stride * (2 * (j + u) + 1) + i] = (i16)pshr32(sub32(t1, t2), 15);
            }
        }
        for (; j < N0; j++) {
            i32 t1 = mul16_16(23170, X[stride * 2 * j + i]);
            i32 t2 = mul16_16(23170, X[stride * (2 * j + 1) + i]);
            X[stride * 2 * j + i] = (i16)pshr32(add32(t1, t2), 15);
            X[stride * (2 * j + 1) + i] = (i16)pshr32(sub32(t1, t2), 15);
        }
    }
}

template <class PT>
CA_DEV void deinterleave_hadamard_ref_via(x16_t *X, PT tmp, int N0, int stride, int hadamard);

template <class D>
CA_DEV void deinterleave_hadamard_ref(D &F, x16_t *X, int N0, int stride, int hadamard)
{
#if defined(CA_LANE_FRAME)
    if (N0 * stride <= 96) deinterleave_hadamard_ref_via(X, lds_col(F.lds_pvq16), N0, stride, hadamard);
    else deinterleave_hadamard_ref_via(X, priv((i16 *)F.tmp), N0, stride, hadamard);
#else
    deinterleave_hadamard_ref_via(X, (i16 *)F.tmp, N0, stride, hadamard);
#endif
}

template <class PT>
CA_DEV void deinterleave_hadamard_ref_via(x16_t *X, PT tmp, int N0, int stride, int hadamard)
{
    const int N = N0 * stride;
    const u8 *ordery = CLT_ordery_table + stride - 2;
    {
        const x16_t *__restrict__ x = X;
        for (int i = 0; i < stride; i++) {
            const int d = hadamard ? ordery[i] : i;
#pragma unroll 4
            for (int j = 0; j < N0; j++) tmp[d * N0 + j] = x[j * stride + i];
        }
    }
    {
        x16_t *__restrict__ x = X;
#pragma unroll 8
        for (int k = 0; k < N; k++) x[k] = tmp[k];
    }
}

template <class PT>
CA_DEV void interleave_hadamard_ref_via(x16_t *X, PT tmp, int N0, int stride, int hadamard);

template <class D>
CA_DEV void interleave_hadamard_ref(D &F, x16_t *X, int N0, int stride, int hadamard)
{
#if defined(CA_LANE_FRAME)
    if (N0 * stride <= 96) interleave_hadamard_ref_via(X, lds_col(F.lds_pvq16), N0, stride, hadamard);
    else interleave_hadamard_ref_via(X, priv((i16 *)F.tmp), N0, stride, hadamard);
#else
    interleave_hadamard_ref_via(X, (i16 *)F.tmp, N0, stride, hadamard);
#endif
}

template <class PT>
CA_DEV void interleave_hadamard_ref_via(x16_t *X, PT tmp, int N0, int stride, int hadamard)
{
    const int N = N0 * stride;
    const u8 *ordery = CLT_ordery_table + stride - 2;
    {
        const x16_t *__restrict__ x = X;
        for (int i = 0; i < stride; i++) {
            const int d = hadamard ? ordery[i] : i;
#pragma unroll 4
            for (int j = 0; j < N0; j++) tmp[j * stride + i] = x[d * N0 + j];
        }
    }
    {
        x16_t *__restrict__ x = X;
#pragma unroll 8
        for (int k = 0; k < N; k++) x[k] = tmp[k];
    }
}

template <class D>
CA_DEV void lowband_prepare(D &F, LowbandPrep &lp)
{
    x16_t *lowband = lp.src;
    if (lp.need_copy) {
        x16_t *__restrict__ d = lp.scratch;
        const x16_t *__restrict__ sQ = lp.src;
#if defined(CA_LANE_FRAME)
        if ((lp.N & 7) == 0 && ((((uintptr_t)d) | ((uintptr_t)sQ)) & 15) == 0) {              // 16 bytes per access, four in flight
#pragma unroll 4
            for (int j = 0; j < lp.N; j += 8) *reinterpret_cast<CA_AS_GLB v4i *>(d + j) = *reinterpret_cast<const CA_AS_GLB v4i *>(sQ + j);
        } else
#endif
        {
#pragma unroll 8
            for (int j = 0; j < lp.N; j++) d[j] = sQ[j];
        }
        lowband = lp.scratch;
    }
    int B = lp.B, N_B = lp.N_B, tf_change = lp.tf_change;
    for (int k = 0; k < lp.recombine; k++) haar1_ref(lowband, lp.N >> k, 1 << k);
    B >>= lp.recombine;
    N_B <<= lp.recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        haar1_ref(lowband, N_B, B);
        B <<= 1;
        N_B >>= 1;
        tf_change++;
    }
    if (B > 1) deinterleave_hadamard_ref(F, lowband, N_B >> lp.recombine, B << lp.recombine, lp.longBlocks);
    lp.ptr = lowband;
    lp.ready = 1;
}

// quant_band_n1 (bands.c:819-862), encode = 0
CA_DEV unsigned quant_band_n1_dec(RangeDec &dec, DecBandCtx &ctx, x16_t *X, x16_t *Y, x16_t *lowband_out)
{
    x16_t *x = X;
    for (int c = 0; c < (Y ? 2 : 1); c++) {
        int sign = 0;
        if (ctx.remaining_bits >= 1 << BITRES) {
            sign = (int)ec_dec_bits(dec, 1);
            ctx.remaining_bits -= 1 << BITRES;
        }
        x[0] = sign ? -16384 : 16384;                                                          // NORM_SCALING
        x = Y;
    }
    if (lowband_out) lowband_out[0] = (i16)(X[0] >> 4);
    return 1;
}

// quant_band (bands.c:1044-1174), encode = 0
template <class D>
CA_DEV unsigned quant_band_dec(D &F, RangeDec &dec, DecBandCtx &ctx, x16_t *X, int N, int b, int B, x16_t *lowband, int LM,
                               x16_t *lowband_out, i32 gain, x16_t *lowband_scratch, int fill)
{
    CA_STAMP_F(F, 16);
    const int N0 = N;
    int N_B = (int)((u32)N / (u32)B);
    int B0 = B, time_divide = 0, recombine = 0;
    const int longBlocks = B0 == 1;
    int tf_change = ctx.tf_change;
    if (N == 1) return quant_band_n1_dec(dec, ctx, X, nullptr, lowband_out);
    if (tf_change > 0) recombine = tf_change;
    LowbandPrep lp;
    lp.src = lowband;
    lp.scratch = lowband_scratch;
    lp.N = N; lp.B = B; lp.N_B = N_B; lp.tf_change = tf_change; lp.recombine = recombine; lp.longBlocks = longBlocks;
    lp.need_copy = lowband_scratch && lowband && (recombine || ((N_B & 1) == 0 && tf_change < 0) || B0 > 1);
    lp.ready = 0;
    lp.ptr = lowband;
    for (int k = 0; k < recombine; k++)
        fill = CLT_bit_interleave_table[fill & 0xF] | CLT_bit_interleave_table[fill >> 4] << 2;
    B >>= recombine;
    N_B <<= recombine;
    while ((N_B & 1) == 0 && tf_change < 0) {
        fill |= fill << B;
        B <<= 1;
        N_B >>= 1;
        time_divide++;
        tf_change++;
    }
    B0 = B;
    const int N_B0 = N_B;
    CA_STAMP_F(F, 7);
    unsigned cm = quant_partition_dec(F, dec, ctx, X, N, b, B, lp, LM, gain, fill);
    CA_STAMP_F(F, 8);
    // resynthesis
#if defined(CA_LANE_FRAME)
    // Lane build: undoing the band's time-frequency re-arrangement (interleave, haar1 levels) is data movement over N0
    // 16-bit bins. In place on X in HBM every step costs 2-byte loads and stores, one cache line per lane each; the band is
    // instead pulled into the idle per-lane LDS scratch with 16-byte loads (through the interleave), transformed there,
    // and written back once with 16-byte stores (same scheme as the encoder's band set-up, celt_enc_back.h).
    if (N0 <= LANE_SCRATCH_N && (N0 & 7) == 0 && ((uintptr_t)X & 15) == 0 && (B0 > 1 || time_divide > 0 || recombine > 0)) {
        LdsCol<i16> T = lds_col(F.lds_pvq16);
        {   // interleave_hadamard (bands.c:551-578) on the way in: X[d*Ni + j] -> T[j*stride + i], d = ordery[i] or i
            const int stride = B0 > 1 ? B0 << recombine : 1, Ni = B0 > 1 ? N_B >> recombine : N0;
            const u8 *ordery = CLT_ordery_table + stride - 2;
            int d = 0, j = 0, i = 0;
            if (stride > 1 && longBlocks) { while (ordery[i] != 0) i++; }
            // (the band's 16-byte loads eight at a time ahead of the scatter: one memory round trip per 64 bins, not per 8)
            for (int k0 = 0; k0 < N0; k0 += 64) {
                v4i wv[8];
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (k0 + 8 * c < N0) wv[c] = *reinterpret_cast<const CA_AS_GLB v4i *>(X + k0 + 8 * c);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    if (k0 + 8 * c < N0) {
                        const v4i w = wv[c];
                        const i32 v[8] = {(i16)w.x, w.x >> 16, (i16)w.y, w.y >> 16, (i16)w.z, w.z >> 16, (i16)w.w, w.w >> 16};
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            T[j * stride + i] = (i16)v[u];
                            if (++j == Ni) {
                                j = 0;
                                d++;
                                i = d;
                                if (stride > 1 && longBlocks && d < stride) { i = 0; while (ordery[i] != d) i++; }
                            }
                        }
                    }
                }
            }
        }
        N_B = N_B0;
        B = B0;
        for (int k = 0; k < time_divide; k++) {
            B >>= 1;
            N_B <<= 1;
            cm |= cm >> B;
            haar1_ref(T, N_B, B);
        }
        for (int k = 0; k < recombine; k++) {
            cm = CLT_bit_deinterleave_table[cm];
            haar1_ref(T, N0 >> k, 1 << k);
        }
        B <<= recombine;
        const i32 n = (i16)celt_sqrt(shl32(N0, 22));
        // (lowband_out = the folding source norm[] in private memory, 16-byte aligned like the band: eight values per store too)
        const bool lo16 = lowband_out && ((uintptr_t)lowband_out & 15) == 0;
        for (int k = 0; k < N0; k += 8) {
            u32 h[8], g[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i32 v = T[k + u];
                h[u] = (u32)v & 0xffffu;
                g[u] = (u32)(i32)(i16)mul16_16_q15(n, v) & 0xffffu;
            }
            v4i w;
            w.x = (i32)(h[0] | (h[1] << 16)); w.y = (i32)(h[2] | (h[3] << 16));
            w.z = (i32)(h[4] | (h[5] << 16)); w.w = (i32)(h[6] | (h[7] << 16));
            *reinterpret_cast<CA_AS_GLB v4i *>(X + k) = w;
            if (lo16) {
                w.x = (i32)(g[0] | (g[1] << 16)); w.y = (i32)(g[2] | (g[3] << 16));
                w.z = (i32)(g[4] | (g[5] << 16)); w.w = (i32)(g[6] | (g[7] << 16));
                *reinterpret_cast<CA_AS_GLB v4i *>(lowband_out + k) = w;
            } else if (lowband_out) {
#pragma unroll
                for (int u = 0; u < 8; u++) lowband_out[k + u] = (i16)g[u];
            }
        }
        cm &= (1u << B) - 1;
        CA_STAMP_F(F, 9);
        return cm;
    }
#endif
    if (B0 > 1) interleave_hadamard_ref(F, X, N_B >> recombine, B0 << recombine, longBlocks);
    N_B = N_B0;
    B = B0;
    for (int k = 0; k < time_divide; k++) {
        B >>= 1;
        N_B <<= 1;
        cm |= cm >> B;
        haar1_ref(X, N_B, B);
    }
    for (int k = 0; k < recombine; k++) {
        cm = CLT_bit_deinterleave_table[cm];
        haar1_ref(X, N0 >> k, 1 << k);
    }
    B <<= recombine;
    if (lowband_out) {
        i32 n = (i16)celt_sqrt(shl32(N0, 22));
        {
            x16_t *__restrict__ d = lowband_out;
            const x16_t *__restrict__ sQ = X;
#pragma unroll 8
            for (int j = 0; j < N0; j++) d[j] = (i16)mul16_16_q15(n, sQ[j]);
        }
    }
    cm &= (1u << B) - 1;
    CA_STAMP_F(F, 9);
    return cm;
}

#if defined(CA_LANE_FRAME)
// eight bins of this stream's X (a row of opusgpu_celt_dec_state in HBM, 16-byte aligned) per access: celt_enc_mid.h ld_bins8 / st_bins8
CA_DEV void dec_ld8(const x16_t *p, i32 v[8]) { ld_bins8(p, v); }
CA_DEV void dec_st8(x16_t *p, const i32 v[8]) { st_bins8(p, v); }
#endif

CA_DEV void stereo_merge_dec(x16_t *X, x16_t *Y, i32 mid, int N)                                    // bands.c:375-427
{
#if defined(CA_LANE_FRAME)
    // Lane build: X / Y are this stream's rows in HBM, so every access costs the wavefront a cache line per lane whatever its
    // width, and a load per trip is an exposed memory round trip per trip: eight bins per access, several accesses in flight,
    // a group's loads ahead of its stores. (The sums wrap, so their order is free.)
    if ((N & 7) == 0 && ((((uintptr_t)X) | ((uintptr_t)Y)) & 15) == 0) {
        i32 xp = 0, side = 0;
#pragma unroll 4
        for (int j = 0; j < N; j += 8) {
            i32 xv[8], yv[8];
            dec_ld8(X + j, xv);
            dec_ld8(Y + j, yv);
#pragma unroll
            for (int u = 0; u < 8; u++) { xp = mac16_16(xp, yv[u], xv[u]); side = mac16_16(side, yv[u], yv[u]); }
        }
        xp = mul16_32_q15(mid, xp);
        const i32 mid2 = (i16)(mid >> 1);
        const i32 El = sub32(add32(mul16_16(mid2, mid2), side), shl32(xp, 1));
        const i32 Er = add32(add32(mul16_16(mid2, mid2), side), shl32(xp, 1));
        if (Er < 161061 || El < 161061) {                                                      // QCONST32(6e-4f, 28)
#pragma unroll 4
            for (int j = 0; j < N; j += 8) {
                i32 xv[8];
                dec_ld8(X + j, xv);
                dec_st8(Y + j, xv);
            }
            return;
        }
        int kl = celt_ilog2(El) >> 1, kr = celt_ilog2(Er) >> 1;
        i32 t = vshr32(El, (kl - 7) << 1);
        const i32 lgain = celt_rsqrt_norm(t);
        t = vshr32(Er, (kr - 7) << 1);
        const i32 rgain = celt_rsqrt_norm(t);
        if (kl < 7) kl = 7;
        if (kr < 7) kr = 7;
        auto merge8 = [&](i32 *xv, i32 *yv) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const i32 l = (i16)mul16_16_p15(mid, xv[u]);
                const i32 r = yv[u];
                xv[u] = (i16)pshr32(mul16_16(lgain, sub16(l, r)), kl + 1);
                yv[u] = (i16)pshr32(mul16_16(rgain, add16(l, r)), kr + 1);
            }
        };
        int j = 0;
        for (; j + 16 <= N; j += 16) {
            i32 xv[2][8], yv[2][8];
#pragma unroll
            for (int g = 0; g < 2; g++) { dec_ld8(X + j + 8 * g, xv[g]); dec_ld8(Y + j + 8 * g, yv[g]); }
#pragma unroll
            for (int g = 0; g < 2; g++) {
                merge8(xv[g], yv[g]);
                dec_st8(X + j + 8 * g, xv[g]);
                dec_st8(Y + j + 8 * g, yv[g]);
            }
        }
        for (; j < N; j += 8) {
            i32 xv[8], yv[8];
            dec_ld8(X + j, xv);
            dec_ld8(Y + j, yv);
            merge8(xv, yv);
            dec_st8(X + j, xv);
            dec_st8(Y + j, yv);
        }
        return;
    }
#endif
    i32 xp = 0, side = 0;
    for (int j = 0; j < N; j++) { xp = mac16_16(xp, Y[j], X[j]); side = mac16_16(side, Y[j], Y[j]); }
    xp = mul16_32_q15(mid, xp);
    i32 mid2 = (i16)(mid >> 1);
    i32 El = sub32(add32(mul16_16(mid2, mid2), side), shl32(xp, 1));
    i32 Er = add32(add32(mul16_16(mid2, mid2), side), shl32(xp, 1));
    if (Er < 161061 || El < 161061) {                                                          // QCONST32(6e-4f, 28)
        for (int j = 0; j < N; j++) Y[j] = X[j];
        return;
    }
    int kl = celt_ilog2(El) >> 1, kr = celt_ilog2(Er) >> 1;
    i32 t = vshr32(El, (kl - 7) << 1);
    i32 lgain = celt_rsqrt_norm(t);
    t = vshr32(Er, (kr - 7) << 1);
    i32 rgain = celt_rsqrt_norm(t);
    if (kl < 7) kl = 7;
    if (kr < 7) kr = 7;
    for (int j = 0; j < N; j += 4) {                 // loads of a group before its stores
        i32 xv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { xv[u] = j + u < N ? (i32)X[j + u] : 0; yv[u] = j + u < N ? (i32)Y[j + u] : 0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (j + u >= N) continue;
            i32 l = (i16)mul16_16_p15(mid, xv[u]);
            i32 r = yv[u];
            X[j + u] = (i16)pshr32(mul16_16(lgain, sub16(l, r)), kl + 1);
            Y[j + u] = (i16)pshr32(mul16_16(rgain, add16(l, r)), kr + 1);
        }
    }
}

// quant_band_stereo (bands.c:1176-1335), encode = 0, is folded into quant_all_bands_dec below: a band is up to two
// quant_band jobs (left / right of a dual-stereo band, or mid / side in the order of their budgets) that go through ONE call
// site. quant_band_dec with everything it inlines (partition walk, pulse decoding, folding, resynthesis) is most of the
// kernel's code; written as the reference writes it -- two calls for dual stereo, four for the two orders of mid / side, one
// for two-bin bands -- it was inlined seven times, and a wavefront whose streams disagree on dual stereo or on the order of
// mid and side ran the copies one after the other.
CA_DEV void negate_band_dec(x16_t *Y, int N)
{
#if defined(CA_LANE_FRAME)
    if ((N & 7) == 0 && (((uintptr_t)Y) & 15) == 0) {
#pragma unroll 2
        for (int j = 0; j < N; j += 8) {
            i32 yv[8];
            dec_ld8(Y + j, yv);
#pragma unroll
            for (int u = 0; u < 8; u++) yv[u] = (i16)(-yv[u]);
            dec_st8(Y + j, yv);
        }
        return;
    }
#endif
    for (int j = 0; j < N; j++) Y[j] = (i16)(-Y[j]);
}

// quant_all_bands (bands.c:1337-1502), encode = 0, start 0, end 21, LM 3, C = 2
template <class D>
CA_DEV void quant_all_bands_dec(D &F, RangeDec &dec, int shortBlocks, int spread, int dual_stereo, int intensity,
                                i32 total_bits, i32 balance, int codedBands, u32 *seed)
{
    const int LM = LM3, M = M8, C = 2;
    const int B = shortBlocks ? M : 1;
    const i16 *eB = CLT_eband5ms;
    x16_t *X_ = F.X, *Y_ = F.X + FRAME;
    x16_t *norm = F.norm, *norm2 = F.norm + M * eB[NB - 1];
    x16_t *lowband_scratch = X_ + M * eB[NB - 1];
    int lowband_offset = 0, update_lowband = 1;
    DecBandCtx ctx;
    ctx.intensity = intensity;
    ctx.spread = spread;
    ctx.seed = *seed;
    for (int i = 0; i < NB; i++) {
        CA_STAMP_F(F, 14);
        ctx.i = i;
        const int last = i == NB - 1;
        x16_t *X = X_ + M * eB[i], *Y = Y_ + M * eB[i];
        const int N = M * eB[i + 1] - M * eB[i];
        i32 tell = (i32)ec_tell_frac(dec);
        if (i != 0) balance -= tell;
        i32 remaining_bits = total_bits - tell - 1;
        ctx.remaining_bits = remaining_bits;
        int b;
        if (i <= codedBands - 1) {
            i32 curr_balance = balance / imin(3, codedBands - i);
            b = imax(0, imin(16383, imin(remaining_bits + 1, F.pulses[i] + curr_balance)));
        } else {
            b = 0;
        }
        if (M * eB[i] - N >= M * eB[0] && (update_lowband || lowband_offset == 0)) lowband_offset = i;
        ctx.tf_change = F.tf_res[i];
        if (last) lowband_scratch = nullptr;
        int effective_lowband = -1;
        unsigned x_cm, y_cm;
        if (lowband_offset != 0 && (spread != SPREAD_AGGRESSIVE || B > 1 || ctx.tf_change < 0)) {
            effective_lowband = imax(0, M * eB[lowband_offset] - N);
            int fold_start = lowband_offset;
            while (M * eB[--fold_start] > effective_lowband) {}
            int fold_end = lowband_offset - 1;
            while (M * eB[++fold_end] < effective_lowband + N) {}
            x_cm = y_cm = 0;
            int fold_i = fold_start;
            do {
                x_cm |= F.collapse_masks[fold_i * C + 0];
                y_cm |= F.collapse_masks[fold_i * C + C - 1];
            } while (++fold_i < fold_end);
        } else {
            x_cm = y_cm = (1u << B) - 1;
        }
        CA_STAMP_F(F, 15);
        if (dual_stereo && i == intensity) {
            dual_stereo = 0;
            for (int j = 0; j < M * eB[i]; j++) norm[j] = (i16)(((i32)norm[j] + norm2[j]) >> 1);
        }
        x16_t *const lb = effective_lowband != -1 ? norm + effective_lowband : (x16_t *)nullptr;
        x16_t *const lb2 = effective_lowband != -1 ? norm2 + effective_lowband : (x16_t *)nullptr;
        x16_t *const lo_out = last ? (x16_t *)nullptr : norm + M * eB[i];
        x16_t *const lo_out2 = last ? (x16_t *)nullptr : norm2 + M * eB[i];
        // plan the band's jobs (bands.c:1176-1335 for the stereo cases)
        enum { DUAL = 0, STEREO = 1, STEREO_N2 = 2, DONE = 3 };
        int kind = DUAL, njobs = 2, rebal = 0, allow2 = 0, inv = 0, sign = 1, n2_swap = 0;
        i32 mid = 0, side = 0;
        x16_t *jx0 = X, *jx1 = Y, *jl0 = lb, *jl1 = lb2, *jo0 = lo_out, *jo1 = lo_out2, *js0 = lowband_scratch, *js1 = lowband_scratch;
        int jb0 = b / 2, jb1 = b / 2, jf0 = (int)x_cm, jf1 = (int)y_cm;
        i32 jg0 = 32767, jg1 = 32767;
        if (!dual_stereo) {
            if (N == 1) {
                x_cm = y_cm = quant_band_n1_dec(dec, ctx, X, Y, lo_out);
                kind = DONE;
                njobs = 0;
            } else {
                int fill = (int)(x_cm | y_cm);
                const int orig_fill = fill;
                int bs = b;                          // (quant_band_stereo's own copy: the band's b decides update_lowband below)
                CA_STAMP_F(F, 12);
                DecSplit sc = compute_theta_dec(dec, ctx, N, &bs, B, B, LM, 1, &fill);
                CA_STAMP_F(F, 13);
                inv = sc.inv;
                mid = sc.imid;
                side = sc.iside;
                const int itheta = sc.itheta;
                if (N == 2) {
                    int mbits = bs, sbits = 0;
                    if (itheta != 0 && itheta != 16384) sbits = 1 << BITRES;
                    mbits -= sbits;
                    n2_swap = itheta > 8192;
                    ctx.remaining_bits -= sc.qalloc + sbits;
                    if (sbits) sign = 1 - 2 * (int)ec_dec_bits(dec, 1);
                    kind = STEREO_N2;
                    njobs = 1;
                    jx0 = n2_swap ? Y : X; jb0 = mbits; jl0 = lb; jo0 = lo_out; jg0 = 32767; js0 = lowband_scratch; jf0 = orig_fill;
                } else {
                    const int mbits = imax(0, imin(bs, (bs - sc.delta) / 2));
                    const int sbits = bs - mbits;
                    ctx.remaining_bits -= sc.qalloc;
                    kind = STEREO;
                    njobs = 2;
                    rebal = 1;
                    if (mbits >= sbits) {
                        jx0 = X; jb0 = mbits; jl0 = lb; jo0 = lo_out; jg0 = 32767; js0 = lowband_scratch; jf0 = fill;
                        jx1 = Y; jb1 = sbits; jl1 = nullptr; jo1 = nullptr; jg1 = side; js1 = nullptr; jf1 = fill >> B;
                        allow2 = itheta != 0;
                    } else {
                        jx0 = Y; jb0 = sbits; jl0 = nullptr; jo0 = nullptr; jg0 = side; js0 = nullptr; jf0 = fill >> B;
                        jx1 = X; jb1 = mbits; jl1 = lb; jo1 = lo_out; jg1 = 32767; js1 = lowband_scratch; jf1 = fill;
                        allow2 = itheta != 16384;
                    }
                }
            }
        }
        // ... and run them through one call site
        const i32 rebalance0 = ctx.remaining_bits;
        unsigned cm0 = 0, cm1 = 0;
        for (int j = 0; j < njobs; j++) {
            int jb = j ? jb1 : jb0;
            if (j == 1 && rebal) {
                const i32 rebalance = jb0 - (rebalance0 - ctx.remaining_bits);
                if (rebalance > 3 << BITRES && allow2) jb += rebalance - (3 << BITRES);
            }
            const unsigned c = quant_band_dec(F, dec, ctx, j ? jx1 : jx0, N, jb, B, j ? jl1 : jl0, LM, j ? jo1 : jo0, j ? jg1 : jg0,
                                              j ? js1 : js0, j ? jf1 : jf0);
            if (j) cm1 = c; else cm0 = c;
        }
        if (kind == DUAL) {
            x_cm = cm0;
            y_cm = cm1;
        } else if (kind == STEREO) {
            CA_STAMP_F(F, 17);
            stereo_merge_dec(X, Y, mid, N);
            CA_STAMP_F(F, 18);
            if (inv) negate_band_dec(Y, N);
            x_cm = y_cm = cm0 | cm1;
        } else if (kind == STEREO_N2) {
            x16_t *x2 = n2_swap ? Y : X, *y2 = n2_swap ? X : Y;
            y2[0] = (i16)(-sign * x2[1]);
            y2[1] = (i16)(sign * x2[0]);
            X[0] = (i16)mul16_16_q15(mid, X[0]);
            X[1] = (i16)mul16_16_q15(mid, X[1]);
            Y[0] = (i16)mul16_16_q15(side, Y[0]);
            Y[1] = (i16)mul16_16_q15(side, Y[1]);
            i32 t = X[0];
            X[0] = (i16)sub16(t, Y[0]);
            Y[0] = (i16)add16(t, Y[0]);
            t = X[1];
            X[1] = (i16)sub16(t, Y[1]);
            Y[1] = (i16)add16(t, Y[1]);
            if (inv) negate_band_dec(Y, N);
            x_cm = y_cm = cm0;
        }
        F.collapse_masks[i * C + 0] = (u8)x_cm;
        F.collapse_masks[i * C + C - 1] = (u8)y_cm;
        balance += F.pulses[i] + tell;
        update_lowband = b > (N << BITRES);
    }
    *seed = ctx.seed;
}

// anti_collapse (bands.c:241-335), C = 2, LM = 3
template <class D>
CA_DEV void anti_collapse_dec(D &F, const i16 *logE, const i16 *prev1logE, const i16 *prev2logE, u32 seed)
{
    const int LM = LM3, C = 2;
    for (int i = 0; i < NB; i++) {
        const int N0 = CLT_eband5ms[i + 1] - CLT_eband5ms[i];
        int depth = (int)((u32)(1 + F.pulses[i]) / (u32)N0) >> LM;
        i32 thresh32 = celt_exp2((i16)neg32(shl16(depth, 10 - BITRES))) >> 1;
        i32 thresh = (i16)mul16_32_q15(16384, imin(32767, thresh32));
        i32 sqrt_1;
        int shift;
        {
            i32 t = N0 << LM;
            shift = celt_ilog2(t) >> 1;
            t = shl32(t, (7 - shift) << 1);
            sqrt_1 = celt_rsqrt_norm(t);
        }
        for (int c = 0; c < C; c++) {
            i32 prev1 = prev1logE[c * NB + i], prev2 = prev2logE[c * NB + i];
            i32 Ediff = (i32)logE[c * NB + i] - imin(prev1, prev2);
            Ediff = imax(0, Ediff);
            i32 r;
            if (Ediff < 16384) {
                i32 r32 = celt_exp2((i16)neg32((i16)Ediff)) >> 1;
                r = (i16)(2 * imin(16383, r32));
            } else {
                r = 0;
            }
            r = (i16)mul16_16_q14(23170, imin(23169, r));                                      // LM == 3
            r = (i16)(imin(thresh, r) >> 1);
            r = (i16)(mul16_16_q15(sqrt_1, r) >> shift);
            x16_t *X = F.X + c * FRAME + (CLT_eband5ms[i] << LM);
            int renormalize = 0;
            for (int k = 0; k < 1 << LM; k++) {
                if (!(F.collapse_masks[i * C + c] & 1 << k)) {
                    for (int j = 0; j < N0; j++) {
                        seed = celt_lcg_rand(seed);
                        X[(j << LM) + k] = (i16)((seed & 0x8000) ? r : -r);
                    }
                    renormalize = 1;
                }
            }
            if (renormalize) renormalise_vector_dec(X, N0 << LM, 32767);
        }
    }
}

// denormalise_bands (bands.c:169-238) for one channel, start 0, end 21, downsample 1; lanes share a band's bins
// denormalise_bands (bands.c:169-239), start 0, end 21, M = 8, no downsampling. The reference walks the bands; a wavefront doing
// that pays 21 dependent steps (the band's energy from HBM, then its 8-176 bins, a sixth of the lanes busy on the narrow bands).
// Here the 21 gains are computed side by side (one lane per band), and the 960 bins are then walked 64 at a time with all of a
// lane's loads issued together; a bin finds its band through a 120-entry table (bands start at multiples of eight bins).
template <class S>
CA_DEV void denormalise_bands_dec(S &L, const i16 *X, const i16 *bandLogE, int silence)
{
    const int N = FRAME;
    i32 *freq = L.freq;
    for (int i = lane(); i < NB; i += LANES) {
        i32 lg = add16(bandLogE[i], shl16(CLT_eMeans[i], 6));
        int shift = 16 - (lg >> 10);
        i32 g;
        if (shift > 31) { shift = 0; g = 0; }
        else g = celt_exp2_frac(lg & 1023);
        if (shift < -2) { g = 32767; shift = -2; }
        L.dn_gain[i] = (i16)g;
        L.dn_shift[i] = (i8)shift;
    }
    for (int k = lane(); k < N / 8; k += LANES) {
        int b = 0;
#pragma unroll
        for (int i = 1; i < NB; i++) b += k >= CLT_eband5ms[i];
        L.dn_band8[k] = (u8)b;
    }
    wave_sync();
    if (silence) {
        for (int k = lane(); k < N; k += LANES) freq[k] = 0;
        wave_sync();
        return;
    }
    const int bound = M8 * CLT_eband5ms[NB];
    if (LANES == 64) {
        i32 xv[(FRAME + 63) / 64];
#pragma unroll
        for (int m = 0; m < (FRAME + 63) / 64; m++) xv[m] = lane() + 64 * m < bound ? (i32)X[lane() + 64 * m] : 0;
#pragma unroll
        for (int m = 0; m < (FRAME + 63) / 64; m++) {
            const int j = lane() + 64 * m;
            if (j < N) {
                i32 v = 0;
                if (j < bound) {
                    const int b = L.dn_band8[j >> 3];
                    const i32 g = L.dn_gain[b];
                    const int shift = L.dn_shift[b];
                    v = shift < 0 ? shl32(mul16_16(xv[m], g), -shift) : mul16_16(xv[m], g) >> shift;
                }
                freq[j] = v;
            }
        }
    } else {
        for (int j = lane(); j < N; j += LANES) {
            i32 v = 0;
            if (j < bound) {
                const int b = L.dn_band8[j >> 3];
                const i32 g = L.dn_gain[b];
                const int shift = L.dn_shift[b];
                v = shift < 0 ? shl32(mul16_16(X[j], g), -shift) : mul16_16(X[j], g) >> shift;
            }
            freq[j] = v;
        }
    }
    wave_sync();
}

// comb_filter (celt.c:183-237) as the decoder uses it: y == x, in place, looking back into the history
CA_DEV void comb_filter_inplace_dec(i32 *x, int T0, int T1, int N, i32 g0, i32 g1, int tapset0, int tapset1)
{
    if (g0 == 0 && g1 == 0) return;
    const i16 *G = CLT_comb_gains;
    i32 g00 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 0]), g01 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 1]),
        g02 = (i16)mul16_16_p15(g0, G[tapset0 * 3 + 2]);
    i32 g10 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 0]), g11 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 1]),
        g12 = (i16)mul16_16_p15(g1, G[tapset1 * 3 + 2]);
    i32 x1 = x[-T1 + 1], x2 = x[-T1], x3 = x[-T1 - 1], x4 = x[-T1 - 2];
    int overlap = OVL;
    if (g0 == g1 && T0 == T1 && tapset0 == tapset1) overlap = 0;
    int i;
    for (i = 0; i < overlap; i++) {
        i32 x0 = x[i - T1 + 2];
        i32 w = CLT_window120[i];
        i32 f = (i16)mul16_16_q15(w, w);
        i32 nf = (i16)(32767 - f);
        i32 y = x[i];
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(nf, g00), x[i - T0]));
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(nf, g01), add32(x[i - T0 + 1], x[i - T0 - 1])));
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(nf, g02), add32(x[i - T0 + 2], x[i - T0 - 2])));
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g10), x2));
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g11), add32(x1, x3)));
        y = add32(y, mul16_32_q15((i16)mul16_16_q15(f, g12), add32(x0, x4)));
        x[i] = y;
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
    if (g1 == 0) return;
    // comb_filter_const_c (celt.c:156-181), starting where the cross-fade stopped
    i32 *xc = x + i;
    const int Nc = N - i;
    x4 = xc[-T1 - 2]; x3 = xc[-T1 - 1]; x2 = xc[-T1]; x1 = xc[-T1 + 1];
    for (int k = 0; k < Nc; k++) {
        i32 x0 = xc[k - T1 + 2];
        xc[k] = add32(add32(add32(xc[k], mul16_32_q15(g10, x2)), mul16_32_q15(g11, add32(x1, x3))), mul16_32_q15(g12, add32(x0, x4)));
        x4 = x3; x3 = x2; x2 = x1; x1 = x0;
    }
}

struct DecResult { int samples; u32 final_range; };

// Stage 1 of opus_decode() of one CELT-only 20 ms stereo packet (code 0; data: the whole packet incl. TOC):
// everything up to the normalised bands (one lane per stream). Leaves X and the frame parameters in the state's
// hand-off fields; st->mid_valid == 0 tells the later stages to leave this stream alone.
template <class D>
CA_DEV DecResult celt_decode_front(D &F, opusgpu_celt_dec_state *st, const u8 *data, int len)
{
    DecResult res;
    res.samples = OPUSGPU_INVALID_PACKET;
    res.final_range = 0;
    st->mid_valid = 0;
    F.X = (x16_t *)st->mid_X;
    F.norm = (x16_t *)st->mid_norm;
    const int C = 2, N = FRAME, LM = LM3, M = M8;
    if (len < 2) { res.samples = len < 1 ? OPUSGPU_BAD_ARG : OPUSGPU_UNIMPLEMENTED; return res; }   // len == 1: PLC/DTX, not implemented
    // TOC (src/opus_decoder.c, opus_packet_parse_impl): CELT-only (0x80), fullband (3 << 5), 20 ms (3 << 3), stereo (4), code 0
    if (data[0] != 0xFC) { res.samples = OPUSGPU_UNIMPLEMENTED; return res; }
    data++;
    len--;
    if (len > 1275) { res.samples = OPUSGPU_INVALID_PACKET; return res; }
    RangeDec dec;
    ec_dec_init(dec, data, (u32)len);
    i16 *oldBandE = st->oldBandE, *oldLogE = st->oldLogE, *oldLogE2 = st->oldLogE2, *backgroundLogE = st->backgroundLogE;

    i32 total_bits = len * 8;
    i32 tell = ec_tell(dec);
    int silence;
    if (tell >= total_bits) silence = 1;
    else if (tell == 1) silence = ec_dec_bit_logp(dec, 15);
    else silence = 0;
    if (silence) {
        tell = len * 8;
        dec.nbits_total += tell - ec_tell(dec);
    }
    i32 postfilter_gain = 0;
    int postfilter_pitch = 0, postfilter_tapset = 0;
    if (tell + 16 <= total_bits) {
        if (ec_dec_bit_logp(dec, 1)) {
            int octave = (int)ec_dec_uint(dec, 6);
            postfilter_pitch = (16 << octave) + (int)ec_dec_bits(dec, (u32)(4 + octave)) - 1;
            int qg = (int)ec_dec_bits(dec, 3);
            if (ec_tell(dec) + 2 <= total_bits) postfilter_tapset = ec_dec_icdf(dec, CLT_tapset_icdf, 2);
            postfilter_gain = 3072 * (qg + 1);
        }
        tell = ec_tell(dec);
    }
    int isTransient = 0;
    if (tell + 3 <= total_bits) {
        isTransient = ec_dec_bit_logp(dec, 3);
        tell = ec_tell(dec);
    }
    const int shortBlocks = isTransient ? M : 0;
    const int intra_ener = tell + 3 <= total_bits ? ec_dec_bit_logp(dec, 3) : 0;
    CA_STAMP_F(F, 0);
    unquant_coarse_energy_dec(oldBandE, intra_ener, dec, C);
    CA_STAMP_F(F, 1);
    tf_decode_dec(isTransient, F.tf_res, dec);
    tell = ec_tell(dec);
    int spread_decision = SPREAD_NORMAL;
    if (tell + 4 <= total_bits) spread_decision = ec_dec_icdf(dec, CLT_spread_icdf, 5);
    for (int i = 0; i < NB; i++) {                                                              // init_caps (celt.c:255)
        int Nb = (CLT_eband5ms[i + 1] - CLT_eband5ms[i]) << LM;
        F.cap[i] = ((CLT_cache_caps50[NB * (2 * LM + C - 1) + i] + 64) * C * Nb) >> 2;
    }
    int dynalloc_logp = 6;
    total_bits <<= BITRES;
    tell = (i32)ec_tell_frac(dec);
    for (int i = 0; i < NB; i++) {
        int width = C * (CLT_eband5ms[i + 1] - CLT_eband5ms[i]) << LM;
        int quanta = imin(width << BITRES, imax(6 << BITRES, width));
        int dynalloc_loop_logp = dynalloc_logp, boost = 0;
        while (tell + (dynalloc_loop_logp << BITRES) < total_bits && boost < F.cap[i]) {
            int flag = ec_dec_bit_logp(dec, (u32)dynalloc_loop_logp);
            tell = (i32)ec_tell_frac(dec);
            if (!flag) break;
            boost += quanta;
            total_bits -= quanta;
            dynalloc_loop_logp = 1;
        }
        F.offsets[i] = boost;
        if (boost > 0) dynalloc_logp = imax(2, dynalloc_logp - 1);
    }
    const int alloc_trim = tell + (6 << BITRES) <= total_bits ? ec_dec_icdf(dec, CLT_trim_icdf, 7) : 5;
    i32 bits = ((len * 8) << BITRES) - (i32)ec_tell_frac(dec) - 1;
    const int anti_collapse_rsv = isTransient && LM >= 2 && bits >= ((LM + 2) << BITRES) ? (1 << BITRES) : 0;
    bits -= anti_collapse_rsv;
    AllocOut al = compute_allocation_wave(F, dec, C, alloc_trim, 0, 0, bits, 0, 0);
    unquant_fine_energy_dec(oldBandE, F.fine_quant, dec, C);
    CA_STAMP_F(F, 2);

    u32 rng = st->rng;
    quant_all_bands_dec(F, dec, shortBlocks, spread_decision, al.dual_stereo, al.intensity,
                        len * (8 << BITRES) - anti_collapse_rsv, al.balance, al.codedBands, &rng);
    st->rng = rng;
    CA_STAMP_F(F, 10);
    int anti_collapse_on = 0;
    if (anti_collapse_rsv > 0) anti_collapse_on = (int)ec_dec_bits(dec, 1);
    unquant_energy_finalise_dec(oldBandE, F.fine_quant, F.fine_priority, len * 8 - ec_tell(dec), dec, C);
    if (anti_collapse_on) anti_collapse_dec(F, oldBandE, oldLogE, oldLogE2, st->rng);
    if (silence)
        for (int i = 0; i < C * NB; i++) oldBandE[i] = -28672;

    // hand the frame over to the synthesis and post-filter stages
    st->mid_valid = 1;
    st->mid_isTransient = isTransient;
    st->mid_silence = silence;
    st->postfilter_period = imax(st->postfilter_period, MINP);
    st->postfilter_period_old = imax(st->postfilter_period_old, MINP);
    st->mid_pf_period_old = st->postfilter_period_old; st->mid_pf_period = st->postfilter_period; st->mid_pf_period_new = postfilter_pitch;
    st->mid_pf_gain_old = st->postfilter_gain_old; st->mid_pf_gain = st->postfilter_gain; st->mid_pf_gain_new = postfilter_gain;
    st->mid_pf_tapset_old = st->postfilter_tapset_old; st->mid_pf_tapset = st->postfilter_tapset; st->mid_pf_tapset_new = postfilter_tapset;
    st->postfilter_period = postfilter_pitch;                                                    // celt_decoder.c:1000-1011, LM != 0
    st->postfilter_gain = postfilter_gain;
    st->postfilter_tapset = postfilter_tapset;
    st->postfilter_period_old = st->postfilter_period;
    st->postfilter_gain_old = st->postfilter_gain;
    st->postfilter_tapset_old = st->postfilter_tapset;

    if (!isTransient) {
        for (int i = 0; i < 2 * NB; i++) { oldLogE2[i] = oldLogE[i]; oldLogE[i] = oldBandE[i]; }
        const i32 max_background_increase = st->loss_count < 10 ? M * 1 : 1024;               // M*QCONST16(0.001f, DB_SHIFT)
        for (int i = 0; i < 2 * NB; i++) backgroundLogE[i] = (i16)imin(backgroundLogE[i] + max_background_increase, oldBandE[i]);
    } else {
        for (int i = 0; i < 2 * NB; i++) oldLogE[i] = (i16)imin(oldLogE[i], oldBandE[i]);
    }
    st->rng = dec.rng;

    st->loss_count = 0;
    CA_STAMP_F(F, 11);
    if (ec_tell(dec) > 8 * len) { res.samples = OPUSGPU_INTERNAL_ERROR; return res; }
    if (dec.error) st->error = 1;
    res.samples = N;
    res.final_range = dec.rng;
    return res;
}

// Stage 2 (one wavefront per stream; LANES == 1 in the emulation): celt_synthesis (celt_decoder.c:287-350) for
// C == CC == 2 -- shift the history, denormalise, inverse MDCT with TDAC into decode_mem.
template <class S>
CA_DEV void celt_decode_synth(S &L, opusgpu_celt_dec_state *st)
{
    if (!uni(st->mid_valid)) return;
    const int N = FRAME;
    const int isTransient = uni(st->mid_isTransient), silence = uni(st->mid_silence);
    // OPUS_MOVE(decode_mem, decode_mem + N, DECODE_BUFFER_SIZE - N + overlap/2) of both channels first
    if (LANES == 64) {
        // the whole move through registers, 16 bytes per lane and access: the ten loads of the two channels, then their ten
        // stores -- block by block (load, store, load, ...) every block's load queued behind the previous block's store, 18
        // exposed memory round trips per channel, and the second channel's behind the first one's synthesis stores
        enum { NQ = (DEC_BUF - FRAME + OVL / 2) / 4, NM = (NQ + 63) / 64 };
        static_assert((DEC_BUF - FRAME + OVL / 2) % 4 == 0 && (FRAME % 4) == 0, "history move in 16-byte units");
        typedef int hv4 __attribute__((vector_size(16)));          // (gcc builds the wave-per-frame emulation: no ext_vector_type)
        hv4 hv[2][NM];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const hv4 *src = reinterpret_cast<const hv4 *>(st->decode_mem[c] + N);
#pragma unroll
            for (int m = 0; m < NM; m++)
                if (lane() + 64 * m < NQ) hv[c][m] = src[lane() + 64 * m];
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < 2; c++) {
            hv4 *dst = reinterpret_cast<hv4 *>(st->decode_mem[c]);
#pragma unroll
            for (int m = 0; m < NM; m++)
                if (lane() + 64 * m < NQ) dst[lane() + 64 * m] = hv[c][m];
        }
    } else {
        // ascending blocks, each read before it is written (the source runs 960 ahead of the destination)
        for (int c = 0; c < 2; c++) {
            i32 *mem = st->decode_mem[c];
            for (int k0 = 0; k0 < DEC_BUF - N + OVL / 2; k0 += LANES) {
                const int k = k0 + lane();
                i32 v = 0;
                if (k < DEC_BUF - N + OVL / 2) v = mem[k + N];
                wave_sync();
                if (k < DEC_BUF - N + OVL / 2) mem[k] = v;
            }
            wave_sync();
        }
    }
    wave_sync();
    for (int c = 0; c < 2; c++) {
        i32 *mem = st->decode_mem[c];
        i32 *out_syn = mem + DEC_BUF - N;
        denormalise_bands_dec(L, st->mid_X + c * N, st->oldBandE + c * NB, silence);
        if (isTransient) {
            const MdctTab T = mdct_global_tab<3>();
            mdct_backward_wave<3, 8>(L.freq, 1, L.f2, out_syn, T, lane());
        } else {
            const MdctTab T = mdct_global_tab<0>();
            mdct_backward_wave<0, 1>(L.freq, 1, L.f2, out_syn, T, lane());
        }
        wave_sync();
    }
}

// Stage 3 (one lane per (stream, channel)): the pitch post-filter in place on the synthesis output
// (celt_decoder.c:983-998) and deemphasis to interleaved int16 (celt_decoder.c:183-285: coef0 = 27853, no
// downsampling, no accumulation).
CA_DEV void celt_decode_post_channel(opusgpu_celt_dec_state *st, int c, i16 *pcm)
{
    if (!st->mid_valid) return;
    const int N = FRAME;
    i32 *out_syn = st->decode_mem[c] + DEC_BUF - N;
    comb_filter_inplace_dec(out_syn, st->mid_pf_period_old, st->mid_pf_period, 120, st->mid_pf_gain_old, st->mid_pf_gain,
                            st->mid_pf_tapset_old, st->mid_pf_tapset);
    comb_filter_inplace_dec(out_syn + 120, st->mid_pf_period, st->mid_pf_period_new, N - 120, st->mid_pf_gain, st->mid_pf_gain_new,
                            st->mid_pf_tapset, st->mid_pf_tapset_new);
    i32 m = st->preemph_memD[c];
    for (int j = 0; j < N; j++) {
        i32 t = add32(out_syn[j], m);
        m = mul16_32_q15(27853, t);
        i32 v = pshr32(t, 12);
        v = imax(v, -32768);
        v = imin(v, 32767);
        pcm[j * 2 + c] = (i16)v;
    }
    st->preemph_memD[c] = m;
}

// all three stages, for the host emulation
template <class D, class S>
CA_DEV DecResult celt_decode_frame(D &F, S &L, opusgpu_celt_dec_state *st, const u8 *data, int len, i16 *pcm)
{
    DecResult r = celt_decode_front(F, st, data, len);
    celt_decode_synth(L, st);
    for (int c = 0; c < 2; c++) celt_decode_post_channel(st, c, pcm);
    return r;
}

}  // namespace ca
