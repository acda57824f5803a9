// celt_enc_lane.h -- the back phase of the CELT frame encoder for ONE LANE per frame, with no private-memory working set.
//
// Round 2's lane kernel compiled the wave-per-frame sources with one lane and kept their per-frame arrays -- thirteen
// run-time indexed int[21] allocation / Viterbi arrays, five energy arrays, the coarse-energy byte save, the parked
// partitions, fall-back buffers for wide leaves: 4.3 KB per lane -- in private memory: 566 MB of swizzled scratch for a
// 65 536-frame launch, far past L2 and the Infinity Cache, 16.9 KB written per frame for a 319-byte packet. This file is the
// same arithmetic (celt_encoder.c:552-838, quant_bands.c:158-439, rate.c:248-639) laid out for a lane instead:
//
//  * every small per-band array lives in this lane's LDS column ([slot][lane], 16-bit slots; celt_back_lane_kernel owns 264 of
//    them per lane). The column is time-multiplexed: TF analysis stages a band in it, the stages between TF and PVQ keep the
//    energies / allocation arrays in it (slot map below), the PVQ walk has slots 0..239 for its band buffers and finds the
//    per-band bit budgets in 240..260;
//  * what is a pure function of (band, C, LM, alloc_trim) is recomputed where it is used instead of being stored:
//    thresh, trim_offset, cap (rate.c:570-591, celt.c:246-256) -- a table look-up and one or two multiplies each;
//  * decisions that are one bit per band (the TF Viterbi's back pointers, tf_res) are bit masks in a register, the TF metric
//    (-7 .. 6) is a nibble;
//  * read-only inputs are read where they already are (the frame's FrameMid record in HBM: bandE, oldLogE, oldLogE2), and
//    what must survive the PVQ walk but is only touched again after it (oldBandE, the fine-energy decisions and error
//    signs of quant_energy_finalise) is parked in fields of that record the back phase has finished reading;
//  * the parked second children of split partitions are three packed words each in registers (celt_enc_back.h), a leaf too
//    wide for the column is searched in the bins of X this frame has already coded (they are dead: resynth == 0).
// The result: .private_segment_fixed_size of celt_back_lane_kernel is the compiler's own spills only.
//
// Slot map of the lane's column between TF analysis and the PVQ walk (M_*), 16-bit slots:
//     0.. 41  bandLogE (copy of the hand-off's)
//    42.. 83  oldBandE: the hand-off's, then coarse energy's inter pass in place (quant_bands.c:158-267: each element is read
//             before it is written), fine energy's updates
//    84..125  error (of the pass that won)
//   126..167  coarse energy: oldE of the intra pass    | dynalloc: follower      | allocation: bits1 as (lo, hi) pairs
//   168..209  coarse energy: error of the intra pass   | 168..188 noise_floor,   | allocation: bits2 as (lo, hi) pairs
//   210..257  coarse energy: the intra pass's bytes    | 189..230 bandLogE2 copy |   then fine_quant 126.., fine_priority 147..
//   231..251  dynalloc offsets (dead before the first per-band budget is written)
//   240..260  per-band bit budgets `pulses` -- stay through the PVQ walk
#pragma once
#if !defined(CA_LANE_FRAME)
#error "celt_enc_lane.h is the lane-per-frame build's back phase"
#endif
#include "celt_enc_back.h"

namespace ca {

enum { M_LOGE = 0, M_OLDE = 42, M_ERR = 84, M_EA = 126, M_ERRA = 168, M_SAVE = 210, M_SAVE_SLOTS = 48,
       M_FOLLOWER = 126, M_NOISE = 168, M_E2 = 189, M_OFFSETS = 231, M_BITS1 = 126, M_BITS2 = 168, M_FQ = 126, M_FP = 147 };
static_assert(M_SAVE + M_SAVE_SLOTS <= LS_SLOTS && M_OFFSETS + NB <= LS_SLOTS && LS_PULSES + NB <= LS_SLOTS, "lane column");

typedef LdsCol<i16> Col;
CA_DEV Col mcol(BackLds &F, int slot) { return lds_col(F.col) + slot; }
// a 32-bit value in two 16-bit slots of the column
CA_DEV i32 col_get32(Col c, int k) { return (i32)((u32)(u16)c[2 * k] | ((u32)(u16)c[2 * k + 1] << 16)); }
CA_DEV void col_set32(Col c, int k, i32 v) { c[2 * k] = (i16)v; c[2 * k + 1] = (i16)((u32)v >> 16); }

// pure functions of the band (C = 2, LM = 3, end = 21)
CA_DEV int band_w(int j) { return CLT_eband5ms[j + 1] - CLT_eband5ms[j]; }
CA_DEV i32 alloc_cap(int j) { return ((i32)(CLT_cache_caps50[NB * (2 * LM3 + 1) + j] + 64) * 2 * (band_w(j) << LM3)) >> 2; }      // celt.c:246-256
CA_DEV i32 alloc_thresh(int j) { return imax(2 << BITRES, ((3 * band_w(j)) << LM3 << BITRES) >> 4); }                            // rate.c:573
CA_DEV i32 alloc_trim_offset(int j, int alloc_trim)                                                                             // rate.c:575-580
{
    const int w = band_w(j);
    i32 to = (2 * w * (alloc_trim - 5 - LM3) * (NB - j - 1) * (1 << (LM3 + BITRES))) >> 6;
    if ((w << LM3) == 1) to -= 2 << BITRES;
    return to;
}

// ---- tf_analysis (celt_encoder.c:552-713) -------------------------------------------------------------------------
// One band's metric with ONE staging buffer: the reference analyses a second copy of the band (tmp_1, transient frames) with
// one more haar level; here the same buffer takes that level first and the band is then fetched again.
CA_DEV void tf_stage_band(const x16_t *Xb, Col tmp, int N)
{
#pragma unroll 4                                   // (several loads in flight: one per trip is an exposed round trip per trip)
    for (int j = 0; j < N; j += 8) {
        i32 v[8];
        ld_bins8(Xb + j, v);
#pragma unroll
        for (int u = 0; u < 8; u++) tmp[j + u] = (i16)v[u];
    }
}

CA_DEV int tf_band_metric_lane(const x16_t *Xb, Col tmp, int N, int narrow, int isTransient, i32 bias)
{
    const int LM = LM3;
    i32 L1_m1 = 0;
    const bool extra = isTransient && !narrow;
    if (extra) {
        CA_COUNT("lane.tf_extra_level", N);
        tf_stage_band(Xb, tmp, N);
        haar1_wave(tmp, N >> LM, 1 << LM);
        L1_m1 = l1_metric_wave(tmp, N, LM + 1, bias);
    }
    tf_stage_band(Xb, tmp, N);
    i32 L1 = l1_metric_wave(tmp, N, isTransient ? LM : 0, bias);
    i32 best_L1 = L1;
    int best_level = 0;
    if (extra && L1_m1 < best_L1) { best_L1 = L1_m1; best_level = -1; }
    for (int k = 0; k < LM + !(isTransient || narrow); k++) {
        int B = isTransient ? (LM - k - 1) : (k + 1);
        haar1_wave(tmp, N >> k, 1 << k);
        L1 = l1_metric_wave(tmp, N, B, bias);
        if (L1 < best_L1) { best_L1 = L1; best_level = k + 1; }
    }
    int metric = isTransient ? 2 * best_level : -2 * best_level;
    if (narrow && (metric == 0 || metric == -2 * LM)) metric -= 1;
    return metric;
}

// metric of band i out of three words of eight nibbles (value + 8)
CA_DEV int tf_nib(u32 m0, u32 m1, u32 m2, int i)
{
    const u32 w = i < 8 ? m0 : i < 16 ? m1 : m2;
    return (int)((w >> (4 * (i & 7))) & 15) - 8;
}

// returns tf_select; F.tf_bits = the per-band decisions tf_res[i] (bit i)
CA_DEVFN int tf_analysis_lane(BackLds &F, int isTransient, int lambda, i32 tf_estimate, int tf_chan)
{
    const int len = NB, LM = LM3;
    const x16_t *X = F.x16;
    const i32 bias = (i16)mul16_16_q14(1311, imax(-4096, 8192 - tf_estimate));
    u32 m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < len; i++) {
        const int width = band_w(i);
        const int N = width << LM, narrow = width == 1;
        const int metric = tf_band_metric_lane(X + tf_chan * FRAME + (CLT_eband5ms[i] << LM), mcol(F, 0), N, narrow, isTransient, bias);
        const u32 nib = (u32)(metric + 8) << (4 * (i & 7));
        if (i < 8) m0 |= nib; else if (i < 16) m1 |= nib; else m2 |= nib;
    }
    const i8 *tab = CLT_tf_select_table + LM * 8;
    int tf_select = 0, selcost0 = 0, selcost1 = 0;
    for (int sel = 0; sel < 2; sel++) {
        int cost0 = 0, cost1 = isTransient ? 0 : lambda;
        const int t0 = 2 * tab[4 * isTransient + 2 * sel + 0], t1 = 2 * tab[4 * isTransient + 2 * sel + 1];
        for (int i = 1; i < len; i++) {
            const int curr0 = imin(cost0, cost1 + lambda), curr1 = imin(cost0 + lambda, cost1);
            const int m = tf_nib(m0, m1, m2, i);
            cost0 = curr0 + iabs(m - t0);
            cost1 = curr1 + iabs(m - t1);
        }
        if (sel == 0) selcost0 = imin(cost0, cost1); else selcost1 = imin(cost0, cost1);
    }
    if (selcost1 < selcost0 && isTransient) tf_select = 1;
    int cost0 = 0, cost1 = isTransient ? 0 : lambda;
    u32 path0 = 0, path1 = 0;
    const int t0 = 2 * tab[4 * isTransient + 2 * tf_select + 0], t1 = 2 * tab[4 * isTransient + 2 * tf_select + 1];
    for (int i = 1; i < len; i++) {
        int curr0, curr1, from0 = cost0, from1 = cost1 + lambda;
        if (from0 < from1) curr0 = from0; else { curr0 = from1; path0 |= 1u << i; }
        from0 = cost0 + lambda;
        from1 = cost1;
        if (from0 < from1) curr1 = from0; else { curr1 = from1; path1 |= 1u << i; }
        const int m = tf_nib(m0, m1, m2, i);
        cost0 = curr0 + iabs(m - t0);
        cost1 = curr1 + iabs(m - t1);
    }
    u32 r = cost0 < cost1 ? 0u : 1u, res = r << (len - 1);
    for (int i = len - 2; i >= 0; i--) {
        r = ((r ? path1 : path0) >> (i + 1)) & 1u;
        res |= r << i;
    }
    F.tf_bits = res;
    return tf_select;
}

// tf_encode (celt_encoder.c:715-754); afterwards tf_change of band i = tab[F.tf_sel + bit i of F.tf_bits]
CA_DEVFN void tf_encode_lane(BackLds &F, RangeEnc &enc, int isTransient, int tf_select)
{
    const int LM = LM3;
    u32 budget = enc.storage * 8;
    u32 tell = (u32)ec_tell(enc);
    int logp = isTransient ? 2 : 4;
    const int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= tf_select_rsv;
    int curr = 0, tf_changed = 0;
    u32 res = 0;
    for (int i = 0; i < NB; i++) {
        int r = (int)((F.tf_bits >> i) & 1u);
        if (tell + logp <= budget) {
            ec_enc_bit_logp(enc, r ^ curr, logp);
            tell = (u32)ec_tell(enc);
            curr = r;
            tf_changed |= curr;
        } else {
            r = curr;
        }
        res |= (u32)r << i;
        logp = isTransient ? 4 : 5;
    }
    const i8 *tab = CLT_tf_select_table + LM * 8;
    if (tf_select_rsv && tab[4 * isTransient + 0 + tf_changed] != tab[4 * isTransient + 2 + tf_changed])
        ec_enc_bit_logp(enc, tf_select, 1);
    else
        tf_select = 0;
    F.tf_bits = res;
    F.tf_sel = 4 * isTransient + 2 * tf_select;
}

CA_DEV int tf_change_lane(const BackLds &F, int i) { return (CLT_tf_select_table + LM3 * 8)[F.tf_sel + (int)((F.tf_bits >> i) & 1u)]; }

// ---- quant_coarse_energy (quant_bands.c:158-367) ---------------------------------------------------------------------
// One pass of quant_coarse_energy_impl: reads oldE_in, writes oldE_out / err (oldE_out may be oldE_in: read before written).
CA_DEVFN int coarse_energy_pass_lane(Col eBands, Col oldE_in, Col oldE_out, Col err, RangeEnc &enc, i32 budget, i32 tell,
                                     const u8 *prob_model, int intra, i32 max_decay)
{
    const int LM = LM3, C = 2;
    int badness = 0;
    i32 prev0 = 0, prev1 = 0;
    i32 coef, beta;
    if (tell + 3 <= budget) ec_enc_bit_logp(enc, intra, 3);
    if (intra) { coef = 0; beta = 4915; }
    else { beta = CLT_beta_coef[LM]; coef = CLT_pred_coef[LM]; }
    for (int i = 0; i < NB; i++) {
        for (int c = 0; c < C; c++) {
            const i32 x = eBands[i + c * NB];
            const i32 oldraw = oldE_in[i + c * NB];
            const i32 oldEc = imax(-9216, oldraw);
            const i32 prevc = c == 0 ? prev0 : prev1;
            const i32 f = sub32(sub32(shl32(x, 7), pshr32(mul16_16(coef, oldEc), 8)), prevc);
            int qi = add32(f, 65536) >> 17;
            const i32 decay_bound = (i16)imax(-28672, sub32(oldraw, max_decay));
            if (qi < 0 && x < decay_bound) {
                qi += (int)(sub16(decay_bound, x) >> 10);
                if (qi > 0) qi = 0;
            }
            const int qi0 = qi;
            tell = ec_tell(enc);
            const int bits_left = budget - tell - 3 * C * (NB - i);
            if (i != 0 && bits_left < 30) {
                if (bits_left < 24) qi = imin(1, qi);
                if (bits_left < 16) qi = imax(-1, qi);
            }
            if (budget - tell >= 15) {
                const int pi = 2 * imin(i, 20);
                ec_laplace_encode(enc, qi, (u32)prob_model[pi] << 7, (int)prob_model[pi + 1] << 6);
            } else if (budget - tell >= 2) {
                qi = imax(-1, imin(qi, 1));
                ec_enc_icdf(enc, (2 * qi) ^ -(qi < 0), CLT_small_energy_icdf, 2);
            } else if (budget - tell >= 1) {
                qi = imin(0, qi);
                ec_enc_bit_logp(enc, -qi, 1);
            } else {
                qi = -1;
            }
            err[i + c * NB] = (i16)(pshr32(f, 7) - shl16(qi, 10));
            badness += iabs(qi0 - qi);
            const i32 q = shl32(qi, 10);
            i32 tmp = add32(add32(pshr32(mul16_16(coef, oldEc), 8), prevc), shl32(q, 7));
            tmp = imax(-3670016, tmp);
            oldE_out[i + c * NB] = (i16)pshr32(tmp, 7);
            const i32 pn = sub32(add32(prevc, shl32(q, 7)), mul16_16(beta, pshr32(q, 8)));
            if (c == 0) prev0 = pn; else prev1 = pn;
        }
    }
    return badness;
}

CA_DEVFN void quant_coarse_energy_lane(BackLds &F, FrameCtx &fc, RangeEnc &enc, u32 budget, int nbAvailableBytes, int two_pass,
                                       int loss_rate)
{
    const int C = 2, LM = LM3;
    const Col loge = mcol(F, M_LOGE), olde = mcol(F, M_OLDE), err = mcol(F, M_ERR), ea = mcol(F, M_EA), erra = mcol(F, M_ERRA),
              save = mcol(F, M_SAVE);
    int intra = (!two_pass && fc.delayedIntra > 2 * C * NB && nbAvailableBytes > NB * C);
    const i32 intra_bias = (i32)((budget * (u32)fc.delayedIntra * (u32)loss_rate) / (u32)(C * 512));
    i32 new_distortion;
    {   // loss_distortion (quant_bands.c:142-156)
        i32 p = 0;
        CA_UNROLL_LANE
        for (int k = 0; k < C * NB; k++) {
            const i32 d = (i16)sub16(loge[k] >> 3, olde[k] >> 3);
            p = mac16_16(p, d, d);
        }
        new_distortion = imin(200, p >> 14);
    }
    const u32 tell = (u32)ec_tell(enc);
    if (tell + 3 > budget) two_pass = intra = 0;
    i32 max_decay = 16384;
    max_decay = imin(max_decay, shl32(nbAvailableBytes, 7));
    const RangeEnc enc_start = enc;
    // Two trips through ONE inlined copy of the coder loop: pass 0 = intra (when two_pass || intra) into (ea, erra), pass 1 =
    // inter (when !intra) in place, with the coder snapshot / restore of quant_bands.c:304-357 in between.
    int badness1 = 0, badness2 = 0;
    i32 tell_intra = 0;
    RangeEnc enc_intra = enc;
    const u32 nstart_bytes = enc_start.offs;
    u32 save_bytes = 0;
    const int force_intra = intra;
    for (int pass = (two_pass || intra) ? 0 : 1; pass < (force_intra ? 1 : 2); pass++) {
        if (pass == 1) {
            tell_intra = (i32)ec_tell_frac(enc);
            enc_intra = enc;
            save_bytes = enc_intra.offs - nstart_bytes;
            // the bytes the intra pass emitted, two per slot: at most 3 + 42 x 15 bits = 80 bytes (ec_laplace_encode codes with
            // ft = 2^15, LAPLACE_MINP = 1), 96 are kept
            for (u32 k = 0; k < save_bytes && k < 2 * M_SAVE_SLOTS; k += 2)
                save[k >> 1] = (i16)((u32)enc.buf[nstart_bytes + k] | ((u32)enc.buf[nstart_bytes + k + 1] << 8));
            enc = enc_start;
        }
        const int bad = coarse_energy_pass_lane(loge, olde, pass == 0 ? ea : olde, pass == 0 ? erra : err, enc, (i32)budget, (i32)tell,
                                                CLT_e_prob_model + (LM * 2 + (pass == 0 ? 1 : 0)) * 42, pass == 0, max_decay);
        if (pass == 0) badness1 = bad; else badness2 = bad;
    }
    bool take_intra = force_intra;
    if (!force_intra && two_pass && (badness1 < badness2 || (badness1 == badness2 && (i32)ec_tell_frac(enc) + intra_bias > tell_intra))) {
        enc = enc_intra;
        CA_COUNT("lane.coarse_intra_restored", save_bytes);
        for (u32 k = 0; k < save_bytes && k < 2 * M_SAVE_SLOTS; k += 2) {
            const u32 w = (u16)save[k >> 1];
            enc.buf[nstart_bytes + k] = (u8)w;
            if (k + 1 < save_bytes) enc.buf[nstart_bytes + k + 1] = (u8)(w >> 8);
        }
        if (save_bytes > 2 * M_SAVE_SLOTS) enc.error = -1;
        intra = 1;
        take_intra = true;
    }
    if (force_intra) CA_COUNT("lane.coarse_intra_forced", 1);
    if (take_intra) {
        CA_UNROLL_LANE
        for (int k = 0; k < C * NB; k++) { olde[k] = ea[k]; err[k] = erra[k]; }
    }
    if (intra) fc.delayedIntra = new_distortion;
    else fc.delayedIntra = add32(mul16_32_q15((i16)mul16_16_q15(CLT_pred_coef[LM], CLT_pred_coef[LM]), fc.delayedIntra), new_distortion);
}

// ---- dynalloc_analysis (celt_encoder.c:932-1065) ------------------------------------------------------------------------
CA_DEV i32 median_of_5_col(Col x)
{
    i32 t0, t1, t2 = x[2], t3, t4;
    const i32 x0 = x[0], x1 = x[1], x3 = x[3], x4 = x[4];
    if (x0 > x1) { t0 = x1; t1 = x0; } else { t0 = x0; t1 = x1; }
    if (x3 > x4) { t3 = x4; t4 = x3; } else { t3 = x3; t4 = x4; }
    if (t0 > t3) { i32 a = t0; t0 = t3; t3 = a; a = t1; t1 = t4; t4 = a; }
    if (t2 > t1) return t1 < t3 ? imin(t2, t3) : imin(t4, t1);
    return t2 < t3 ? imin(t1, t3) : imin(t2, t4);
}
CA_DEV i32 median_of_3_col(Col x)
{
    i32 t0, t1, t2 = x[2];
    const i32 x0 = x[0], x1 = x[1];
    if (x0 > x1) { t0 = x1; t1 = x0; } else { t0 = x0; t1 = x1; }
    if (t1 < t2) return t1;
    if (t0 < t2) return t2;
    return t0;
}

// returns maxDepth; offsets -> column slots M_OFFSETS..
CA_DEVFN i32 dynalloc_analysis_lane(BackLds &F, int lsb_depth, int isTransient, int vbr, int constrained_vbr, int effectiveBytes,
                                    i32 *tot_boost_)
{
    const int C = 2, LM = LM3, end = NB;
    const Col loge = mcol(F, M_LOGE), fo = mcol(F, M_FOLLOWER), nf = mcol(F, M_NOISE), e2 = mcol(F, M_E2), off = mcol(F, M_OFFSETS);
    i32 tot_boost = 0;
    for (int i = 0; i < NB; i++) off[i] = 0;
    i32 maxDepth = -32666;
    for (int i = 0; i < end; i++)
        nf[i] = (i16)(mul16_16(64, CLT_logN400[i]) + 512 + shl16(9 - lsb_depth, 10) - shl16(CLT_eMeans[i], 6) + mul16_16(6, (i + 5) * (i + 5)));
    for (int c = 0; c < C; c++)
        for (int i = 0; i < end; i++) maxDepth = (i16)imax(maxDepth, loge[c * NB + i] - nf[i]);
    if (effectiveBytes > 50 && LM >= 1) {
        for (int k = 0; k < C * NB; k++) e2[k] = F.mid->bandLogE2[k];
        int last = 0;
        for (int c = 0; c < C; c++) {
            const Col f = fo + c * NB, E2 = e2 + c * NB;
            i32 fprev = E2[0], eprev = fprev;
            f[0] = (i16)fprev;
            for (int i = 1; i < end; i++) {
                const i32 e = E2[i];
                if (e > eprev + 512) last = i;
                fprev = imin(fprev + 1536, e);
                f[i] = (i16)fprev;
                eprev = e;
            }
            for (int i = last - 1; i >= 0; i--) f[i] = (i16)imin(f[i], imin(f[i + 1] + 2048, E2[i]));
            const i32 offset = 1024;
            for (int i = 2; i < end - 2; i++) f[i] = (i16)imax(f[i], median_of_5_col(E2 + (i - 2)) - offset);
            i32 tmp = median_of_3_col(E2) - offset;
            f[0] = (i16)imax(f[0], (i16)tmp);
            f[1] = (i16)imax(f[1], (i16)tmp);
            tmp = median_of_3_col(E2 + (end - 3)) - offset;
            f[end - 2] = (i16)imax(f[end - 2], (i16)tmp);
            f[end - 1] = (i16)imax(f[end - 1], (i16)tmp);
            for (int i = 0; i < end; i++) f[i] = (i16)imax(f[i], nf[i]);
        }
        for (int i = 0; i < end; i++) {
            const i32 a = (i16)imax(fo[NB + i], fo[i] - 4096);
            const i32 b = (i16)imax(fo[i], a - 4096);
            fo[NB + i] = (i16)a;
            fo[i] = (i16)((imax(0, loge[i] - b) + imax(0, loge[NB + i] - a)) >> 1);
        }
        if ((!vbr || constrained_vbr) && !isTransient)
            for (int i = 0; i < end; i++) fo[i] = (i16)(fo[i] >> 1);
        for (int i = 0; i < end; i++) {
            i32 v = fo[i];
            if (i < 8) v = (i16)(v * 2);
            if (i >= 12) v = (i16)(v >> 1);
            v = (i16)imin(v, 4096);
            const int width = (C * band_w(i)) << LM;
            int boost, boost_bits;
            if (width < 6) {
                boost = (int)(v >> 10);
                boost_bits = (boost * width) << 3;
            } else if (width > 48) {
                boost = (int)((v * 8) >> 10);
                boost_bits = ((boost * width) << 3) / 8;
            } else {
                boost = (int)((v * width / 6) >> 10);
                boost_bits = (boost * 6) << 3;
            }
            if ((!vbr || (constrained_vbr && !isTransient)) && ((tot_boost + boost_bits) >> 3 >> 3) > effectiveBytes / 4) {
                const i32 cap = (effectiveBytes / 4) << 3 << 3;
                off[i] = (i16)(cap - tot_boost);
                tot_boost = cap;
                break;
            } else {
                off[i] = (i16)boost;
                tot_boost += boost_bits;
            }
        }
    }
    *tot_boost_ = tot_boost;
    return maxDepth;
}

// alloc_trim_analysis's band-energy tilt (celt_encoder.c:808-817) on the column copy
CA_DEV i32 alloc_trim_diff_lane(BackLds &F)
{
    const Col loge = mcol(F, M_LOGE);
    i32 diff = 0;
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < NB - 1; i++) diff += loge[i + c * NB] * (i32)(2 + 2 * i - NB);
    return diff;
}

// ---- compute_allocation (rate.c:527-639) + interp_bits2pulses (rate.c:248-525), encode, start 0, end 21, C 2, LM 3 ------------
// Storage: bits1 / bits2 (32-bit, read seven times each by the interpolation search) as slot pairs; thresh / trim_offset / cap
// recomputed; the per-band budgets `bits` (non-negative, < 2^16 while they carry the spread-out remainder, <= cap at the end) in
// slots LS_PULSES..; ebits / fine_priority in slots M_FQ.. / M_FP.. (bits1 / bits2 are dead by then).
CA_DEVFN AllocOut compute_allocation_lane(BackLds &F, RangeEnc &ec, int alloc_trim, int intensity_in, int dual_stereo_in, i32 total,
                                          int prev, int signalBandwidth)
{
    const int C = 2, LM = LM3, end = NB, start = 0, len = NB;
    const i16 *eB = CLT_eband5ms;
    const Col off = mcol(F, M_OFFSETS), b1 = mcol(F, M_BITS1), b2 = mcol(F, M_BITS2), bits = mcol(F, LS_PULSES), ebits = mcol(F, M_FQ),
              fprio = mcol(F, M_FP);
    AllocOut out;
    out.intensity = intensity_in;
    out.dual_stereo = dual_stereo_in;
    total = imax(total, 0);
    int skip_start = start;
    const int skip_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
    total -= skip_rsv;
    int intensity_rsv = 0, dual_stereo_rsv = 0;
    intensity_rsv = CLT_log2_frac_table[end - start];
    if (intensity_rsv > total) intensity_rsv = 0;
    else {
        total -= intensity_rsv;
        dual_stereo_rsv = total >= 1 << BITRES ? 1 << BITRES : 0;
        total -= dual_stereo_rsv;
    }
    int lo = 1, hi = 11 - 1;
    do {
        int done = 0, psum = 0;
        const int mid = (lo + hi) >> 1;
        for (int j = end; j-- > start;) {
            const int N = eB[j + 1] - eB[j];
            int bitsj = (C * N * CLT_band_allocation[mid * len + j]) << LM >> 2;
            if (bitsj > 0) bitsj = imax(0, bitsj + alloc_trim_offset(j, alloc_trim));
            bitsj += off[j];
            if (bitsj >= alloc_thresh(j) || done) { done = 1; psum += imin(bitsj, alloc_cap(j)); }
            else if (bitsj >= C << BITRES) psum += C << BITRES;
        }
        if (psum > total) hi = mid - 1; else lo = mid + 1;
    } while (lo <= hi);
    hi = lo--;
    for (int j = start; j < end; j++) {
        const int N = eB[j + 1] - eB[j];
        const i32 to = alloc_trim_offset(j, alloc_trim), oj = off[j];
        int bits1j = (C * N * CLT_band_allocation[lo * len + j]) << LM >> 2;
        int bits2j = hi >= 11 ? alloc_cap(j) : (C * N * CLT_band_allocation[hi * len + j]) << LM >> 2;
        if (bits1j > 0) bits1j = imax(0, bits1j + to);
        if (bits2j > 0) bits2j = imax(0, bits2j + to);
        if (lo > 0) bits1j += oj;
        bits2j += oj;
        if (oj > 0) skip_start = j;
        bits2j = imax(0, bits2j - bits1j);
        col_set32(b1, j, bits1j);
        col_set32(b2, j, bits2j);
    }
    // ---- interp_bits2pulses ----
    const int alloc_floor = C << BITRES, stereo = 1, logM = LM << BITRES;
    i32 psum;
    lo = 0;
    hi = 1 << ALLOC_STEPS;
    for (int i = 0; i < ALLOC_STEPS; i++) {
        const int mid = (lo + hi) >> 1;
        int done = 0;
        psum = 0;
        for (int j = end; j-- > start;) {
            const int tmp = col_get32(b1, j) + ((mid * col_get32(b2, j)) >> ALLOC_STEPS);
            if (tmp >= alloc_thresh(j) || done) { done = 1; psum += imin(tmp, alloc_cap(j)); }
            else if (tmp >= alloc_floor) psum += alloc_floor;
        }
        if (psum > total) hi = mid; else lo = mid;
    }
    psum = 0;
    {
        int done = 0;
        for (int j = end; j-- > start;) {
            int tmp = col_get32(b1, j) + ((lo * col_get32(b2, j)) >> ALLOC_STEPS);
            if (tmp < alloc_thresh(j) && !done) tmp = tmp >= alloc_floor ? alloc_floor : 0;
            else done = 1;
            tmp = imin(tmp, alloc_cap(j));
            bits[j] = (i16)tmp;
            psum += tmp;
        }
    }
    int codedBands;
    for (codedBands = end;; codedBands--) {
        const int j = codedBands - 1;
        if (j <= skip_start) { total += skip_rsv; break; }
        i32 left = total - psum;
        const i32 percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
        left -= (eB[codedBands] - eB[start]) * percoeff;
        const int rem = imax(left - (eB[j] - eB[start]), 0);
        const int band_width = eB[codedBands] - eB[j];
        const i32 bj = (u16)bits[j];
        int band_bits = (int)(bj + percoeff * band_width + rem);
        if (band_bits >= imax(alloc_thresh(j), alloc_floor + (1 << BITRES))) {
            if (coder_bit_logp(ec, codedBands <= start + 2 || (band_bits > (((j < prev ? 7 : 9) * band_width) << LM << BITRES) >> 4 && j <= signalBandwidth), 1))
                break;
            psum += 1 << BITRES;
            band_bits -= 1 << BITRES;
        }
        psum -= bj + intensity_rsv;
        if (intensity_rsv > 0) intensity_rsv = CLT_log2_frac_table[j - start];
        psum += intensity_rsv;
        if (band_bits >= alloc_floor) { psum += alloc_floor; bits[j] = (i16)alloc_floor; }
        else bits[j] = 0;
    }
    if (intensity_rsv > 0) {
        out.intensity = imin(out.intensity, codedBands);
        out.intensity = start + (int)coder_uint(ec, (u32)(out.intensity - start), (u32)(codedBands + 1 - start));
    } else {
        out.intensity = 0;
    }
    if (out.intensity <= start) { total += dual_stereo_rsv; dual_stereo_rsv = 0; }
    if (dual_stereo_rsv > 0) out.dual_stereo = coder_bit_logp(ec, out.dual_stereo, 1);
    else out.dual_stereo = 0;
    i32 left = total - psum;
    const i32 percoeff = (u32)left / (u32)(eB[codedBands] - eB[start]);
    left -= (eB[codedBands] - eB[start]) * percoeff;
    // the remainder goes to the first bands, one bit per coefficient (rate.c:438-444), together with the per-coefficient share
    i32 balance = 0;
    int j;
    for (j = start; j < codedBands; j++) {
        const int N0 = eB[j + 1] - eB[j], N = N0 << LM;
        const int give = (int)imin(left, N0);
        left -= give;
        const i32 bit = (i32)(u16)bits[j] + (i32)percoeff * N0 + give + balance;
        i32 excess, bj, ej, fp;
        if (N > 1) {
            excess = imax(bit - alloc_cap(j), 0);
            bj = bit - excess;
            const int den = C * N + ((C == 2 && N > 2 && !out.dual_stereo && j < out.intensity) ? 1 : 0);
            const int NClogN = den * (CLT_logN400[j] + logM);
            int offset = (NClogN >> 1) - den * FINE_OFFSET;
            if (N == 2) offset += den << BITRES >> 2;
            if (bj + offset < (den * 2) << BITRES) offset += NClogN >> 2;
            else if (bj + offset < (den * 3) << BITRES) offset += NClogN >> 3;
            ej = imax(0, bj + offset + (den << (BITRES - 1)));
            ej = (i32)((u32)ej / (u32)den) >> BITRES;
            if (C * ej > (bj >> BITRES)) ej = bj >> stereo >> BITRES;
            ej = imin(ej, MAX_FINE_BITS);
            fp = ej * (den << BITRES) >= bj + offset;
            bj -= (C * ej) << BITRES;
        } else {
            excess = imax(0, bit - (C << BITRES));
            bj = bit - excess;
            ej = 0;
            fp = 1;
        }
        if (excess > 0) {
            const int extra_fine = imin(excess >> (stereo + BITRES), MAX_FINE_BITS - ej);
            ej += extra_fine;
            const int extra_bits = (extra_fine * C) << BITRES;
            fp = extra_bits >= excess - balance;
            excess -= extra_bits;
        }
        balance = excess;
        bits[j] = (i16)bj;
        ebits[j] = (i16)ej;
        fprio[j] = (i16)fp;
    }
    out.balance = balance;
    for (; j < end; j++) {
        const i32 e = (i32)(u16)bits[j] >> stereo >> BITRES;
        ebits[j] = (i16)e;
        bits[j] = 0;
        fprio[j] = (i16)(e < 1);
    }
    out.codedBands = codedBands;
    return out;
}

// quant_fine_energy (quant_bands.c:369-404) on the column
CA_DEVFN void quant_fine_energy_lane(BackLds &F, RangeEnc &enc)
{
    const Col ebits = mcol(F, M_FQ), olde = mcol(F, M_OLDE), err = mcol(F, M_ERR);
    for (int i = 0; i < NB; i++) {
        const int fq = ebits[i];
        if (fq <= 0) continue;
        const i32 frac = (i16)(1 << fq);
        for (int c = 0; c < 2; c++) {
            const i32 e = err[i + c * NB];
            int q2 = (e + 512) >> (10 - fq);
            if (q2 > frac - 1) q2 = frac - 1;
            if (q2 < 0) q2 = 0;
            ec_enc_bits(enc, (u32)q2, (u32)fq);
            const i32 offset = (i16)sub16((shl32(q2, 10) + 512) >> fq, 512);
            olde[i + c * NB] = (i16)(olde[i + c * NB] + offset);
            err[i + c * NB] = (i16)(e - offset);
        }
    }
}

// What quant_energy_finalise needs after the PVQ walk has taken the column: parked in the frame's hand-off record (fields this
// phase has finished reading). info[i] = fine_quant | fine_priority << 4 | (error[i] < 0) << 5 | (error[i + NB] < 0) << 6.
CA_DEV void park_energy_state_lane(BackLds &F)
{
    const Col ebits = mcol(F, M_FQ), fprio = mcol(F, M_FP), olde = mcol(F, M_OLDE), err = mcol(F, M_ERR);
    i16 *info = F.mid->bandLogE2, *pold = F.mid->oldBandE;
    for (int i = 0; i < NB; i++)
        info[i] = (i16)(ebits[i] | (fprio[i] << 4) | ((err[i] < 0) << 5) | ((err[i + NB] < 0) << 6));
    for (int k = 0; k < 2 * NB; k++) pold[k] = olde[k];
}

// quant_energy_finalise (quant_bands.c:406-439) on the parked state
CA_DEVFN void quant_energy_finalise_lane(BackLds &F, RangeEnc &enc, int bits_left)
{
    const int C = 2;
    const i16 *info = F.mid->bandLogE2;
    i16 *pold = F.mid->oldBandE;
    for (int prio = 0; prio < 2; prio++) {
        for (int i = 0; i < NB && bits_left >= C; i++) {
            const int w = info[i], fq = w & 15;
            if (fq >= MAX_FINE_BITS || ((w >> 4) & 1) != prio) continue;
            for (int c = 0; c < C; c++) {
                const int q2 = ((w >> (5 + c)) & 1) ? 0 : 1;
                ec_enc_bits(enc, (u32)q2, 1);
                const i32 offset = (i16)((shl16(q2, 10) - 512) >> (fq + 1));
                pold[i + c * NB] = (i16)(pold[i + c * NB] + offset);
                bits_left--;
            }
        }
    }
}

}  // namespace ca
