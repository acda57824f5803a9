// celt_enc_mid.h -- analysis decisions between normalisation and bit allocation.
//
// Wave-cooperative counterparts of:
//   tf_analysis / l1_metric / haar1      opus-fix/celt/celt_encoder.c:552-713, :541-550; celt/bands.c:581-594
//   tf_encode                            celt/celt_encoder.c:715-754
//   quant_coarse_energy(_impl)           celt/quant_bands.c:158-367 (+ loss_distortion :142-156)
//   spreading_decision                   celt/bands.c:428-510
//   dynalloc_analysis                    celt/celt_encoder.c:932-1065 (median_of_5/3 :875-930)
//   stereo_analysis                      celt/celt_encoder.c:840-873
//   alloc_trim_analysis                  celt/celt_encoder.c:756-838
//   compute_vbr                          celt/celt_encoder.c:1194-1322
//   hysteresis_decision                  celt/bands.c:48-63
#pragma once
#include "celt_enc_front.h"

namespace ca {

CA_DEV int iabs(int a) { return a < 0 ? -a : a; }

#if defined(CA_LANE_FRAME)
// Lane build: eight consecutive bins of this lane's X (16-byte aligned: every band starts at a multiple of eight bins
// and holds a multiple of eight) with ONE 16-byte load. A lane's row of X lies 4.7 KB from its neighbours', so every
// load instruction costs the wavefront one cache line per lane whatever its width.
CA_DEV void ld_bins8(const x16_t *p, i32 v[8])
{
    const v4i w = *reinterpret_cast<const CA_AS_GLB v4i *>(p);
    v[0] = (i16)w.x; v[1] = w.x >> 16; v[2] = (i16)w.y; v[3] = w.y >> 16;
    v[4] = (i16)w.z; v[5] = w.z >> 16; v[6] = (i16)w.w; v[7] = w.w >> 16;
}
// ... and the store of eight bins (values taken modulo 2^16)
CA_DEV void st_bins8(x16_t *p, const i32 v[8])
{
    v4i w;
    w.x = (i32)(((u32)v[0] & 0xffffu) | ((u32)v[1] << 16)); w.y = (i32)(((u32)v[2] & 0xffffu) | ((u32)v[3] << 16));
    w.z = (i32)(((u32)v[4] & 0xffffu) | ((u32)v[5] << 16)); w.w = (i32)(((u32)v[6] & 0xffffu) | ((u32)v[7] << 16));
    *reinterpret_cast<CA_AS_GLB v4i *>(p) = w;
}
#endif

// G butterflies of haar1, pair index p0 .. p0+G-1 of the flattened (j, i) space (stride = 1 << ls): all loads before the first store
template <int G, class P>
CA_DEV void haar1_group(P X, int p0, int ls, int stride)
{
    i32 a[G], b[G];
    int ia[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
        const int p = p0 + u;
        ia[u] = ((p >> ls) << (ls + 1)) + (p & (stride - 1));
        a[u] = X[ia[u]];
        b[u] = X[ia[u] + stride];
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
        i32 t1 = mul16_16(23170, a[u]), t2 = mul16_16(23170, b[u]);
        X[ia[u]] = (i16)pshr32(add32(t1, t2), 15);
        X[ia[u] + stride] = (i16)pshr32(sub32(t1, t2), 15);
    }
}

// haar1 on a vector in LDS (bands.c:581-594): N0 halved, `stride` interleaved sub-vectors; all pairs independent.
template <class P>
CA_DEV void haar1_wave(P X, int N0, int stride)
{
    N0 >>= 1;
    if (LANES == 1) {
        // one lane owns the frame. The N0 x stride butterflies are independent: walked as ONE index space in groups of eight
        // (sixteen loads, then sixteen stores), whatever the stride -- nested (i, j) loops in groups of four left the short
        // inner loops of the narrow bands (N0 = 2 at stride 2 of an 8-bin band) to a one-butterfly-per-round-trip tail.
        const int npairs = N0 * stride;
        if ((stride & (stride - 1)) == 0 && (npairs & 3) == 0) {
            int ls = 0;
            while ((1 << ls) < stride) ls++;
            int p0 = 0;
            for (; p0 + 8 <= npairs; p0 += 8) haar1_group<8>(X, p0, ls, stride);
            if (p0 < npairs) haar1_group<4>(X, p0, ls, stride);
            wave_sync();
            return;
        }
        for (int i = 0; i < stride; i++) {
            int j = 0;
            for (; j + 4 <= N0; j += 4) {
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
        wave_sync();
        return;
    }
    for (int k = lane(); k < stride * N0; k += LANES) {
        int i = k % stride, j = k / stride;
        i32 t1 = mul16_16(23170, X[stride * 2 * j + i]);
        i32 t2 = mul16_16(23170, X[stride * (2 * j + 1) + i]);
        X[stride * 2 * j + i] = (i16)pshr32(add32(t1, t2), 15);
        X[stride * (2 * j + 1) + i] = (i16)pshr32(sub32(t1, t2), 15);
    }
    wave_sync();
}

template <class P>
CA_DEV i32 l1_metric_wave(P tmp, int N, int LM, i32 bias)              // celt_encoder.c:541-550
{
    i32 p = 0;
    CA_UNROLL_LANE
    for (int i = lane(); i < N; i += LANES) { i32 v = tmp[i]; p += v < 0 ? -v : v; }
    i32 L1 = wave_add(p);
    return mac16_32_q15(L1, (i16)(LM * bias), L1);
}

// One band of tf_analysis (celt_encoder.c:585-640): the band's L1 metric at every time-frequency resolution, on copies
// of the band in tmp / tmp_1 (wherever the caller keeps them).
template <class PT, class PT1>
CA_DEV int tf_band_metric(const x16_t *Xb, PT tmp, PT1 tmp_1, int N, int narrow, int isTransient, i32 bias)
{
    const int LM = LM3;
#if defined(CA_LANE_FRAME)
#pragma unroll 4
    for (int j = 0; j < N; j += 8) {
        i32 v[8];
        ld_bins8(Xb + j, v);
#pragma unroll
        for (int u = 0; u < 8; u++) tmp[j + u] = (i16)v[u];
    }
#else
    CA_UNROLL_LANE
    for (int j = lane(); j < N; j += LANES) tmp[j] = Xb[j];
#endif
    wave_sync();
    i32 L1 = l1_metric_wave(tmp, N, isTransient ? LM : 0, bias);
    i32 best_L1 = L1;
    int best_level = 0;
    if (isTransient && !narrow) {
        CA_UNROLL_LANE
        for (int j = lane(); j < N; j += LANES) tmp_1[j] = tmp[j];
        wave_sync();
        haar1_wave(tmp_1, N >> LM, 1 << LM);
        L1 = l1_metric_wave(tmp_1, N, LM + 1, bias);
        if (L1 < best_L1) { best_L1 = L1; best_level = -1; }
    }
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

// tf_analysis(m, len = 21, isTransient, tf_res, lambda, X, N0 = 960, LM = 3, &tf_sum, tf_estimate, tf_chan)
template <class L>
CA_DEVFN int tf_analysis_wave(L &F, int isTransient, int lambda, i32 tf_estimate, int tf_chan)
{
    const int len = NB, LM = LM3;
    const x16_t *X = frame_X(F);
    i32 bias = (i16)mul16_16_q14(1311, imax(-4096, 8192 - tf_estimate));          // .04 Q15, -.25 Q14, .5 Q14
    for (int i = 0; i < len; i++) {
        const int width = CLT_eband5ms[i + 1] - CLT_eband5ms[i];
        const int N = width << LM, narrow = width == 1;
        const x16_t *Xb = X + tf_chan * FRAME + (CLT_eband5ms[i] << LM);
        int metric;
#if defined(CA_LANE_FRAME)
        // the band is analysed in the workgroup's LDS scratch ([element][lane], LANE_SCRATCH_N slots per lane); its second
        // copy (transient frames) too when both fit. The accessor TYPES differ (LDS column / private array), hence the
        // three instantiations.
        if (2 * N <= LANE_SCRATCH_N)
            metric = tf_band_metric(Xb, lds_col(F.lds_pvq16), lds_col(F.lds_pvq16) + LANE_SCRATCH_N / 2, N, narrow, isTransient, bias);
        else if (N <= LANE_SCRATCH_N)
            metric = tf_band_metric(Xb, lds_col(F.lds_pvq16), priv((i16 *)F.s.tf.tmp1), N, narrow, isTransient, bias);
        else
            metric = tf_band_metric(Xb, priv((i16 *)F.s.tf.tmp), priv((i16 *)F.s.tf.tmp1), N, narrow, isTransient, bias);
#else
        metric = tf_band_metric(Xb, F.s.tf.tmp, F.s.tf.tmp1, N, narrow, isTransient, bias);
#endif
        st0(&F.metric[i], metric);
        wave_sync();
    }
    // Viterbi over the 21 bands: uniform scalar code
    const i8 *tab = CLT_tf_select_table + LM * 8;
    int tf_select = 0, selcost0 = 0, selcost1 = 0;
    for (int sel = 0; sel < 2; sel++) {
        int cost0 = 0, cost1 = isTransient ? 0 : lambda;
        for (int i = 1; i < len; i++) {
            int curr0 = imin(cost0, cost1 + lambda), curr1 = imin(cost0 + lambda, cost1);
            cost0 = curr0 + iabs(uni(F.metric[i]) - 2 * tab[4 * isTransient + 2 * sel + 0]);
            cost1 = curr1 + iabs(uni(F.metric[i]) - 2 * tab[4 * isTransient + 2 * sel + 1]);
        }
        if (sel == 0) selcost0 = imin(cost0, cost1); else selcost1 = imin(cost0, cost1);
    }
    if (selcost1 < selcost0 && isTransient) tf_select = 1;
    int cost0 = 0, cost1 = isTransient ? 0 : lambda;
    for (int i = 1; i < len; i++) {
        int curr0, curr1, from0 = cost0, from1 = cost1 + lambda;
        if (from0 < from1) { curr0 = from0; st0(&F.path0[i], 0); } else { curr0 = from1; st0(&F.path0[i], 1); }
        from0 = cost0 + lambda;
        from1 = cost1;
        if (from0 < from1) { curr1 = from0; st0(&F.path1[i], 0); } else { curr1 = from1; st0(&F.path1[i], 1); }
        cost0 = curr0 + iabs(uni(F.metric[i]) - 2 * tab[4 * isTransient + 2 * tf_select + 0]);
        cost1 = curr1 + iabs(uni(F.metric[i]) - 2 * tab[4 * isTransient + 2 * tf_select + 1]);
    }
    wave_sync();
    int r = cost0 < cost1 ? 0 : 1;
    st0(&F.tf_res[len - 1], r);
    for (int i = len - 2; i >= 0; i--) {
        r = r == 1 ? uni(F.path1[i + 1]) : uni(F.path0[i + 1]);
        st0(&F.tf_res[i], r);
    }
    wave_sync();
    return tf_select;
}

template <class L>
CA_DEVFN void tf_encode_wave(L &F, RangeEnc &enc, int isTransient, int tf_select)   // celt_encoder.c:715-754
{
    const int LM = LM3;
    u32 budget = enc.storage * 8;
    u32 tell = (u32)ec_tell(enc);
    int logp = isTransient ? 2 : 4;
    int tf_select_rsv = LM > 0 && tell + logp + 1 <= budget;
    budget -= tf_select_rsv;
    int curr = 0, tf_changed = 0;
    i32 *res = F.path0;                               // scratch copy of the (possibly overridden) decisions
    for (int i = 0; i < NB; i++) {
        int r = uni(F.tf_res[i]);
        if (tell + logp <= budget) {
            ec_enc_bit_logp(enc, r ^ curr, logp);
            tell = (u32)ec_tell(enc);
            curr = r;
            tf_changed |= curr;
        } else {
            r = curr;
        }
        st0(&res[i], r);
        logp = isTransient ? 4 : 5;
    }
    const i8 *tab = CLT_tf_select_table + LM * 8;
    if (tf_select_rsv && tab[4 * isTransient + 0 + tf_changed] != tab[4 * isTransient + 2 + tf_changed])
        ec_enc_bit_logp(enc, tf_select, 1);
    else
        tf_select = 0;
    wave_sync();
    CA_UNROLL_LANE
    for (int i = lane(); i < NB; i += LANES) F.tf_res[i] = (i32)tab[4 * isTransient + 2 * tf_select + res[i]];
    wave_sync();
}

// quant_coarse_energy_impl (quant_bands.c:158-267). eBands = F.bandLogE; oldE/err are LDS arrays written
// by lane 0 only (each element is read before it is written within one pass).
CA_DEVFN int coarse_energy_impl(const i16 *eBands, i16 *oldE, i16 *err, RangeEnc &enc, i32 budget, i32 tell,
                                const u8 *prob_model, int C, int intra, i32 max_decay)
{
    const int LM = LM3;
    int badness = 0;
    i32 prev0 = 0, prev1 = 0;
    i32 coef, beta;
    if (tell + 3 <= budget) ec_enc_bit_logp(enc, intra, 3);
    if (intra) { coef = 0; beta = 4915; }
    else { beta = CLT_beta_coef[LM]; coef = CLT_pred_coef[LM]; }
    for (int i = 0; i < NB; i++) {
        for (int c = 0; c < C; c++) {
            i32 x = uni((i32)eBands[i + c * NB]);
            i32 oldraw = uni((i32)oldE[i + c * NB]);
            i32 oldEc = imax(-9216, oldraw);                                      // -QCONST16(9.f,DB_SHIFT)
            const i32 prevc = c == 0 ? prev0 : prev1;
            i32 f = sub32(sub32(shl32(x, 7), pshr32(mul16_16(coef, oldEc), 8)), prevc);
            int qi = add32(f, 65536) >> 17;                                       // QCONST32(.5f,DB_SHIFT+7)
            i32 decay_bound = (i16)imax(-28672, sub32(oldraw, max_decay));
            if (qi < 0 && x < decay_bound) {
                qi += (int)(sub16(decay_bound, x) >> 10);                          // SHR16(SUB16()) is plain int
                if (qi > 0) qi = 0;
            }
            int qi0 = qi;
            tell = ec_tell(enc);
            int bits_left = budget - tell - 3 * C * (NB - i);
            if (i != 0 && bits_left < 30) {
                if (bits_left < 24) qi = imin(1, qi);
                if (bits_left < 16) qi = imax(-1, qi);
            }
            if (budget - tell >= 15) {
                int pi = 2 * imin(i, 20);
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
            st0(&err[i + c * NB], (i16)(pshr32(f, 7) - shl16(qi, 10)));
            badness += iabs(qi0 - qi);
            i32 q = shl32(qi, 10);
            i32 tmp = add32(add32(pshr32(mul16_16(coef, oldEc), 8), prevc), shl32(q, 7));
            tmp = imax(-3670016, tmp);                                            // -QCONST32(28.f,DB_SHIFT+7)
            st0(&oldE[i + c * NB], (i16)pshr32(tmp, 7));
            const i32 pn = sub32(add32(prevc, shl32(q, 7)), mul16_16(beta, pshr32(q, 8)));
            if (c == 0) prev0 = pn; else prev1 = pn;
        }
    }
    return badness;
}

// quant_coarse_energy (quant_bands.c:269-367) with start 0, end = effEnd = 21, force_intra 0, lfe 0.
template <class L>
CA_DEVFN void quant_coarse_energy_wave(L &F, FrameCtx &fc, RangeEnc &enc, u32 budget, int nbAvailableBytes,
                                       int two_pass, int loss_rate)
{
    const int C = fc.C, LM = LM3;
    int intra = (!two_pass && fc.delayedIntra > 2 * C * NB && nbAvailableBytes > NB * C);
    i32 intra_bias = (i32)((budget * (u32)fc.delayedIntra * (u32)loss_rate) / (u32)(C * 512));
    i32 new_distortion;
    {   // loss_distortion (quant_bands.c:142-156)
        i32 p = 0;
        CA_UNROLL_LANE
        for (int k = lane(); k < C * NB; k += LANES) {
            i32 d = (i16)sub16(F.bandLogE[k] >> 3, F.oldBandE[k] >> 3);
            p = mac16_16(p, d, d);
        }
        new_distortion = imin(200, wave_add(p) >> 14);
    }
    u32 tell = (u32)ec_tell(enc);
    if (tell + 3 > budget) two_pass = intra = 0;
    i32 max_decay = 16384;
    max_decay = imin(max_decay, shl32(nbAvailableBytes, 7));                      // end-start > 10
    RangeEnc enc_start = enc;
    CA_UNROLL_LANE
    for (int k = lane(); k < C * NB; k += LANES) F.oldE_intra[k] = F.oldBandE[k];
    wave_sync();
    // Two trips through ONE inlined copy of the coder loop: pass 0 = intra (when two_pass || intra),
    // pass 1 = inter (when !intra), with the coder snapshot/restore of quant_bands.c:304-357 in between.
    int badness1 = 0, badness2 = 0;
    i32 tell_intra = 0;
    RangeEnc enc_intra = enc;
    u32 nstart_bytes = enc_start.offs, save_bytes = 0;
    const int force_intra = intra;
    for (int pass = (two_pass || intra) ? 0 : 1; pass < (force_intra ? 1 : 2); pass++) {
        if (pass == 1) {
            tell_intra = (i32)ec_tell_frac(enc);
            enc_intra = enc;
            save_bytes = enc_intra.offs - nstart_bytes;
            // save the bytes the intra pass emitted (256 bytes hold 42 symbols with margin)
            for (u32 k = lane(); k < save_bytes && k < 256; k += LANES) F.coarse_save[k] = enc.buf[nstart_bytes + k];
            wave_sync();
            enc = enc_start;
        }
        int bad = coarse_energy_impl(F.bandLogE, pass == 0 ? F.oldE_intra : F.oldBandE, pass == 0 ? F.error_intra : F.error,
                                     enc, (i32)budget, (i32)tell, CLT_e_prob_model + (LM * 2 + (pass == 0 ? 1 : 0)) * 42,
                                     C, pass == 0, max_decay);
        if (pass == 0) badness1 = bad; else badness2 = bad;
        wave_sync();
    }
    if (!force_intra) {
        if (two_pass && (badness1 < badness2 || (badness1 == badness2 && (i32)ec_tell_frac(enc) + intra_bias > tell_intra))) {
            enc = enc_intra;
            for (u32 k = lane(); k < save_bytes && k < 256; k += LANES) enc.buf[nstart_bytes + k] = F.coarse_save[k];
            CA_UNROLL_LANE
            for (int k = lane(); k < C * NB; k += LANES) { F.oldBandE[k] = F.oldE_intra[k]; F.error[k] = F.error_intra[k]; }
            if (save_bytes > 256) enc.error = -1;
            intra = 1;
        }
    } else {
        CA_UNROLL_LANE
        for (int k = lane(); k < C * NB; k += LANES) { F.oldBandE[k] = F.oldE_intra[k]; F.error[k] = F.error_intra[k]; }
    }
    wave_sync();
    if (intra) fc.delayedIntra = new_distortion;
    else fc.delayedIntra = add32(mul16_32_q15((i16)mul16_16_q15(CLT_pred_coef[LM], CLT_pred_coef[LM]), fc.delayedIntra), new_distortion);
}

// spreading_decision (bands.c:428-510) with end = 21, M = 8
template <class L>
CA_DEVFN int spreading_decision_wave(L &F, FrameCtx &fc, int update_hf)
{
    const int C = fc.C, M = M8, end = NB;
    const x16_t *X = frame_X(F);
    int sum = 0, nbBands = 0, hf_sum = 0;
    // M*(eBands[end]-eBands[end-1]) = 176 > 8: never SPREAD_NONE by width
    for (int c = 0; c < C; c++) {
        for (int i = 0; i < end; i++) {
            const int N = M * (CLT_eband5ms[i + 1] - CLT_eband5ms[i]);
            if (N <= 8) continue;
            const x16_t *x = X + M * CLT_eband5ms[i] + c * FRAME;
            i32 t = 0;                                                            // three 10-bit counters
#if defined(CA_LANE_FRAME)
#pragma unroll 4
            for (int j = 0; j < N; j += 8) {
                i32 v[8];
                ld_bins8(x + j, v);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    i32 x2N = mul16_16(mul16_16_q15(v[u], v[u]), N);
                    if (x2N < 2048) t += 1;
                    if (x2N < 512) t += 1 << 10;
                    if (x2N < 128) t += 1 << 20;
                }
            }
#else
            CA_UNROLL_LANE
            for (int j = lane(); j < N; j += LANES) {
                i32 x2N = mul16_16(mul16_16_q15(x[j], x[j]), N);
                if (x2N < 2048) t += 1;
                if (x2N < 512) t += 1 << 10;
                if (x2N < 128) t += 1 << 20;
            }
#endif
            t = wave_add(t);
            int t0 = t & 1023, t1 = (t >> 10) & 1023, t2 = (t >> 20) & 1023;
            if (i > NB - 4) hf_sum += (32 * (t1 + t0)) / N;
            int tmp = (2 * t2 >= N) + (2 * t1 >= N) + (2 * t0 >= N);
            sum += tmp * 256;
            nbBands++;
        }
    }
    if (update_hf) {
        if (hf_sum) hf_sum = hf_sum / (C * (4 - NB + end));
        fc.hf_average = (fc.hf_average + hf_sum) >> 1;
        hf_sum = fc.hf_average;
        if (fc.tapset_decision == 2) hf_sum += 4;
        else if (fc.tapset_decision == 0) hf_sum -= 4;
        if (hf_sum > 22) fc.tapset_decision = 2;
        else if (hf_sum > 18) fc.tapset_decision = 1;
        else fc.tapset_decision = 0;
    }
    sum = sum / nbBands;
    sum = (sum + fc.tonal_average) >> 1;
    fc.tonal_average = sum;
    sum = (3 * sum + (((3 - fc.spread_decision) << 7) + 64) + 2) >> 2;
    if (sum < 80) return SPREAD_AGGRESSIVE;
    if (sum < 256) return SPREAD_NORMAL;
    if (sum < 384) return SPREAD_LIGHT;
    return SPREAD_NONE;
}

CA_DEV i32 median_of_5(const i16 *x)                                              // celt_encoder.c:875-912
{
    i32 t0, t1, t2 = x[2], t3, t4;
    if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
    if (x[3] > x[4]) { t3 = x[4]; t4 = x[3]; } else { t3 = x[3]; t4 = x[4]; }
    if (t0 > t3) { i32 a = t0; t0 = t3; t3 = a; a = t1; t1 = t4; t4 = a; }
    if (t2 > t1) return t1 < t3 ? imin(t2, t3) : imin(t4, t1);
    return t2 < t3 ? imin(t1, t3) : imin(t2, t4);
}

CA_DEV i32 median_of_3(const i16 *x)                                              // celt_encoder.c:914-930
{
    i32 t0, t1, t2 = x[2];
    if (x[0] > x[1]) { t0 = x[1]; t1 = x[0]; } else { t0 = x[0]; t1 = x[1]; }
    if (t1 < t2) return t1;
    if (t0 < t2) return t2;
    return t0;
}

// dynalloc_analysis (celt_encoder.c:932-1065), start 0, end 21, lfe 0, surround_dynalloc all zero.
// Small sequential recurrences over 21 bands: run on lane 0, results published through LDS.
template <class L>
CA_DEVFN i32 dynalloc_analysis_wave(L &F, const FrameCtx &fc, int lsb_depth, int isTransient, int vbr,
                                    int constrained_vbr, int effectiveBytes, i32 *tot_boost_)
{
    const int C = fc.C, LM = LM3, end = NB;
    if (lane() == 0) {
        i32 tot_boost = 0;
        for (int i = 0; i < NB; i++) F.offsets[i] = 0;
        i32 maxDepth = -32666;                                                     // -QCONST16(31.9f,DB_SHIFT)
        for (int i = 0; i < end; i++)
            F.noise_floor[i] = (i16)(mul16_16(64, CLT_logN400[i]) + 512 + shl16(9 - lsb_depth, 10)
                                     - shl16(CLT_eMeans[i], 6) + mul16_16(6, (i + 5) * (i + 5)));
        for (int c = 0; c < C; c++)
            for (int i = 0; i < end; i++) maxDepth = (i16)imax(maxDepth, F.bandLogE[c * NB + i] - F.noise_floor[i]);
        if (effectiveBytes > 50 && LM >= 1) {
            int last = 0;
            for (int c = 0; c < C; c++) {
                i16 *f = &F.follower[c * NB];
                const i16 *E2 = &F.bandLogE2[c * NB];
                f[0] = E2[0];
                for (int i = 1; i < end; i++) {
                    if (E2[i] > E2[i - 1] + 512) last = i;
                    f[i] = (i16)imin(f[i - 1] + 1536, E2[i]);
                }
                for (int i = last - 1; i >= 0; i--) f[i] = (i16)imin(f[i], imin(f[i + 1] + 2048, E2[i]));
                const i32 offset = 1024;
                for (int i = 2; i < end - 2; i++) f[i] = (i16)imax(f[i], median_of_5(&E2[i - 2]) - offset);
                i32 tmp = median_of_3(&E2[0]) - offset;
                f[0] = (i16)imax(f[0], (i16)tmp);
                f[1] = (i16)imax(f[1], (i16)tmp);
                tmp = median_of_3(&E2[end - 3]) - offset;
                f[end - 2] = (i16)imax(f[end - 2], (i16)tmp);
                f[end - 1] = (i16)imax(f[end - 1], (i16)tmp);
                for (int i = 0; i < end; i++) f[i] = (i16)imax(f[i], F.noise_floor[i]);
            }
            i16 *fo = F.follower;
            if (C == 2) {
                for (int i = 0; i < end; i++) {
                    fo[NB + i] = (i16)imax(fo[NB + i], fo[i] - 4096);
                    fo[i] = (i16)imax(fo[i], fo[NB + i] - 4096);
                    fo[i] = (i16)((imax(0, F.bandLogE[i] - fo[i]) + imax(0, F.bandLogE[NB + i] - fo[NB + i])) >> 1);
                }
            } else {
                for (int i = 0; i < end; i++) fo[i] = (i16)imax(0, F.bandLogE[i] - fo[i]);
            }
            // surround_dynalloc is all zero: follower = MAX16(follower, 0) is the identity after the lines above
            if ((!vbr || constrained_vbr) && !isTransient)
                for (int i = 0; i < end; i++) fo[i] = (i16)(fo[i] >> 1);
            for (int i = 0; i < end; i++) {
                if (i < 8) fo[i] = (i16)(fo[i] * 2);
                if (i >= 12) fo[i] = (i16)(fo[i] >> 1);
                fo[i] = (i16)imin(fo[i], 4096);
                int width = (C * (CLT_eband5ms[i + 1] - CLT_eband5ms[i])) << LM;
                int boost, boost_bits;
                if (width < 6) {
                    boost = (int)((i32)fo[i] >> 10);
                    boost_bits = (boost * width) << 3;
                } else if (width > 48) {
                    boost = (int)(((i32)fo[i] * 8) >> 10);
                    boost_bits = ((boost * width) << 3) / 8;
                } else {
                    boost = (int)(((i32)fo[i] * width / 6) >> 10);
                    boost_bits = (boost * 6) << 3;
                }
                if ((!vbr || (constrained_vbr && !isTransient)) && ((tot_boost + boost_bits) >> 3 >> 3) > effectiveBytes / 4) {
                    i32 cap = (effectiveBytes / 4) << 3 << 3;
                    F.offsets[i] = cap - tot_boost;
                    tot_boost = cap;
                    break;
                } else {
                    F.offsets[i] = boost;
                    tot_boost += boost_bits;
                }
            }
        }
        F.scal[8] = tot_boost;
        F.scal[9] = maxDepth;
    }
    wave_sync();
    *tot_boost_ = F.scal[8];
    i32 md = F.scal[9];
    wave_sync();
    return md;
}

// stereo_analysis (celt_encoder.c:840-873), LM = 3
template <class L>
CA_DEVFN int stereo_analysis_wave(L &F)
{
    const x16_t *X = frame_X(F);
    i32 pLR = 0, pMS = 0;
    const int jend = CLT_eband5ms[13] << LM3;
#if defined(CA_LANE_FRAME)
#pragma unroll 4
    for (int j = 0; j < jend; j += 8) {
        i32 lv[8], rv[8];
        ld_bins8(X + j, lv);
        ld_bins8(X + FRAME + j, rv);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i32 Lv = lv[u], R = rv[u];
            const i32 Mi = add32(Lv, R), S = sub32(Lv, R);
            pLR = add32(pLR, add32(Lv < 0 ? -Lv : Lv, R < 0 ? -R : R));
            pMS = add32(pMS, add32(Mi < 0 ? -Mi : Mi, S < 0 ? -S : S));
        }
    }
#else
    CA_UNROLL_LANE
    for (int j = lane(); j < jend; j += LANES) {
        i32 Lv = X[j], R = X[FRAME + j];
        i32 Mi = add32(Lv, R), S = sub32(Lv, R);
        pLR = add32(pLR, add32(Lv < 0 ? -Lv : Lv, R < 0 ? -R : R));
        pMS = add32(pMS, add32(Mi < 0 ? -Mi : Mi, S < 0 ? -S : S));
    }
#endif
    i32 sumLR = add32(1, wave_add(pLR)), sumMS = add32(1, wave_add(pMS));
    sumMS = mul16_32_q15(23170, sumMS);                                            // QCONST16(0.707107f,15)
    int thetas = 13;
    return mul16_32_q15((i16)((CLT_eband5ms[13] << (LM3 + 1)) + thetas), sumMS)
         > mul16_32_q15((i16)(CLT_eband5ms[13] << (LM3 + 1)), sumLR);
}

CA_DEV int hysteresis_decision(i32 val, const i16 *thresholds, const i16 *hysteresis, int N, int prev)   // bands.c:48-63
{
    int i;
    for (i = 0; i < N; i++)
        if (val < thresholds[i]) break;
    if (i > prev && val < thresholds[prev] + hysteresis[prev]) i = prev;
    if (i < prev && val > thresholds[prev - 1] - hysteresis[prev - 1]) i = prev;
    return i;
}

// alloc_trim_analysis (celt_encoder.c:756-838), end 21, LM 3, no float analysis, surround_trim 0
template <class L>
CA_DEVFN int alloc_trim_analysis_wave(L &F, FrameCtx &fc, i32 tf_estimate, int intensity)
{
    const int C = fc.C, LM = LM3, end = NB;
    const x16_t *X = frame_X(F);
    i32 diff = 0;
    i32 trim = 1280;                                                                // QCONST16(5.f,8)
    if (C == 2) {
        i32 sum = 0, minXC;
        for (int i = 0; i < 8; i++) {
            const int j0 = CLT_eband5ms[i] << LM, n = (CLT_eband5ms[i + 1] - CLT_eband5ms[i]) << LM;
            i32 p = 0;
#if defined(CA_LANE_FRAME)
#pragma unroll 4
            for (int j = 0; j < n; j += 8) {
                i32 lv[8], rv[8];
                ld_bins8(X + j0 + j, lv);
                ld_bins8(X + FRAME + j0 + j, rv);
#pragma unroll
                for (int u = 0; u < 8; u++) p = mac16_16(p, lv[u], rv[u]);
            }
#else
            CA_UNROLL_LANE
            for (int j = lane(); j < n; j += LANES) p = mac16_16(p, X[j0 + j], X[FRAME + j0 + j]);
#endif
            sum = add16(sum, (i16)(wave_add(p) >> 18));
        }
        sum = (i16)mul16_16_q15(4096, sum);                                         // QCONST16(1.f/8,15)
        sum = imin(1024, sum < 0 ? -sum : sum);
        minXC = sum;
        for (int i = 8; i < intensity; i++) {
            const int j0 = CLT_eband5ms[i] << LM, n = (CLT_eband5ms[i + 1] - CLT_eband5ms[i]) << LM;
            i32 p = 0;
#if defined(CA_LANE_FRAME)
#pragma unroll 4
            for (int j = 0; j < n; j += 8) {
                i32 lv[8], rv[8];
                ld_bins8(X + j0 + j, lv);
                ld_bins8(X + FRAME + j0 + j, rv);
#pragma unroll
                for (int u = 0; u < 8; u++) p = mac16_16(p, lv[u], rv[u]);
            }
#else
            CA_UNROLL_LANE
            for (int j = lane(); j < n; j += LANES) p = mac16_16(p, X[j0 + j], X[FRAME + j0 + j]);
#endif
            i32 v = (i16)(wave_add(p) >> 18);
            minXC = imin(minXC, v < 0 ? -v : v);
        }
        minXC = imin(1024, minXC < 0 ? -minXC : minXC);
        i32 logXC = celt_log2(sub32(1049625, mul16_16(sum, sum)));                  // QCONST32(1.001f,20)
        i32 logXC2 = imax(logXC >> 1, celt_log2(sub32(1049625, mul16_16(minXC, minXC))));
        logXC = (i16)pshr32(logXC - 6144, 2);
        logXC2 = (i16)pshr32(logXC2 - 6144, 2);
        trim = (i16)(trim + imax(-1024, mul16_16_q15(24576, logXC)));
        fc.stereo_saving = (i16)imin(fc.stereo_saving + 64, -(logXC2 >> 1));
    }
#if defined(CA_LANE_FRAME)
    diff = alloc_trim_diff_lane(F);                                                 // celt_enc_lane.h: bandLogE lives in the lane's column
#else
    for (int c = 0; c < C; c++)
        for (int i = 0; i < end - 1; i++) diff += F.bandLogE[i + c * NB] * (i32)(2 + 2 * i - end);
#endif
    diff /= C * (end - 1);
    trim = (i16)(trim - imax(-512, imin(512, ((diff + 1024) >> 2) / 6)));
    // surround_trim = 0
    trim = (i16)(trim - 2 * ((i16)tf_estimate >> 6));
    int trim_index = pshr32(trim, 8);
    return imax(0, imin(10, trim_index));
}

// compute_vbr (celt_encoder.c:1194-1322): no float analysis, no surround mask, lfe 0
CA_DEVFN i32 compute_vbr_wave(const FrameCtx &fc, i32 base_target, i32 bitrate, int constrained_vbr, i32 tot_boost,
                              i32 tf_estimate, i32 maxDepth, i32 temporal_vbr)
{
    const int C = fc.C, LM = LM3;
    int coded_bands = fc.lastCodedBands ? fc.lastCodedBands : NB;
    int coded_bins = CLT_eband5ms[coded_bands] << LM;
    if (C == 2) coded_bins += CLT_eband5ms[imin(fc.intensity, coded_bands)] << LM;
    i32 target = base_target;
    if (C == 2) {
        int coded_stereo_bands = imin(fc.intensity, coded_bands);
        int coded_stereo_dof = (CLT_eband5ms[coded_stereo_bands] << LM) - coded_stereo_bands;
        i32 max_frac = (i16)(mul16_16(26214, coded_stereo_dof) / (i16)coded_bins);  // DIV32_16(MULT16_16(.8 Q15, dof), bins)
        i32 stereo_saving = imin(fc.stereo_saving, 256);
        target -= imin(mul16_32_q15(max_frac, target),
                       mul16_16((i16)(stereo_saving - 26), (i16)(coded_stereo_dof << 3)) >> 8);
    }
    target += tot_boost - (16 << LM);
    i32 tf_calibration = 655;                                                       // QCONST16(0.04f,14)
    target += shl32(mul16_32_q15((i16)(tf_estimate - tf_calibration), target), 1);
    {
        int bins = CLT_eband5ms[NB - 2] << LM;
        i32 floor_depth = mul16_16((i16)((C * bins) << 3), maxDepth) >> 10;
        floor_depth = imax(floor_depth, target >> 2);
        target = imin(target, floor_depth);
    }
    if (constrained_vbr || bitrate < 64000) {
        i32 rate_factor = imax(0, bitrate - 32000);
        if (constrained_vbr) rate_factor = imin(rate_factor, 21955);                // QCONST16(0.67f,15)
        target = base_target + mul16_32_q15((i16)rate_factor, target - base_target);
    }
    if (tf_estimate < 3277) {                                                       // QCONST16(.2f,14)
        i32 amount = (i16)mul16_16_q15(3329, imax(0, imin(32000, 96000 - bitrate)));   // QCONST16(.0000031f,30)
        i32 tvbr_factor = (i16)(mul16_16(temporal_vbr, amount) >> 10);
        target += mul16_32_q15(tvbr_factor, target);
    }
    target = imin(2 * base_target, target);
    return target;
}

}  // namespace ca
