// silk_shape_dev.h -- silk_noise_shape_analysis_FIX (opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146-466): the noise-shaping
// filters, the initial subframe gains and the tilt / harmonic-shaping controls of a SILK frame (fifth slice).
//
//   silk_noise_shape_analysis_FIX      opus-fix/silk/fixed/noise_shape_analysis_FIX.c:146-466
//   warped_gain / limit_warped_coefs   opus-fix/silk/fixed/noise_shape_analysis_FIX.c:41-136
//   silk_warped_autocorrelation_FIX    opus-fix/silk/fixed/warped_autocorrelation_FIX.c:44-93
//   silk_autocorr -> _celt_autocorr    opus-fix/silk/fixed/autocorr_FIX.c:35-48, opus-fix/celt/celt_lpc.c:232-328 (overlap == 0)
//   silk_apply_sine_window             opus-fix/silk/fixed/apply_sine_window_FIX.c:51-101
//   silk_schur64                       opus-fix/silk/fixed/schur64_FIX.c:36-92
//   silk_k2a_Q16                       opus-fix/silk/fixed/k2a_Q16_FIX.c:35-53
//   silk_LPC_inverse_pred_gain_Q24     opus-fix/silk/LPC_inv_pred_gain.c:136-153 (+ :41-106)
//
// One lane owns one frame. Constants are the values SILK_FIX_CONST yields for the reference's tuning parameters (several are
// single-precision literals: the values below were printed by a probe compiled against the reference's headers).
#pragma once
#include "silk_gains_dev.h"

namespace ca {

enum { MAX_SHAPE_LPC_ORDER = 16 };

// round(65536 * pi / (L + 1)) for window lengths L = 16, 20, ..., 120  (apply_sine_window_FIX.c:45-48; tests/test_tables.py
// re-derives it and compares with the compiled reference)
CA_DEVICE_CONST i16 SILK_sine_window_freq_table_Q16[27] = {
    12111, 9804, 8235, 7100, 6239, 5565, 5022, 4575, 4202, 3885, 3612, 3375, 3167, 2984, 2820, 2674, 2542, 2422,
    2313, 2214, 2123, 2038, 1961, 1889, 1822, 1760, 1702,
};

// apply_sine_window_FIX.c:51-101; out / in are sample accessors
template <class OUT, class XA>
CA_DEV void silk_apply_sine_window_dev(OUT px_win, XA px, int win_type, int length)
{
    const int f_Q16 = SILK_sine_window_freq_table_Q16[(length >> 2) - 4];
    const i32 c_Q16 = s_smulwb((i32)f_Q16, -f_Q16);
    i32 S0_Q16, S1_Q16;
    if (win_type == 1) {
        S0_Q16 = 0;
        S1_Q16 = f_Q16 + (length >> 3);
    } else {
        S0_Q16 = (i32)1 << 16;
        S1_Q16 = ((i32)1 << 16) + (c_Q16 >> 1) + (length >> 4);
    }
    for (int k = 0; k < length; k += 4) {
        px_win[k] = (i16)s_smulwb((S0_Q16 + S1_Q16) >> 1, (i32)px[k]);
        px_win[k + 1] = (i16)s_smulwb(S1_Q16, (i32)px[k + 1]);
        S0_Q16 = s_smulwb(S1_Q16, c_Q16) + shl32(S1_Q16, 1) - S0_Q16 + 1;
        S0_Q16 = imin(S0_Q16, (i32)1 << 16);
        px_win[k + 2] = (i16)s_smulwb((S0_Q16 + S1_Q16) >> 1, (i32)px[k + 2]);
        px_win[k + 3] = (i16)s_smulwb(S0_Q16, (i32)px[k + 3]);
        S1_Q16 = s_smulwb(S0_Q16, c_Q16) + shl32(S0_Q16, 1) - S1_Q16;
        S1_Q16 = imin(S1_Q16, (i32)1 << 16);
    }
}

// warped_autocorrelation_FIX.c:44-93 (QC = 10, QS = 14)
template <class XA>
CA_DEV void silk_warped_autocorrelation_dev(i32 *corr, int *scale, XA input, int warping_Q16, int length, int order)
{
    i32 state_QS[MAX_SHAPE_LPC_ORDER + 1];
    i64 corr_QC[MAX_SHAPE_LPC_ORDER + 1];
    for (int i = 0; i <= MAX_SHAPE_LPC_ORDER; i++) { state_QS[i] = 0; corr_QC[i] = 0; }
    // The section loop is unrolled to the maximum order with the trip test inside, so that state_QS[] / corr_QC[] are indexed
    // by constants only and live in registers (a run-time index would put both arrays, 204 bytes per lane, in scratch memory).
    for (int n = 0; n < length; n++) {
        i32 tmp1_QS = shl32((i32)input[n], 14);
#pragma unroll
        for (int i = 0; i < MAX_SHAPE_LPC_ORDER; i += 2) {
            if (i < order) {
                const i32 tmp2_QS = s_smlawb(state_QS[i], s_subw(state_QS[i + 1], tmp1_QS), warping_Q16);
                state_QS[i] = tmp1_QS;
                corr_QC[i] += ((i64)tmp1_QS * state_QS[0]) >> (2 * 14 - 10);
                tmp1_QS = s_smlawb(state_QS[i + 1], s_subw(state_QS[i + 2], tmp2_QS), warping_Q16);
                state_QS[i + 1] = tmp2_QS;
                corr_QC[i + 1] += ((i64)tmp2_QS * state_QS[0]) >> (2 * 14 - 10);
                if (i + 2 == order) {                                                       // after the last section (:76-77)
                    state_QS[i + 2] = tmp1_QS;
                    corr_QC[i + 2] += ((i64)tmp1_QS * state_QS[0]) >> (2 * 14 - 10);
                }
            }
        }
    }
    const i32 c_hi = (i32)(corr_QC[0] >> 32);                                               // silk_CLZ64 (macros.h)
    int lsh = (c_hi == 0 ? 32 + s_clz32((i32)corr_QC[0]) : s_clz32(c_hi)) - 35;
    lsh = s_limit(lsh, -12 - 10, 30 - 10);
    *scale = -(10 + lsh);
#pragma unroll
    for (int i = 0; i <= MAX_SHAPE_LPC_ORDER; i++) {
        const i32 v = lsh >= 0 ? (i32)(corr_QC[i] << lsh) : (i32)(corr_QC[i] >> -lsh);
        if (i <= order) corr[i] = v;
    }
}

// silk_autocorr (autocorr_FIX.c:35-48) = _celt_autocorr with no window (celt_lpc.c:232-328): ac[k] = sum_j x[j] * x[j + k] over the
// (possibly pre-shifted) input, MAC16_16 sums that wrap, then the normalisation of :300-323. xs: scratch for the shifted copy.
template <class XA, class XS>
CA_DEV int silk_autocorr_dev(i32 *ac, XA x, XS xs, int n, int correlationCount)
{
    const int lag = imin(n, correlationCount) - 1;
    i32 ac0 = 1 + (n << 7);
    if (n & 1) { const i32 v = x[0]; ac0 += __mul24(v, v) >> 9; }
#pragma unroll 4
    for (int i = n & 1; i < n; i += 2) {
        const i32 a = x[i], b = x[i + 1];
        ac0 += __mul24(a, a) >> 9;
        ac0 += __mul24(b, b) >> 9;
    }
    int shift = (31 - s_clz32(ac0)) - 30 + 10;          // celt_ilog2(ac0) - 30 + 10
    shift = shift / 2;
    if (shift > 0) {
#pragma unroll 8
        for (int i = 0; i < n; i++) xs[i] = (i16)(((i32)x[i] + ((i32)1 << (shift - 1))) >> shift);
    } else {
        shift = 0;
#pragma unroll 8
        for (int i = 0; i < n; i++) xs[i] = (i16)(i32)x[i];
    }
    {
        // all lags in one pass: the previous samples travel in a register window that starts at zero (the products a lag's loop
        // bound leaves out vanish); MAC16_16 sums wrap, so the order of the products is free
        enum { MAXLAG = MAX_SHAPE_LPC_ORDER + 1 };
        i32 acc[MAXLAG], w[MAXLAG];
#pragma unroll
        for (int k = 0; k < MAXLAG; k++) { acc[k] = 0; w[k] = 0; }
#pragma unroll 4
        for (int j = 0; j < n; j++) {
            const i32 xj = (i32)xs[j];
#pragma unroll
            for (int k = MAXLAG - 1; k > 0; k--) w[k] = w[k - 1];
            w[0] = xj;
#pragma unroll
            for (int k = 0; k < MAXLAG; k++) acc[k] = s_addw(acc[k], __mul24(xj, w[k]));
        }
        for (int k = 0; k <= lag; k++) ac[k] = acc[k];
    }
    shift = 2 * shift;
    if (shift <= 0) ac[0] = s_addw(ac[0], shl32((i32)1, -shift));
    if (ac[0] < 268435456) {
        const int shift2 = 29 - (32 - s_clz32(ac[0]));                                     // 29 - EC_ILOG(ac[0])
        for (int i = 0; i <= lag; i++) ac[i] = shl32(ac[i], shift2);
        shift -= shift2;
    } else if (ac[0] >= 536870912) {
        const int shift2 = ac[0] >= 1073741824 ? 2 : 1;
        for (int i = 0; i <= lag; i++) ac[i] >>= shift2;
        shift += shift2;
    }
    return shift;
}

CA_DEV i32 silk_schur64_dev(i32 *rc_Q16, const i32 *c, int order)                           // schur64_FIX.c:36-92
{
    i32 C0[MAX_SHAPE_LPC_ORDER + 1], C1[MAX_SHAPE_LPC_ORDER + 1];
    if (c[0] <= 0) {
        for (int k = 0; k < order; k++) rc_Q16[k] = 0;
        return 0;
    }
    for (int k = 0; k < order + 1; k++) C0[k] = C1[k] = c[k];
    int k;
    for (k = 0; k < order; k++) {
        if (s_abs(C0[k + 1]) >= C1[0]) {
            rc_Q16[k] = C0[k + 1] > 0 ? -64881 : 64881;                                     // SILK_FIX_CONST(.99f, 16)
            k++;
            break;
        }
        const i32 rc_tmp_Q31 = s_div32_varq(-C0[k + 1], C1[0], 31);
        rc_Q16[k] = s_rshift_round(rc_tmp_Q31, 15);
        for (int n = 0; n < order - k; n++) {
            const i32 Ctmp1_Q30 = C0[n + k + 1], Ctmp2_Q30 = C1[n];
            C0[n + k + 1] = s_addw(Ctmp1_Q30, s_smmul(shl32(Ctmp2_Q30, 1), rc_tmp_Q31));
            C1[n] = s_addw(Ctmp2_Q30, s_smmul(shl32(Ctmp1_Q30, 1), rc_tmp_Q31));
        }
    }
    for (; k < order; k++) rc_Q16[k] = 0;
    return imax(1, C1[0]);
}

CA_DEV void silk_k2a_Q16_dev(i32 *A_Q24, const i32 *rc_Q16, int order)                      // k2a_Q16_FIX.c:35-53
{
    i32 Atmp[MAX_SHAPE_LPC_ORDER];
    for (int k = 0; k < order; k++) {
        for (int n = 0; n < k; n++) Atmp[n] = A_Q24[n];
        for (int n = 0; n < k; n++) A_Q24[n] = s_smlaww(A_Q24[n], Atmp[k - n - 1], rc_Q16[k]);
        A_Q24[k] = (i32)(0u - (u32)shl32(rc_Q16[k], 8));
    }
}

CA_DEV i32 silk_LPC_inverse_pred_gain_Q24_dev(const i32 *A_Q24, int order)                  // LPC_inv_pred_gain.c:136-153 + :41-106, QA = 24
{
    const i32 A_LIMIT = 16773022;                  // SILK_FIX_CONST(0.99975, 24)
    i32 A[2][MAX_SHAPE_LPC_ORDER];
    i32 *Anew = A[order & 1];
    for (int k = 0; k < order; k++) Anew[k] = A_Q24[k];
    i32 invGain_Q30 = (i32)1 << 30;
    for (int k = order - 1; k > 0; k--) {
        if (Anew[k] > A_LIMIT || Anew[k] < -A_LIMIT) return 0;
        const i32 rc_Q31 = (i32)(0u - (u32)shl32(Anew[k], 31 - 24));
        const i32 rc_mult1_Q30 = ((i32)1 << 30) - s_smmul(rc_Q31, rc_Q31);
        const int mult2Q = 32 - s_clz32(s_abs(rc_mult1_Q30));
        const i32 rc_mult2 = s_inverse32_varq(rc_mult1_Q30, mult2Q + 30);
        invGain_Q30 = shl32(s_smmul(invGain_Q30, rc_mult1_Q30), 2);
        i32 *Aold = Anew;
        Anew = A[k & 1];
        for (int n = 0; n < k; n++) {
            const i32 tmp = Aold[n] - s_mul32_frac_q(Aold[k - n - 1], rc_Q31, 31);
            Anew[n] = s_mul32_frac_q(tmp, rc_mult2, mult2Q);
        }
    }
    if (Anew[0] > A_LIMIT || Anew[0] < -A_LIMIT) return 0;
    const i32 rc_Q31 = (i32)(0u - (u32)shl32(Anew[0], 31 - 24));
    const i32 rc_mult1_Q30 = ((i32)1 << 30) - s_smmul(rc_Q31, rc_Q31);
    return shl32(s_smmul(invGain_Q30, rc_mult1_Q30), 2);
}

CA_DEV i32 shape_warped_gain(const i32 *coefs_Q24, int lambda_Q16, int order)               // noise_shape_analysis_FIX.c:41-56
{
    lambda_Q16 = -lambda_Q16;
    i32 gain_Q24 = coefs_Q24[order - 1];
    for (int i = order - 2; i >= 0; i--) gain_Q24 = s_smlawb(coefs_Q24[i], gain_Q24, lambda_Q16);
    gain_Q24 = s_smlawb(16777216, gain_Q24, -lambda_Q16);
    return s_inverse32_varq(gain_Q24, 40);
}

CA_DEV void shape_to_monic(i32 *syn, i32 *ana, int lambda_Q16, int order, i32 *gain_syn_Q16, i32 *gain_ana_Q16)   // :70-85, :121-134
{
    for (int i = order - 1; i > 0; i--) {
        syn[i - 1] = s_smlawb(syn[i - 1], syn[i], -lambda_Q16);
        ana[i - 1] = s_smlawb(ana[i - 1], ana[i], -lambda_Q16);
    }
    const i32 nom_Q16 = s_smlawb(65536, -(i32)lambda_Q16, lambda_Q16);
    i32 den_Q24 = s_smlawb(16777216, syn[0], lambda_Q16);
    *gain_syn_Q16 = s_div32_varq(nom_Q16, den_Q24, 24);
    den_Q24 = s_smlawb(16777216, ana[0], lambda_Q16);
    *gain_ana_Q16 = s_div32_varq(nom_Q16, den_Q24, 24);
    for (int i = 0; i < order; i++) {
        syn[i] = s_smulww(*gain_syn_Q16, syn[i]);
        ana[i] = s_smulww(*gain_ana_Q16, ana[i]);
    }
}

CA_DEV void shape_limit_warped_coefs(i32 *syn, i32 *ana, int lambda_Q16, i32 limit_Q24, int order)   // :60-136
{
    i32 gain_syn_Q16, gain_ana_Q16;
    int ind = 0;
    shape_to_monic(syn, ana, lambda_Q16, order, &gain_syn_Q16, &gain_ana_Q16);
    for (int iter = 0; iter < 10; iter++) {
        i32 maxabs_Q24 = -1;
        for (int i = 0; i < order; i++) {
            const i32 tmp = imax(s_abs(syn[i]), s_abs(ana[i]));
            if (tmp > maxabs_Q24) { maxabs_Q24 = tmp; ind = i; }
        }
        if (maxabs_Q24 <= limit_Q24) return;
        for (int i = 1; i < order; i++) {
            syn[i - 1] = s_smlawb(syn[i - 1], syn[i], lambda_Q16);
            ana[i - 1] = s_smlawb(ana[i - 1], ana[i], lambda_Q16);
        }
        gain_syn_Q16 = s_inverse32_varq(gain_syn_Q16, 32);
        gain_ana_Q16 = s_inverse32_varq(gain_ana_Q16, 32);
        for (int i = 0; i < order; i++) {
            syn[i] = s_smulww(gain_syn_Q16, syn[i]);
            ana[i] = s_smulww(gain_ana_Q16, ana[i]);
        }
        // SILK_FIX_CONST(0.99, 16) = 64881, (0.8, 10) = 819, (0.1, 10) = 102
        const i32 chirp_Q16 = 64881 - s_div32_varq(s_smulwb(maxabs_Q24 - limit_Q24, 819 + s_smulbb(102, iter)), s_mulw(maxabs_Q24, ind + 1), 22);
        silk_bwexpander_32_dev(syn, order, chirp_Q16);
        silk_bwexpander_32_dev(ana, order, chirp_Q16);
        shape_to_monic(syn, ana, lambda_Q16, order, &gain_syn_Q16, &gain_ana_Q16);
    }
}

struct ShapeCfg {                                       // the psEnc / psEncCtrl fields the call reads
    int fs_kHz, nb_subfr, subfr_length, la_shape, shapeWinLength, shapingLPCOrder, warping_Q16, SNR_dB_Q7, useCBR, speech_activity_Q8,
        signalType, input_quality_bands_Q15[2], LTPCorr_Q15, predGain_Q16, pitchL[4];
};

struct ShapeOut {                                       // what it writes (psEncCtrl, psEnc->sCmn.indices.quantOffsetType, psEnc->sShape)
    i32 Gains_Q16[4];
    int GainsPre_Q14[4];
    i16 AR1_Q13[4 * MAX_SHAPE_LPC_ORDER], AR2_Q13[4 * MAX_SHAPE_LPC_ORDER];
    i32 LF_shp_Q14[4];
    int HarmBoost_Q14[4], HarmShapeGain_Q14[4], Tilt_Q14[4];
    i32 HarmBoost_smth_Q16, HarmShapeGain_smth_Q16, Tilt_smth_Q16;       // I/O
    int input_quality_Q14, coding_quality_Q14, sparseness_Q8, quantOffsetType;
};

// x: index 0 = the reference's x[0] (frame start; indices from -la_shape); pitch_res: the frame's LPC residual; xw / xs: two
// scratch arrays of shapeWinLength samples in the caller's fast storage.
template <class XG, class SCR>
CA_DEV void silk_noise_shape_analysis_order_dev(const ShapeCfg &c, XG pitch_res, XG x, SCR xw, SCR xs, ShapeOut &o, const int order)
{
    i32 SNR_adj_dB_Q7 = c.SNR_dB_Q7;
    o.input_quality_Q14 = ((i32)c.input_quality_bands_Q15[0] + c.input_quality_bands_Q15[1]) >> 2;
    o.coding_quality_Q14 = silk_sigm_Q15_dev(s_rshift_round(SNR_adj_dB_Q7 - 2560, 4)) >> 1;
    if (c.useCBR == 0) {
        i32 b_Q8 = 256 - c.speech_activity_Q8;
        b_Q8 = s_smulwb(shl32(b_Q8, 8), b_Q8);
        SNR_adj_dB_Q7 = s_smlawb(SNR_adj_dB_Q7, s_smulbb(-8, b_Q8), s_smulwb(16384 + o.input_quality_Q14, o.coding_quality_Q14));
    }
    if (c.signalType == 2) {
        SNR_adj_dB_Q7 = s_smlawb(SNR_adj_dB_Q7, 512, c.LTPCorr_Q15);
    } else {
        SNR_adj_dB_Q7 = s_smlawb(SNR_adj_dB_Q7, s_smlawb(3072, -104858, c.SNR_dB_Q7), 16384 - o.input_quality_Q14);
    }
    // sparseness (:204-239)
    if (c.signalType == 2) {
        o.quantOffsetType = 0;
        o.sparseness_Q8 = 0;
    } else {
        const int nSamples = c.fs_kHz << 1;
        i32 energy_variation_Q7 = 0, log_energy_prev_Q7 = 0;
        const int nblk = s_smulbb(5, c.nb_subfr) / 2;
        for (int k = 0; k < nblk; k++) {
            i32 nrg;
            int scale;
            silk_sum_sqr_shift_dev(&nrg, &scale, pitch_res + k * nSamples, nSamples);
            nrg += nSamples >> scale;
            const i32 log_energy_Q7 = s_lin2log(nrg);
            if (k > 0) energy_variation_Q7 += s_abs(log_energy_Q7 - log_energy_prev_Q7);
            log_energy_prev_Q7 = log_energy_Q7;
        }
        o.sparseness_Q8 = silk_sigm_Q15_dev(s_smulwb(energy_variation_Q7 - 640, 6554)) >> 7;
        o.quantOffsetType = o.sparseness_Q8 > 192 ? 0 : 1;
        SNR_adj_dB_Q7 = s_smlawb(SNR_adj_dB_Q7, 65536, o.sparseness_Q8 - 128);
    }
    // bandwidth expansion (:241-262)
    i32 strength_Q16 = s_smulwb(c.predGain_Q16, 66);
    i32 BWExp1_Q16, BWExp2_Q16;
    BWExp1_Q16 = BWExp2_Q16 = s_div32_varq(62259, s_smlaww(65536, strength_Q16, strength_Q16), 16);
    const i32 delta_Q16 = s_smulwb(65536 - s_smulbb(3, o.coding_quality_Q14), 655);
    BWExp1_Q16 = s_subw(BWExp1_Q16, delta_Q16);
    BWExp2_Q16 = s_addw(BWExp2_Q16, delta_Q16);
    BWExp1_Q16 = shl32(BWExp1_Q16, 14) / (BWExp2_Q16 >> 2);
    const int warping_Q16 = c.warping_Q16 > 0 ? s_smlawb(c.warping_Q16, (i32)o.coding_quality_Q14, 2621) : 0;
    // AR coefficients and gains per subframe (:264-353)
    const int flat_part = c.fs_kHz * 3, slope_part = (c.shapeWinLength - flat_part) >> 1;
    for (int k = 0; k < c.nb_subfr; k++) {
        const XG x_ptr = x + (k * c.subfr_length - c.la_shape);
        silk_apply_sine_window_dev(xw, x_ptr, 1, slope_part);
        for (int i = 0; i < flat_part; i++) xw[slope_part + i] = (i16)(i32)x_ptr[slope_part + i];
        silk_apply_sine_window_dev(xw + (slope_part + flat_part), x_ptr + (slope_part + flat_part), 2, slope_part);
        i32 auto_corr[MAX_SHAPE_LPC_ORDER + 1], refl_coef_Q16[MAX_SHAPE_LPC_ORDER], AR1_Q24[MAX_SHAPE_LPC_ORDER], AR2_Q24[MAX_SHAPE_LPC_ORDER];
        int scale;
        if (c.warping_Q16 > 0) silk_warped_autocorrelation_dev(auto_corr, &scale, xw, warping_Q16, c.shapeWinLength, order);
        else scale = silk_autocorr_dev(auto_corr, xw, xs, c.shapeWinLength, order + 1);
        auto_corr[0] = s_addw(auto_corr[0], imax(s_smulwb(auto_corr[0] >> 4, 52), 1));
        i32 nrg = silk_schur64_dev(refl_coef_Q16, auto_corr, order);
        silk_k2a_Q16_dev(AR2_Q24, refl_coef_Q16, order);
        int Qnrg = -scale;
        if (Qnrg & 1) { Qnrg -= 1; nrg >>= 1; }
        const i32 tmp32 = s_sqrt_approx(nrg);
        Qnrg >>= 1;
        o.Gains_Q16[k] = s_lshift_sat32(tmp32, 16 - Qnrg);
        if (c.warping_Q16 > 0) {
            const i32 gain_mult_Q16 = shape_warped_gain(AR2_Q24, warping_Q16, order);
            if ((((i64)s_rshift_round(o.Gains_Q16[k], 1) * gain_mult_Q16) >> 16) >= (0x7FFFFFFF >> 1)) o.Gains_Q16[k] = 0x7FFFFFFF;
            else o.Gains_Q16[k] = s_smulww(o.Gains_Q16[k], gain_mult_Q16);
        }
        silk_bwexpander_32_dev(AR2_Q24, order, BWExp2_Q16);
        for (int i = 0; i < order; i++) AR1_Q24[i] = AR2_Q24[i];
        silk_bwexpander_32_dev(AR1_Q24, order, BWExp1_Q16);
        i32 pre_nrg_Q30 = silk_LPC_inverse_pred_gain_Q24_dev(AR2_Q24, order);
        nrg = silk_LPC_inverse_pred_gain_Q24_dev(AR1_Q24, order);
        pre_nrg_Q30 = shl32(s_smulwb(pre_nrg_Q30, 22938), 1);
        o.GainsPre_Q14[k] = 4915 + s_div32_varq(pre_nrg_Q30, nrg, 14);
        shape_limit_warped_coefs(AR2_Q24, AR1_Q24, warping_Q16, 67092087, order);
        for (int i = 0; i < order; i++) {
            const i32 a1 = s_rshift_round(AR1_Q24[i], 11), a2 = s_rshift_round(AR2_Q24[i], 11);
            o.AR1_Q13[k * MAX_SHAPE_LPC_ORDER + i] = (i16)(a1 > 32767 ? 32767 : (a1 < -32768 ? -32768 : a1));
            o.AR2_Q13[k * MAX_SHAPE_LPC_ORDER + i] = (i16)(a2 > 32767 ? 32767 : (a2 < -32768 ? -32768 : a2));
        }
    }
    // gain tweaking (:355-373)
    i32 gain_mult_Q16 = s_log2lin(-s_smlawb(-2048, SNR_adj_dB_Q7, 10486));
    const i32 gain_add_Q16 = s_log2lin(s_smlawb(2048, 256, 10486));
    for (int k = 0; k < c.nb_subfr; k++) {
        o.Gains_Q16[k] = s_smulww(o.Gains_Q16[k], gain_mult_Q16);
        const i32 s = s_addw(o.Gains_Q16[k], gain_add_Q16);                                 // silk_ADD_POS_SAT32
        o.Gains_Q16[k] = (s & 0x80000000) ? 0x7FFFFFFF : s;
    }
    gain_mult_Q16 = 65536 + s_rshift_round(s_addw(3355443, s_mulw(o.coding_quality_Q14, 410)), 10);
    for (int k = 0; k < c.nb_subfr; k++) o.GainsPre_Q14[k] = s_smulwb(gain_mult_Q16, o.GainsPre_Q14[k]);
    // low-frequency shaping and tilt (:375-409)
    strength_Q16 = s_mulw(64, s_smlawb(4096, 4096, c.input_quality_bands_Q15[0] - 32768));
    strength_Q16 = s_mulw(strength_Q16, c.speech_activity_Q8) >> 8;
    i32 Tilt_Q16;
    if (c.signalType == 2) {
        const int fs_kHz_inv = 3277 / c.fs_kHz;
        for (int k = 0; k < c.nb_subfr; k++) {
            const int b_Q14 = fs_kHz_inv + 49152 / c.pitchL[k];
            o.LF_shp_Q14[k] = shl32(16384 - b_Q14 - s_smulwb(strength_Q16, b_Q14), 16);
            o.LF_shp_Q14[k] |= (i32)(u16)(b_Q14 - 16384);
        }
        Tilt_Q16 = -16384 - s_smulwb(65536 - 16384, s_smulwb(5872026, c.speech_activity_Q8));
    } else {
        const int b_Q14 = 21299 / c.fs_kHz;
        o.LF_shp_Q14[0] = shl32(16384 - b_Q14 - s_smulwb(strength_Q16, s_smulwb(39322, b_Q14)), 16);
        o.LF_shp_Q14[0] |= (i32)(u16)(b_Q14 - 16384);
        for (int k = 1; k < c.nb_subfr; k++) o.LF_shp_Q14[k] = o.LF_shp_Q14[0];
        Tilt_Q16 = -16384;
    }
    // harmonic shaping control (:411-436)
    i32 HarmBoost_Q16 = s_smulwb(s_smulwb(131072 - shl32(o.coding_quality_Q14, 3), c.LTPCorr_Q15), 6554);
    HarmBoost_Q16 = s_smlawb(HarmBoost_Q16, 65536 - shl32(o.input_quality_Q14, 2), 6554);
    i32 HarmShapeGain_Q16 = 0;
    if (c.signalType == 2) {
        HarmShapeGain_Q16 = s_smlawb(19661, 65536 - s_smulwb(262144 - shl32(o.coding_quality_Q14, 4), o.input_quality_Q14), 13107);
        HarmShapeGain_Q16 = s_smulwb(shl32(HarmShapeGain_Q16, 1), s_sqrt_approx(shl32(c.LTPCorr_Q15, 15)));
    }
    // smoothing over subframes (:438-452), always MAX_NB_SUBFR steps; SILK_FIX_CONST(SUBFR_SMTH_COEF, 16) = 26214
    for (int k = 0; k < 4; k++) {
        o.HarmBoost_smth_Q16 = s_smlawb(o.HarmBoost_smth_Q16, HarmBoost_Q16 - o.HarmBoost_smth_Q16, 26214);
        o.HarmShapeGain_smth_Q16 = s_smlawb(o.HarmShapeGain_smth_Q16, HarmShapeGain_Q16 - o.HarmShapeGain_smth_Q16, 26214);
        o.Tilt_smth_Q16 = s_smlawb(o.Tilt_smth_Q16, Tilt_Q16 - o.Tilt_smth_Q16, 26214);
        o.HarmBoost_Q14[k] = s_rshift_round(o.HarmBoost_smth_Q16, 2);
        o.HarmShapeGain_Q14[k] = s_rshift_round(o.HarmShapeGain_smth_Q16, 2);
        o.Tilt_Q14[k] = s_rshift_round(o.Tilt_smth_Q16, 2);
    }
}

// The shaping orders the encoder's complexity settings use (control_codec.c:333-404: 16, 14, 12; 10 / 8 at the lowest settings go
// through the generic instance) as separate instances of the inlined body: with the order a constant the coefficient loops of the
// autocorrelation, Schur, k2a, bandwidth expansion and coefficient limiting unroll and their arrays are registers.
template <class XG, class SCR>
CA_DEV void silk_noise_shape_analysis_dev(const ShapeCfg &c, XG pitch_res, XG x, SCR xw, SCR xs, ShapeOut &o)
{
    if (c.shapingLPCOrder == 16) silk_noise_shape_analysis_order_dev(c, pitch_res, x, xw, xs, o, 16);
    else if (c.shapingLPCOrder == 14) silk_noise_shape_analysis_order_dev(c, pitch_res, x, xw, xs, o, 14);
    else if (c.shapingLPCOrder == 12) silk_noise_shape_analysis_order_dev(c, pitch_res, x, xw, xs, o, 12);
    else silk_noise_shape_analysis_order_dev(c, pitch_res, x, xw, xs, o, c.shapingLPCOrder);
}

}  // namespace ca
