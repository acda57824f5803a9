// silk_lpc_dev.h -- silk_find_LPC_FIX and what it calls, one lane per frame (first slice: the SILK
// analysis step that FEEDS silk_burg_modified on the hot path and turns its output into the NLSFs the encoder quantises).
//
//   silk_find_LPC_FIX            opus-fix/silk/fixed/find_LPC_FIX.c:37-151
//   silk_A2NLSF                  opus-fix/silk/A2NLSF.c:40-267   (trans_poly :45-59, eval_poly :62-95, init :97-127)
//   silk_NLSF2A                  opus-fix/silk/NLSF2A.c:43-178   (find_poly :43-63)
//   silk_LPC_inverse_pred_gain   opus-fix/silk/LPC_inv_pred_gain.c:34-135
//   silk_bwexpander_32           opus-fix/silk/bwexpander_32.c:35-50
//   silk_interpolate             opus-fix/silk/interpolate.c:35-51
//   silk_LPC_analysis_filter     opus-fix/silk/LPC_analysis_filter.c:47-108 (FIXED_POINT branch -> celt_fir, celt/celt_lpc.c:95-149)
//   silk_sum_sqr_shift           opus-fix/silk/sum_sqr_shift.c:36-86
//   silk_LSFCosTab_FIX_Q12       opus-fix/silk/table_LSF_cos.c:36-70
//
// Everything here is a serial recurrence per frame (root bracketing over the cosine table, polynomial deflation,
// the step-down recursion), a few hundred operations on sixteen coefficients: one lane owns one frame, the sixteen-
// element arrays are registers / private memory, the frame's samples are read through an accessor (the [sample][lane]
// LDS block of the Burg kernel on the GPU, a plain pointer in the host build). The residual of the NLSF interpolation
// search is never stored: silk_LPC_analysis_filter's output samples are generated in order and fed straight into
// silk_sum_sqr_shift's running sum. Compiles for the device and, with CA_HOST_EMU, for the CPU tests (tests/emu).
#pragma once
#include "silk_burg_dev.h"

namespace ca {

enum { LSF_COS_TAB_SZ = 128, A2NLSF_BIN_DIV_STEPS = 3, A2NLSF_MAX_ITER = 30, SILK_MAX_LPC = 16 };

CA_DEVICE_CONST i16 SILK_LSFCosTab_Q12[LSF_COS_TAB_SZ + 1] = {
    8192, 8190, 8182, 8170, 8152, 8130, 8104, 8072, 8034, 7994, 7946, 7896, 7840, 7778, 7714, 7644,
    7568, 7490, 7406, 7318, 7226, 7128, 7026, 6922, 6812, 6698, 6580, 6458, 6332, 6204, 6070, 5934,
    5792, 5648, 5502, 5352, 5198, 5040, 4880, 4718, 4552, 4382, 4212, 4038, 3862, 3684, 3502, 3320,
    3136, 2948, 2760, 2570, 2378, 2186, 1990, 1794, 1598, 1400, 1202, 1002, 802, 602, 402, 202,
    0, -202, -402, -602, -802, -1002, -1202, -1400, -1598, -1794, -1990, -2186, -2378, -2570, -2760, -2948,
    -3136, -3320, -3502, -3684, -3862, -4038, -4212, -4382, -4552, -4718, -4880, -5040, -5198, -5352, -5502, -5648,
    -5792, -5934, -6070, -6204, -6332, -6458, -6580, -6698, -6812, -6922, -7026, -7128, -7226, -7318, -7406, -7490,
    -7568, -7644, -7714, -7778, -7840, -7896, -7946, -7994, -8034, -8072, -8104, -8130, -8152, -8170, -8182, -8190,
    -8192};

// silk_SMULL + silk_RSHIFT_ROUND64 (macros.h / SigProc_FIX.h:545): ((a*b >> (Q-1)) + 1) >> 1, low 32 bits
CA_DEV i32 s_mul32_frac_q(i32 a, i32 b, int Q)
{
    const i64 p = (i64)a * (i64)b;
    return (i32)(Q == 1 ? (p >> 1) + (p & 1) : ((p >> (Q - 1)) + 1) >> 1);
}

CA_DEV void silk_bwexpander_32_dev(i32 *ar, int d, i32 chirp_Q16)                 // bwexpander_32.c:35-50
{
    const i32 chirp_minus_one_Q16 = chirp_Q16 - 65536;
    for (int i = 0; i < d - 1; i++) {
        ar[i] = s_smulww(chirp_Q16, ar[i]);
        chirp_Q16 += s_rshift_round((i32)((u32)chirp_Q16 * (u32)chirp_minus_one_Q16), 16);
    }
    ar[d - 1] = s_smulww(chirp_Q16, ar[d - 1]);
}

// ---- silk_A2NLSF --------------------------------------------------------------------------------------------------
CA_DEV void a2nlsf_trans_poly(i32 *p, int dd)                                      // A2NLSF.c:45-59
{
    for (int k = 2; k <= dd; k++) {
        for (int n = dd; n > k; n--) p[n - 2] -= p[n];
        p[k - 2] -= shl32(p[k], 1);
    }
}

CA_DEV i32 a2nlsf_eval_poly(const i32 *p, i32 x, int dd)                           // A2NLSF.c:62-95
{
    i32 y32 = p[dd];
    const i32 x_Q16 = shl32(x, 4);
    for (int n = dd - 1; n >= 0; n--) y32 = s_smlaww(p[n], y32, x_Q16);
    return y32;
}

CA_DEV void a2nlsf_init(const i32 *a_Q16, i32 *P, i32 *Q, int dd)                  // A2NLSF.c:97-127
{
    P[dd] = 1 << 16;
    Q[dd] = 1 << 16;
    for (int k = 0; k < dd; k++) {
        P[k] = (i32)(0u - (u32)a_Q16[dd - k - 1] - (u32)a_Q16[dd + k]);
        Q[k] = (i32)(0u - (u32)a_Q16[dd - k - 1] + (u32)a_Q16[dd + k]);
    }
    for (int k = dd; k > 0; k--) {
        P[k - 1] -= P[k];
        Q[k - 1] += Q[k];
    }
    a2nlsf_trans_poly(P, dd);
    a2nlsf_trans_poly(Q, dd);
}

// NLSF[d] out; a_Q16[d] in/out (bandwidth-expanded when the root search fails). d even, <= 16.
CA_DEV void silk_A2NLSF_dev(i16 *NLSF, i32 *a_Q16, int d)                          // A2NLSF.c:131-267
{
    i32 PQ[2][SILK_MAX_LPC / 2 + 1];
    const int dd = d >> 1;
    a2nlsf_init(a_Q16, PQ[0], PQ[1], dd);
    int sel = 0;                                   // which polynomial: 0 = P, 1 = Q
    i32 xlo = SILK_LSFCosTab_Q12[0];
    i32 ylo = a2nlsf_eval_poly(PQ[0], xlo, dd);
    int root_ix;
    if (ylo < 0) {
        NLSF[0] = 0;
        sel = 1;
        ylo = a2nlsf_eval_poly(PQ[1], xlo, dd);
        root_ix = 1;
    } else {
        root_ix = 0;
    }
    int k = 1, i = 0;
    i32 thr = 0;
    for (;;) {
        i32 xhi = SILK_LSFCosTab_Q12[k];
        i32 yhi = a2nlsf_eval_poly(PQ[sel], xhi, dd);
        if ((ylo <= 0 && yhi >= thr) || (ylo >= 0 && yhi <= -thr)) {
            thr = yhi == 0 ? 1 : 0;
            i32 ffrac = -256;
            for (int m = 0; m < A2NLSF_BIN_DIV_STEPS; m++) {
                const i32 xmid = s_rshift_round(xlo + xhi, 1);
                const i32 ymid = a2nlsf_eval_poly(PQ[sel], xmid, dd);
                if ((ylo <= 0 && ymid >= 0) || (ylo >= 0 && ymid <= 0)) {
                    xhi = xmid;
                    yhi = ymid;
                } else {
                    xlo = xmid;
                    ylo = ymid;
                    ffrac = ffrac + (128 >> m);
                }
            }
            if (s_abs(ylo) < 65536) {
                const i32 den = ylo - yhi;
                const i32 nom = shl32(ylo, 8 - A2NLSF_BIN_DIV_STEPS) + (den >> 1);
                if (den != 0) ffrac += nom / den;
            } else {
                ffrac += ylo / ((ylo - yhi) >> (8 - A2NLSF_BIN_DIV_STEPS));
            }
            const i32 v = shl32((i32)k, 8) + ffrac;
            NLSF[root_ix] = (i16)(v < 32767 ? v : 32767);
            root_ix++;
            if (root_ix >= d) break;
            sel = root_ix & 1;
            xlo = SILK_LSFCosTab_Q12[k - 1];
            ylo = shl32(1 - (root_ix & 2), 12);
        } else {
            k++;
            xlo = xhi;
            ylo = yhi;
            thr = 0;
            if (k > LSF_COS_TAB_SZ) {
                i++;
                if (i > A2NLSF_MAX_ITER) {
                    NLSF[0] = (i16)((1 << 15) / (d + 1));
                    for (k = 1; k < d; k++) NLSF[k] = (i16)s_smulbb(k + 1, NLSF[0]);
                    return;
                }
                silk_bwexpander_32_dev(a_Q16, d, 65536 - s_smulbb(10 + i, i));
                a2nlsf_init(a_Q16, PQ[0], PQ[1], dd);
                sel = 0;
                xlo = SILK_LSFCosTab_Q12[0];
                ylo = a2nlsf_eval_poly(PQ[0], xlo, dd);
                if (ylo < 0) {
                    NLSF[0] = 0;
                    sel = 1;
                    ylo = a2nlsf_eval_poly(PQ[1], xlo, dd);
                    root_ix = 1;
                } else {
                    root_ix = 0;
                }
                k = 1;
            }
        }
    }
}

// ---- silk_LPC_inverse_pred_gain (Q12 input) ---------------------------------------------------------------------------
CA_DEV i32 silk_LPC_inverse_pred_gain_dev(const i16 *A_Q12, int order)             // LPC_inv_pred_gain.c:42-135, QA = 24
{
    enum { IQA = 24 };
    const i32 A_LIMIT = 16773022;                  // SILK_FIX_CONST(0.99975, 24)
    i32 A[2][SILK_MAX_LPC];
    i32 DC_resp = 0;
    i32 *Anew = A[order & 1];
    for (int k = 0; k < order; k++) {
        DC_resp += (i32)A_Q12[k];
        Anew[k] = shl32((i32)A_Q12[k], IQA - 12);
    }
    if (DC_resp >= 4096) return 0;
    i32 invGain_Q30 = (i32)1 << 30;
    for (int k = order - 1; k > 0; k--) {
        if (Anew[k] > A_LIMIT || Anew[k] < -A_LIMIT) return 0;
        const i32 rc_Q31 = (i32)(0u - (u32)shl32(Anew[k], 31 - IQA));
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
    const i32 rc_Q31 = (i32)(0u - (u32)shl32(Anew[0], 31 - IQA));
    const i32 rc_mult1_Q30 = ((i32)1 << 30) - s_smmul(rc_Q31, rc_Q31);
    return shl32(s_smmul(invGain_Q30, rc_mult1_Q30), 2);
}

// ---- silk_NLSF2A ----------------------------------------------------------------------------------------------------------
CA_DEV void nlsf2a_find_poly(i32 *out, const i32 *cLSF, int dd)                    // NLSF2A.c:43-63, QA = 16
{
    out[0] = 1 << 16;
    out[1] = (i32)(0u - (u32)cLSF[0]);
    for (int k = 1; k < dd; k++) {
        const i32 ftmp = cLSF[2 * k];
        out[k + 1] = shl32(out[k - 1], 1) - s_mul32_frac_q(ftmp, out[k], 16);
        for (int n = k; n > 1; n--) out[n] += out[n - 2] - s_mul32_frac_q(ftmp, out[n - 1], 16);
        out[1] -= ftmp;
    }
}

CA_DEV void silk_NLSF2A_dev(i16 *a_Q12, const i16 *NLSF, int d)                    // NLSF2A.c:66-178
{
    const unsigned long long ord16 = 0x1E965AD23CB478F0ull;     // ordering16[] = 0,15,8,7,4,11,12,3,2,13,10,5,6,9,14,1 (4 bits each)
    const unsigned long long ord10 = 0x7218543690ull;           // ordering10[] = 0,9,6,3,4,5,8,1,2,7
    const unsigned long long ord = d == 16 ? ord16 : ord10;
    i32 cos_LSF_QA[SILK_MAX_LPC];
    for (int k = 0; k < d; k++) {
        const i32 f_int = NLSF[k] >> (15 - 7);
        const i32 f_frac = NLSF[k] - shl32(f_int, 15 - 7);
        const i32 cos_val = SILK_LSFCosTab_Q12[f_int];
        const i32 delta = SILK_LSFCosTab_Q12[f_int + 1] - cos_val;
        cos_LSF_QA[(int)((ord >> (4 * k)) & 15)] = s_rshift_round(shl32(cos_val, 8) + delta * f_frac, 20 - 16);
    }
    const int dd = d >> 1;
    i32 P[SILK_MAX_LPC / 2 + 1], Q[SILK_MAX_LPC / 2 + 1], a32[SILK_MAX_LPC];
    nlsf2a_find_poly(P, &cos_LSF_QA[0], dd);
    nlsf2a_find_poly(Q, &cos_LSF_QA[1], dd);
    for (int k = 0; k < dd; k++) {
        const i32 Ptmp = P[k + 1] + P[k], Qtmp = Q[k + 1] - Q[k];
        a32[k] = -Qtmp - Ptmp;
        a32[d - k - 1] = Qtmp - Ptmp;
    }
    int i;
    for (i = 0; i < 10; i++) {
        i32 maxabs = 0;
        int idx = 0;
        for (int k = 0; k < d; k++) {
            const i32 absval = s_abs(a32[k]);
            if (absval > maxabs) { maxabs = absval; idx = k; }
        }
        maxabs = s_rshift_round(maxabs, 16 + 1 - 12);
        if (maxabs > 32767) {
            maxabs = maxabs < 163838 ? maxabs : 163838;
            const i32 sc_Q16 = 65470 - shl32(maxabs - 32767, 14) / (((i32)((u32)maxabs * (u32)(idx + 1))) >> 2);   // SILK_FIX_CONST(0.999, 16)
            silk_bwexpander_32_dev(a32, d, sc_Q16);
        } else {
            break;
        }
    }
    if (i == 10) {
        for (int k = 0; k < d; k++) {
            const i32 v = s_rshift_round(a32[k], 16 + 1 - 12);
            a_Q12[k] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            a32[k] = shl32((i32)a_Q12[k], 16 + 1 - 12);
        }
    } else {
        for (int k = 0; k < d; k++) a_Q12[k] = (i16)s_rshift_round(a32[k], 16 + 1 - 12);
    }
    for (i = 0; i < 16; i++) {                                                      // MAX_LPC_STABILIZE_ITERATIONS
        if (silk_LPC_inverse_pred_gain_dev(a_Q12, d) < 107374) {                    // SILK_FIX_CONST(1 / MAX_PREDICTION_POWER_GAIN = 1e-4, 30)
            silk_bwexpander_32_dev(a32, d, 65536 - shl32(2, i));
            for (int k = 0; k < d; k++) a_Q12[k] = (i16)s_rshift_round(a32[k], 16 + 1 - 12);
        } else {
            break;
        }
    }
}

// ---- residual energy of one subframe under a_Q12: silk_LPC_analysis_filter + silk_sum_sqr_shift, fused -------------------
// res(ix) = SAT16(x[ix] + PSHR32(sum_m (-B[m]) * x[ix-1-m], 12)) for ix in [first, first + len) (celt_fir, celt_lpc.c:131-146),
// fed in order into the running, conditionally down-shifted sum of squares of sum_sqr_shift.c:46-85.
template <class XA>
CA_DEV void lpc_residual_energy(XA x, const i16 *B, int d, int first, int len, i32 *energy, int *shift)
{
    // the calls below ask for res(first), res(first + 1), ... in order: the d previous samples travel in a register window, so
    // every sample of x is read once (not d + 1 times)
    i32 nB[SILK_MAX_LPC], w[SILK_MAX_LPC];
    for (int j = 0; j < SILK_MAX_LPC; j++) { nB[j] = j < d ? (i32)(i16)(-(i32)B[j]) : 0; w[j] = j < d ? (i32)x[first - 1 - j] : 0; }
    auto res = [&](int ix) -> i32 {
        i32 sum = 0;
#pragma unroll
        for (int m = 0; m < SILK_MAX_LPC; m++) sum = (i32)((u32)sum + (u32)__mul24(nB[m], w[m]));
        const i32 xi = (i32)x[ix];
        const i32 v = xi + pshr32(sum, 12);
#pragma unroll
        for (int m = SILK_MAX_LPC - 1; m > 0; m--) w[m] = w[m - 1];
        w[0] = xi;
        return (i32)(i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    };
    i32 nrg = 0;
    int shft = 0, i;
    const int n1 = len - 1;
    // one loop for sum_sqr_shift.c's two (before / after the first overflow): with shft = 0 the second form is the first
#pragma unroll 4
    for (i = 0; i < n1; i += 2) {
        const i32 a = res(first + i), b = res(first + i + 1);
        i32 t = __mul24(a, a);
        t = (i32)((u32)t + (u32)__mul24(b, b));
        nrg = (i32)((u32)nrg + ((u32)t >> shft));
        if (nrg < 0) {
            nrg = (i32)((u32)nrg >> 2);
            shft += 2;
        }
    }
    if (i == n1) {
        const i32 a = res(first + i);
        nrg = (i32)((u32)nrg + ((u32)__mul24(a, a) >> shft));
    }
    if (nrg & 0xC0000000) {
        nrg = (i32)((u32)nrg >> 2);
        shft += 2;
    }
    *shift = shft;
    *energy = nrg;
}

// ---- silk_find_LPC_FIX ----------------------------------------------------------------------------------------------------
// x: (subfr_length + order) * nb_subfr samples (LPC_in_pre of find_pred_coefs_FIX.c:136); subfr_length as in psEncC (without
// the order samples). Writes NLSF_Q15[order] and returns psEncC->indices.NLSFInterpCoef_Q2.
// e: the Burg recursion's edge accessor over the same signal (silk_burg_dev.h), already staged.
template <class XA, class XE>
CA_DEV int silk_find_LPC_order_dev(XA x, XE e, i32 minInvGain_Q30, int subfr_length_enc, int nb_subfr, const int order, int useInterpolatedNLSFs,
                                   int first_frame_after_reset, const i16 *prev_NLSFq_Q15, i16 *NLSF_Q15)
{
    const int subfr_length = subfr_length_enc + order;
    int interp = 4;
    i32 a_Q16[SILK_MAX_LPC], res_nrg;
    int res_nrg_Q;
    silk_burg_modified_dev(x, e, minInvGain_Q30, subfr_length, nb_subfr, order, a_Q16, &res_nrg, &res_nrg_Q);
    if (useInterpolatedNLSFs && !first_frame_after_reset && nb_subfr == 4) {
        i32 a_tmp_Q16[SILK_MAX_LPC], res_tmp_nrg;
        int res_tmp_nrg_Q;
        silk_burg_modified_dev(x + 2 * subfr_length, e.from(2), minInvGain_Q30, subfr_length, 2, order, a_tmp_Q16, &res_tmp_nrg, &res_tmp_nrg_Q);
        int shift = res_tmp_nrg_Q - res_nrg_Q;
        if (shift >= 0) {
            if (shift < 32) res_nrg = res_nrg - (res_tmp_nrg >> shift);
        } else {
            res_nrg = (res_nrg >> -shift) - res_tmp_nrg;
            res_nrg_Q = res_tmp_nrg_Q;
        }
        silk_A2NLSF_dev(NLSF_Q15, a_tmp_Q16, order);
        for (int k = 3; k >= 0; k--) {
            i16 NLSF0_Q15[SILK_MAX_LPC], a_tmp_Q12[SILK_MAX_LPC];
            for (int i = 0; i < order; i++)                                          // silk_interpolate
                NLSF0_Q15[i] = (i16)((i32)prev_NLSFq_Q15[i] + (s_smulbb((i32)NLSF_Q15[i] - (i32)prev_NLSFq_Q15[i], k) >> 2));
            silk_NLSF2A_dev(a_tmp_Q12, NLSF0_Q15, order);
            i32 res_nrg0, res_nrg1;
            int rshift0, rshift1;
            lpc_residual_energy(x, a_tmp_Q12, order, order, subfr_length - order, &res_nrg0, &rshift0);
            lpc_residual_energy(x, a_tmp_Q12, order, order + subfr_length, subfr_length - order, &res_nrg1, &rshift1);
            int res_nrg_interp_Q;
            shift = rshift0 - rshift1;
            if (shift >= 0) {
                res_nrg1 = res_nrg1 >> shift;
                res_nrg_interp_Q = -rshift0;
            } else {
                res_nrg0 = res_nrg0 >> -shift;
                res_nrg_interp_Q = -rshift1;
            }
            const i32 res_nrg_interp = (i32)((u32)res_nrg0 + (u32)res_nrg1);
            shift = res_nrg_interp_Q - res_nrg_Q;
            bool lower;
            if (shift >= 0) lower = (res_nrg_interp >> shift) < res_nrg;
            else lower = -shift < 32 ? res_nrg_interp < (res_nrg >> -shift) : false;
            if (lower) {
                res_nrg = res_nrg_interp;
                res_nrg_Q = res_nrg_interp_Q;
                interp = k;
            }
        }
    }
    if (interp == 4) silk_A2NLSF_dev(NLSF_Q15, a_Q16, order);
    return interp;
}

// The two LPC orders of the format as two instances of the inlined body (order a constant: the coefficient loops of Burg, A2NLSF,
// NLSF2A and the stabilisers unroll and their small arrays are registers).
template <class XA, class XE>
CA_DEV int silk_find_LPC_dev(XA x, XE e, i32 minInvGain_Q30, int subfr_length_enc, int nb_subfr, int order, int useInterpolatedNLSFs,
                             int first_frame_after_reset, const i16 *prev_NLSFq_Q15, i16 *NLSF_Q15)
{
    if (order == 16)
        return silk_find_LPC_order_dev(x, e, minInvGain_Q30, subfr_length_enc, nb_subfr, 16, useInterpolatedNLSFs, first_frame_after_reset, prev_NLSFq_Q15, NLSF_Q15);
    return silk_find_LPC_order_dev(x, e, minInvGain_Q30, subfr_length_enc, nb_subfr, order == 10 ? 10 : order, useInterpolatedNLSFs, first_frame_after_reset,
                                   prev_NLSFq_Q15, NLSF_Q15);
}

template <class XA>
CA_DEV int silk_find_LPC_dev(XA x, i32 minInvGain_Q30, int subfr_length_enc, int nb_subfr, int order, int useInterpolatedNLSFs,
                             int first_frame_after_reset, const i16 *prev_NLSFq_Q15, i16 *NLSF_Q15)
{
    BurgEdgesOf<XA> e;
    e.x = x; e.L = subfr_length_enc + order;
    return silk_find_LPC_dev(x, e, minInvGain_Q30, subfr_length_enc, nb_subfr, order, useInterpolatedNLSFs, first_frame_after_reset, prev_NLSFq_Q15, NLSF_Q15);
}

}  // namespace ca
