// silk_ltp_dev.h -- long-term-prediction analysis of the SILK encoder, the voiced branch of silk_find_pred_coefs_FIX
// (opus-fix/silk/fixed/find_pred_coefs_FIX.c:78-103) (third slice).
//
//   silk_find_LTP_FIX / silk_fit_LTP          opus-fix/silk/fixed/find_LTP_FIX.c:43-245
//   silk_corrMatrix_FIX / silk_corrVector_FIX  opus-fix/silk/fixed/corrMatrix_FIX.c:39-158
//   silk_regularize_correlations_FIX           opus-fix/silk/fixed/regularize_correlations_FIX.c:35-47
//   silk_solve_LDL_FIX (+ its four statics)    opus-fix/silk/fixed/solve_LS_FIX.c:73-249
//   silk_residual_energy16_covar_FIX           opus-fix/silk/fixed/residual_energy16_FIX.c:35-103
//   silk_scale_vector32_Q26_lshift_18          opus-fix/silk/fixed/vector_ops_FIX.c:54-66
//   silk_quant_LTP_gains                       opus-fix/silk/quant_LTP_gains.c:35-129
//   silk_VQ_WMat_EC_c                          opus-fix/silk/VQ_WMat_EC.c:35-120
//   silk_LTP_analysis_filter_FIX               opus-fix/silk/fixed/LTP_analysis_filter_FIX.c:34-90
//   silk_LTP_scale_ctrl_FIX                    opus-fix/silk/fixed/LTP_scale_ctrl_FIX.c:35-53
//   silk_sum_sqr_shift                         opus-fix/silk/sum_sqr_shift.c:36-86
//   silk_log2lin                               opus-fix/silk/log2lin.c:37-59
//
// The reference is compiled with OPUS_FAST_INT64 on x86-64 (silk/macros.h:47): silk_SMULWW / silk_SMLAWW yield 64-bit
// intermediates that are narrowed only on assignment. Where such an intermediate is shifted right BEFORE being narrowed
// (solve_LS_FIX.c:158-159, :183) the 64-bit form is kept here; everywhere else the low 32 bits are all that survives.
#pragma once
#include "silk_nlsf_dev.h"

namespace ca {

enum { LTP_ORDER = 5, LTP_CORRS_HEAD_ROOM = 2 };

CA_DEV i32 s_add_sat32(i32 a, i32 b) { const i64 s = (i64)a + b; return s > 0x7FFFFFFFLL ? 0x7FFFFFFF : (s < -0x80000000LL ? (i32)0x80000000 : (i32)s); }
CA_DEV i32 s_sub_sat32(i32 a, i32 b) { const i64 s = (i64)a - b; return s > 0x7FFFFFFFLL ? 0x7FFFFFFF : (s < -0x80000000LL ? (i32)0x80000000 : (i32)s); }
CA_DEV i32 s_mulw(i32 a, i32 b) { return (i32)((u32)a * (u32)b); }

CA_DEV i32 s_log2lin(i32 inLog_Q7)                                                          // log2lin.c:37-59
{
    if (inLog_Q7 < 0) return 0;
    if (inLog_Q7 >= 3967) return 0x7FFFFFFF;
    i32 out = shl32(1, inLog_Q7 >> 7);
    const i32 frac_Q7 = inLog_Q7 & 0x7F;
    const i32 f = s_smlawb(frac_Q7, s_smulbb(frac_Q7, 128 - frac_Q7), -174);
    if (inLog_Q7 < 2048) out = out + (s_mulw(out, f) >> 7);
    else out = s_addw(out, s_mulw(out >> 7, f));
    return out;
}

template <class XA>
CA_DEV void silk_sum_sqr_shift_dev(i32 *energy, int *shift, XA x, int len)                  // sum_sqr_shift.c:36-86
{
    i32 nrg = 0;
    int shft = 0, i;
    len--;
    // one loop for the reference's two (before / after the first overflow): with shft = 0 the second form is the first
#pragma unroll 4
    for (i = 0; i < len; i += 2) {
        const i32 a = x[i], b = x[i + 1];
        const i32 t = s_addw(__mul24(a, a), __mul24(b, b));
        nrg = (i32)((u32)nrg + ((u32)t >> shft));
        if (nrg < 0) {
            nrg = (i32)((u32)nrg >> 2);
            shft += 2;
        }
    }
    if (i == len) {
        const i32 a = x[i];
        nrg = (i32)((u32)nrg + ((u32)__mul24(a, a) >> shft));
    }
    if (nrg & 0xC0000000) {
        nrg = (i32)((u32)nrg >> 2);
        shft += 2;
    }
    *shift = shft;
    *energy = nrg;
}

// X'*t (corrMatrix_FIX.c:39-72); x: L + ORDER - 1 samples forming the data matrix, t: L samples. One pass: the ORDER samples of x that
// meet t[i] (one per lag) travel in a register window, so x and t are each read once instead of once per lag.
template <int ORDER, class XA>
CA_DEV void silk_corrVector_dev(XA x, XA t, int L, i32 *Xt, int rshifts)
{
    i32 w[ORDER], s[ORDER];
#pragma unroll
    for (int m = 0; m < ORDER; m++) { s[m] = 0; w[m] = m < ORDER - 1 ? (i32)x[m] : 0; }
#pragma unroll 8
    for (int i = 0; i < L; i++) {
        w[ORDER - 1] = (i32)x[i + ORDER - 1];
        const i32 ti = (i32)t[i];
#pragma unroll
        for (int lag = 0; lag < ORDER; lag++) s[lag] = s_addw(s[lag], __mul24(w[ORDER - 1 - lag], ti) >> rshifts);     // ptr1 = &x[ORDER - 1] - lag
#pragma unroll
        for (int m = 0; m < ORDER - 1; m++) w[m] = w[m + 1];
    }
#pragma unroll
    for (int lag = 0; lag < ORDER; lag++) Xt[lag] = s[lag];
}

// X'*X (corrMatrix_FIX.c:75-158). The lag products in one pass over x (the same register window); the diagonals' edge corrections
// read the first and last ORDER - 1 samples, kept in registers.
template <int ORDER, class XA>
CA_DEV void silk_corrMatrix_dev(XA x, int L, int head_room, i32 *XX, int *rshifts)
{
    i32 energy;
    int rshifts_local;
    silk_sum_sqr_shift_dev(&energy, &rshifts_local, x, L + ORDER - 1);
    const int head_room_rshifts = imax(head_room - s_clz32(energy), 0);
    energy >>= head_room_rshifts;
    rshifts_local += head_room_rshifts;
    i32 hd[ORDER - 1], tl[ORDER - 1];                       // x[0 .. ORDER - 1) and x[L .. L + ORDER - 1)
#pragma unroll
    for (int m = 0; m < ORDER - 1; m++) { hd[m] = x[m]; tl[m] = x[L + m]; }
#pragma unroll
    for (int i = 0; i < ORDER - 1; i++) energy -= __mul24(hd[i], hd[i]) >> rshifts_local;
    if (rshifts_local < *rshifts) {
        energy >>= *rshifts - rshifts_local;
        rshifts_local = *rshifts;
    }
    XX[0] = energy;
    // ptr1 = &x[ORDER - 1]: x[p1 + L - j] = tl[ORDER - 1 - j], x[p1 - j] = hd[ORDER - 1 - j]
#pragma unroll
    for (int j = 1; j < ORDER; j++) {
        const i32 a = tl[ORDER - 1 - j], b = hd[ORDER - 1 - j];
        energy = s_subw(energy, __mul24(a, a) >> rshifts_local);
        energy = s_addw(energy, __mul24(b, b) >> rshifts_local);
        XX[j * ORDER + j] = energy;
    }
    i32 E[ORDER], w[ORDER];                                 // E[lag] = sum_i x[p1 + i] * x[p1 - lag + i]
#pragma unroll
    for (int m = 0; m < ORDER; m++) { E[m] = 0; w[m] = m < ORDER - 1 ? hd[m] : 0; }
#pragma unroll 8
    for (int i = 0; i < L; i++) {
        w[ORDER - 1] = (i32)x[i + ORDER - 1];
#pragma unroll
        for (int lag = 1; lag < ORDER; lag++) E[lag] = s_addw(E[lag], __mul24(w[ORDER - 1], w[ORDER - 1 - lag]) >> rshifts_local);
#pragma unroll
        for (int m = 0; m < ORDER - 1; m++) w[m] = w[m + 1];
    }
#pragma unroll
    for (int lag = 1; lag < ORDER; lag++) {
        energy = E[lag];
        XX[lag * ORDER] = energy;
        XX[lag] = energy;
        // ptr2 = &x[ORDER - 1 - lag]: x[p2 + L - j] = tl[ORDER - 1 - lag - j], x[p2 - j] = hd[ORDER - 1 - lag - j]
#pragma unroll
        for (int j = 1; j < ORDER - lag; j++) {
            const i32 a1 = tl[ORDER - 1 - j], a2 = tl[ORDER - 1 - lag - j], b1 = hd[ORDER - 1 - j], b2 = hd[ORDER - 1 - lag - j];
            energy = s_subw(energy, __mul24(a1, a2) >> rshifts_local);
            energy = s_addw(energy, __mul24(b1, b2) >> rshifts_local);
            XX[(lag + j) * ORDER + j] = energy;
            XX[j * ORDER + lag + j] = energy;
        }
    }
    *rshifts = rshifts_local;
}

// solve_LS_FIX.c:73-249 for M = LTP_ORDER: A x = b, A symmetric (A's diagonal may be raised, as in the reference)
CA_DEV void silk_solve_LDL_dev(i32 *A, const i32 *b, i32 *x_Q16)
{
    const int M = LTP_ORDER;
    i32 L_Q16[M * M], Y[M], inv_Q36[M], inv_Q48[M], v_Q0[M], D_Q0[M];
    for (int i = 0; i < M * M; i++) L_Q16[i] = 0;
    // silk_LDL_factorize_FIX (:122-190); SILK_FIX_CONST(FIND_LTP_COND_FAC = 1e-5, 31) = 21475
    int status = 1;
    const i32 diag_min_value = imax(s_smmul(s_add_sat32(A[0], A[M * M - 1]), 21475), 1 << 9);
    for (int loop_count = 0; loop_count < M && status == 1; loop_count++) {
        status = 0;
        for (int j = 0; j < M; j++) {
            const i32 *ptr1 = &L_Q16[j * M];
            i32 tmp_32 = 0;
            for (int i = 0; i < j; i++) {
                v_Q0[i] = s_smulww(D_Q0[i], ptr1[i]);
                tmp_32 = s_smlaww(tmp_32, v_Q0[i], ptr1[i]);
            }
            tmp_32 = s_subw(A[j * M + j], tmp_32);
            if (tmp_32 < diag_min_value) {
                tmp_32 = s_subw(s_smulbb(loop_count + 1, diag_min_value), tmp_32);
                for (int i = 0; i < M; i++) A[i * M + i] = s_addw(A[i * M + i], tmp_32);
                status = 1;
                break;
            }
            D_Q0[j] = tmp_32;
            const i32 one_div_diag_Q36 = s_inverse32_varq(tmp_32, 36);
            const i32 one_div_diag_Q40 = shl32(one_div_diag_Q36, 4);
            const i32 err = s_subw((i32)1 << 24, s_smulww(tmp_32, one_div_diag_Q40));
            const i32 one_div_diag_Q48 = s_smulww(err, one_div_diag_Q40);
            inv_Q36[j] = one_div_diag_Q36;
            inv_Q48[j] = one_div_diag_Q48;
            L_Q16[j * M + j] = 65536;
            const i32 *ptrA = &A[j * M];
            for (int i = j + 1; i < M; i++) {
                const i32 *ptr2 = &L_Q16[i * M];
                tmp_32 = 0;
                for (int k = 0; k < j; k++) tmp_32 = s_smlaww(tmp_32, v_Q0[k], ptr2[k]);
                tmp_32 = s_subw(ptrA[i], tmp_32);
                L_Q16[i * M + j] = (i32)((i64)s_smmul(tmp_32, one_div_diag_Q48) + (((i64)tmp_32 * one_div_diag_Q36) >> 20));
            }
        }
    }
    // silk_LS_SolveFirst_FIX (:211-228)
    for (int i = 0; i < M; i++) {
        i32 tmp_32 = 0;
        for (int j = 0; j < i; j++) tmp_32 = s_smlaww(tmp_32, L_Q16[i * M + j], Y[j]);
        Y[i] = s_subw(b[i], tmp_32);
    }
    // silk_LS_divide_Q16_FIX (:192-208)
    for (int i = 0; i < M; i++) {
        const i32 t = Y[i];
        Y[i] = (i32)((i64)s_smmul(t, inv_Q48[i]) + (((i64)t * inv_Q36[i]) >> 20));
    }
    // silk_LS_SolveLast_FIX (:231-249)
    for (int i = M - 1; i >= 0; i--) {
        i32 tmp_32 = 0;
        for (int j = M - 1; j > i; j--) tmp_32 = s_smlaww(tmp_32, L_Q16[j * M + i], x_Q16[j]);
        x_Q16[i] = s_subw(Y[i], tmp_32);
    }
}

// residual_energy16_FIX.c:35-103 for D = LTP_ORDER
CA_DEV i32 silk_residual_energy16_covar_dev(const i16 *c, const i32 *wXX, const i32 *wXx, i32 wxx, int cQ)
{
    const int D = LTP_ORDER;
    int lshifts = 16 - cQ, Qxtra = lshifts;
    i32 c_max = 0, cn[D];
    for (int i = 0; i < D; i++) c_max = imax(c_max, s_abs((i32)c[i]));
    Qxtra = imin(Qxtra, s_clz32(c_max) - 17);
    const i32 w_max = imax(wXX[0], wXX[D * D - 1]);
    Qxtra = imin(Qxtra, s_clz32(D * (s_smulwb(w_max, c_max) >> 4)) - 5);
    Qxtra = imax(Qxtra, 0);
    for (int i = 0; i < D; i++) cn[i] = shl32((i32)c[i], Qxtra);
    lshifts -= Qxtra;
    i32 tmp = 0;
    for (int i = 0; i < D; i++) tmp = s_smlawb(tmp, wXx[i], cn[i]);
    i32 nrg = s_subw(wxx >> (1 + lshifts), tmp);
    i32 tmp2 = 0;
    for (int i = 0; i < D; i++) {
        tmp = 0;
        const i32 *pRow = &wXX[i * D];
        for (int j = i + 1; j < D; j++) tmp = s_smlawb(tmp, pRow[j], cn[j]);
        tmp = s_smlawb(tmp, pRow[i] >> 1, cn[i]);
        tmp2 = s_smlawb(tmp2, tmp, cn[i]);
    }
    nrg = s_addw(nrg, shl32(tmp2, lshifts));
    if (nrg < 1) nrg = 1;
    else if (nrg > (0x7FFFFFFF >> (lshifts + 2))) nrg = 0x7FFFFFFF >> 1;
    else nrg = shl32(nrg, lshifts + 1);
    return nrg;
}

// find_LTP_FIX.c:43-231. r_lpc: the pitch-analysis residual; subframe k starts at r_lpc[mem_offset + k * subfr_length].
template <class XA>
CA_DEV void silk_find_LTP_dev(i16 *b_Q14, i32 *WLTP, int *LTPredCodGain_Q7, XA r_lpc, const int *lag, const i32 *Wght_Q15, int subfr_length,
                              int nb_subfr, int mem_offset, int *corr_rshifts)
{
    i32 b_Q16[LTP_ORDER], delta_b_Q14[LTP_ORDER], d_Q14[4], nrg[4], w[4], Rr[LTP_ORDER], rr[4];
    for (int k = 0; k < nb_subfr; k++) {
        i16 *b_Q14_ptr = b_Q14 + k * LTP_ORDER;
        i32 *WLTP_ptr = WLTP + k * LTP_ORDER * LTP_ORDER;
        const XA r_ptr = r_lpc + (mem_offset + k * subfr_length);
        const XA lag_ptr = r_lpc + (mem_offset + k * subfr_length - (lag[k] + LTP_ORDER / 2));
        int rr_shifts;
        silk_sum_sqr_shift_dev(&rr[k], &rr_shifts, r_ptr, subfr_length);
        const int LZs = s_clz32(rr[k]);
        if (LZs < LTP_CORRS_HEAD_ROOM) {
            rr[k] = s_rshift_round(rr[k], LTP_CORRS_HEAD_ROOM - LZs);
            rr_shifts += LTP_CORRS_HEAD_ROOM - LZs;
        }
        corr_rshifts[k] = rr_shifts;
        silk_corrMatrix_dev<LTP_ORDER>(lag_ptr, subfr_length, LTP_CORRS_HEAD_ROOM, WLTP_ptr, &corr_rshifts[k]);
        silk_corrVector_dev<LTP_ORDER>(lag_ptr, r_ptr, subfr_length, Rr, corr_rshifts[k]);
        if (corr_rshifts[k] > rr_shifts) rr[k] >>= corr_rshifts[k] - rr_shifts;
        // SILK_FIX_CONST(LTP_DAMPING / 3 = 0.05 / 3, 16) = 1092
        i32 regu = 1;
        regu = s_smlawb(regu, rr[k], 1092);
        regu = s_smlawb(regu, WLTP_ptr[0], 1092);
        regu = s_smlawb(regu, WLTP_ptr[LTP_ORDER * LTP_ORDER - 1], 1092);
        for (int i = 0; i < LTP_ORDER; i++) WLTP_ptr[i * LTP_ORDER + i] = s_addw(WLTP_ptr[i * LTP_ORDER + i], regu);   // regularize_correlations
        rr[k] += regu;
        silk_solve_LDL_dev(WLTP_ptr, Rr, b_Q16);
        for (int i = 0; i < LTP_ORDER; i++) {                                               // silk_fit_LTP (:233-245)
            const i32 v = s_rshift_round(b_Q16[i], 2);
            b_Q14_ptr[i] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
        }
        nrg[k] = silk_residual_energy16_covar_dev(b_Q14_ptr, WLTP_ptr, Rr, rr[k], 14);
        const int extra_shifts = imin(corr_rshifts[k], LTP_CORRS_HEAD_ROOM);
        i32 denom32 = s_lshift_sat32(s_smulwb(nrg[k], Wght_Q15[k]), 1 + extra_shifts)
                      + (s_smulwb((i32)subfr_length, 655) >> (corr_rshifts[k] - extra_shifts));
        denom32 = imax(denom32, 1);
        i32 temp32 = shl32((i32)Wght_Q15[k], 16) / denom32;
        temp32 >>= 31 + corr_rshifts[k] - extra_shifts - 26;
        i32 WLTP_max = 0;
        for (int i = 0; i < LTP_ORDER * LTP_ORDER; i++) WLTP_max = imax(WLTP_ptr[i], WLTP_max);
        const int lshift = s_clz32(WLTP_max) - 1 - 3;
        if (26 - 18 + lshift < 31) temp32 = imin(temp32, shl32((i32)1, 26 - 18 + lshift));
        for (int i = 0; i < LTP_ORDER * LTP_ORDER; i++) WLTP_ptr[i] = (i32)(((i64)WLTP_ptr[i] * temp32) >> 8);        // scale_vector32_Q26_lshift_18
        w[k] = WLTP_ptr[(LTP_ORDER / 2) * LTP_ORDER + LTP_ORDER / 2];
    }
    int maxRshifts = 0;
    for (int k = 0; k < nb_subfr; k++) maxRshifts = imax(corr_rshifts[k], maxRshifts);
    if (LTPredCodGain_Q7) {
        i32 LPC_LTP_res_nrg = 0, LPC_res_nrg = 0;
        for (int k = 0; k < nb_subfr; k++) {
            LPC_res_nrg = s_addw(LPC_res_nrg, (s_smulwb(rr[k], Wght_Q15[k]) + 1) >> (1 + (maxRshifts - corr_rshifts[k])));
            LPC_LTP_res_nrg = s_addw(LPC_LTP_res_nrg, (s_smulwb(nrg[k], Wght_Q15[k]) + 1) >> (1 + (maxRshifts - corr_rshifts[k])));
        }
        LPC_LTP_res_nrg = imax(LPC_LTP_res_nrg, 1);
        const i32 div_Q16 = s_div32_varq(LPC_res_nrg, LPC_LTP_res_nrg, 16);
        *LTPredCodGain_Q7 = s_smulbb(3, s_lin2log(div_Q16) - (16 << 7));
    }
    // smoothing (:166-230)
    for (int k = 0; k < nb_subfr; k++) {
        d_Q14[k] = 0;
        for (int i = 0; i < LTP_ORDER; i++) d_Q14[k] += b_Q14[k * LTP_ORDER + i];
    }
    i32 max_abs_d_Q14 = 0, max_w_bits = 0;
    for (int k = 0; k < nb_subfr; k++) {
        max_abs_d_Q14 = imax(max_abs_d_Q14, s_abs(d_Q14[k]));
        max_w_bits = imax(max_w_bits, 32 - s_clz32(w[k]) + corr_rshifts[k] - maxRshifts);
    }
    int extra_shifts = max_w_bits + 32 - s_clz32(max_abs_d_Q14) - 14;
    extra_shifts -= 32 - 1 - 2 + maxRshifts;
    extra_shifts = imax(extra_shifts, 0);
    const int maxRshifts_wxtra = maxRshifts + extra_shifts;
    i32 temp32 = (262 >> (maxRshifts + extra_shifts)) + 1;
    i32 wd = 0;
    for (int k = 0; k < nb_subfr; k++) {
        const i32 wk = w[k] >> (maxRshifts_wxtra - corr_rshifts[k]);
        temp32 = s_addw(temp32, wk);
        wd = s_addw(wd, shl32(s_smulww(wk, d_Q14[k]), 2));
    }
    const i32 m_Q12 = s_div32_varq(wd, temp32, 12);
    for (int k = 0; k < nb_subfr; k++) {
        i16 *b_Q14_ptr = b_Q14 + k * LTP_ORDER;
        if (2 - corr_rshifts[k] > 0) temp32 = w[k] >> (2 - corr_rshifts[k]);
        else temp32 = s_lshift_sat32(w[k], corr_rshifts[k] - 2);
        // SILK_FIX_CONST(LTP_SMOOTHING = 0.1f, 26) = 6710887 (the product is formed in single precision: 6710886.5, then + 0.5)
        const i32 g_Q26 = s_mulw(6710887 / ((6710887 >> 10) + temp32), s_lshift_sat32(s_sub_sat32(m_Q12, d_Q14[k] >> 2), 4));
        temp32 = 0;
        for (int i = 0; i < LTP_ORDER; i++) {
            delta_b_Q14[i] = imax((i32)b_Q14_ptr[i], 1638);
            temp32 += delta_b_Q14[i];
        }
        temp32 = g_Q26 / temp32;
        for (int i = 0; i < LTP_ORDER; i++)
            b_Q14_ptr[i] = (i16)s_limit((i32)b_Q14_ptr[i] + s_smulwb(s_lshift_sat32(temp32, 4), delta_b_Q14[i]), -16000, 28000);
    }
}

// VQ_WMat_EC.c:35-120
CA_DEV void silk_VQ_WMat_EC_dev(int *ind, i32 *rate_dist_Q14, int *gain_Q7, const i16 *in_Q14, const i32 *W_Q18, const i8 *cb_Q7,
                                const u8 *cb_gain_Q7, const u8 *cl_Q5, int mu_Q9, i32 max_gain_Q7, int L)
{
    *rate_dist_Q14 = 0x7FFFFFFF;
    const i8 *cb_row_Q7 = cb_Q7;
    for (int k = 0; k < L; k++, cb_row_Q7 += LTP_ORDER) {
        const int gain_tmp_Q7 = cb_gain_Q7[k];
        i32 d[5];
        for (int i = 0; i < 5; i++) d[i] = (i16)(in_Q14[i] - shl32((i32)cb_row_Q7[i], 7));
        i32 sum1_Q14 = s_smulbb(mu_Q9, cl_Q5[k]);
        sum1_Q14 = s_addw(sum1_Q14, shl32(imax(gain_tmp_Q7 - max_gain_Q7, 0), 10));
        i32 sum2_Q16;
        sum2_Q16 = s_smulwb(W_Q18[1], d[1]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[2], d[2]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[3], d[3]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[4], d[4]);
        sum2_Q16 = shl32(sum2_Q16, 1);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[0], d[0]);
        sum1_Q14 = s_smlawb(sum1_Q14, sum2_Q16, d[0]);
        sum2_Q16 = s_smulwb(W_Q18[7], d[2]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[8], d[3]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[9], d[4]);
        sum2_Q16 = shl32(sum2_Q16, 1);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[6], d[1]);
        sum1_Q14 = s_smlawb(sum1_Q14, sum2_Q16, d[1]);
        sum2_Q16 = s_smulwb(W_Q18[13], d[3]);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[14], d[4]);
        sum2_Q16 = shl32(sum2_Q16, 1);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[12], d[2]);
        sum1_Q14 = s_smlawb(sum1_Q14, sum2_Q16, d[2]);
        sum2_Q16 = s_smulwb(W_Q18[19], d[4]);
        sum2_Q16 = shl32(sum2_Q16, 1);
        sum2_Q16 = s_smlawb(sum2_Q16, W_Q18[18], d[3]);
        sum1_Q14 = s_smlawb(sum1_Q14, sum2_Q16, d[3]);
        sum2_Q16 = s_smulwb(W_Q18[24], d[4]);
        sum1_Q14 = s_smlawb(sum1_Q14, sum2_Q16, d[4]);
        if (sum1_Q14 < *rate_dist_Q14) {
            *rate_dist_Q14 = sum1_Q14;
            *ind = k;
            *gain_Q7 = gain_tmp_Q7;
        }
    }
}

// quant_LTP_gains.c:35-129
CA_DEV void silk_quant_LTP_gains_dev(i16 *B_Q14, i8 *cbk_index, int *periodicity_index, i32 *sum_log_gain_Q7, const i32 *W_Q18, int mu_Q9,
                                     int lowComplexity, int nb_subfr)
{
    i32 min_rate_dist_Q14 = 0x7FFFFFFF, best_sum_log_gain_Q7 = 0;
    int temp_idx[4] = {0, 0, 0, 0};
    for (int k = 0; k < 3; k++) {
        const i32 gain_safety = 51;                                                         // SILK_FIX_CONST(0.4, 7)
        const int off = SILK_LTP_CB_OFF[k], cbk_size = SILK_LTP_CB_SIZE[k];
        i32 rate_dist_Q14 = 0, sum_log_gain_tmp_Q7 = *sum_log_gain_Q7;
        for (int j = 0; j < nb_subfr; j++) {
            // SILK_FIX_CONST(MAX_SUM_LOG_GAIN_DB / 6.0 = 250 / 6, 7) = 5333, SILK_FIX_CONST(7, 7) = 896
            const i32 max_gain_Q7 = s_log2lin((5333 - sum_log_gain_tmp_Q7) + 896) - gain_safety;
            i32 rate_dist_Q14_subfr;
            int gain_Q7 = 0;
            silk_VQ_WMat_EC_dev(&temp_idx[j], &rate_dist_Q14_subfr, &gain_Q7, B_Q14 + j * LTP_ORDER, W_Q18 + j * LTP_ORDER * LTP_ORDER,
                                &SILK_LTP_VQ_Q7[off * LTP_ORDER], &SILK_LTP_VQ_GAIN_Q7[off], &SILK_LTP_BITS_Q5[off], mu_Q9, max_gain_Q7, cbk_size);
            const i32 s = s_addw(rate_dist_Q14, rate_dist_Q14_subfr);                       // silk_ADD_POS_SAT32
            rate_dist_Q14 = (s & 0x80000000) ? 0x7FFFFFFF : s;
            sum_log_gain_tmp_Q7 = imax(0, sum_log_gain_tmp_Q7 + s_lin2log(gain_safety + gain_Q7) - 896);
        }
        rate_dist_Q14 = imin(0x7FFFFFFF - 1, rate_dist_Q14);
        if (rate_dist_Q14 < min_rate_dist_Q14) {
            min_rate_dist_Q14 = rate_dist_Q14;
            *periodicity_index = k;
            for (int j = 0; j < nb_subfr; j++) cbk_index[j] = (i8)temp_idx[j];
            best_sum_log_gain_Q7 = sum_log_gain_tmp_Q7;
        }
        if (lowComplexity && rate_dist_Q14 < SILK_LTP_GAIN_MIDDLE_AVG_RD_Q14) break;
    }
    const i8 *cbk_ptr_Q7 = &SILK_LTP_VQ_Q7[SILK_LTP_CB_OFF[*periodicity_index] * LTP_ORDER];
    for (int j = 0; j < nb_subfr; j++)
        for (int k = 0; k < LTP_ORDER; k++) B_Q14[j * LTP_ORDER + k] = (i16)shl32((i32)cbk_ptr_Q7[cbk_index[j] * LTP_ORDER + k], 7);
    *sum_log_gain_Q7 = best_sum_log_gain_Q7;
}

// LTP_analysis_filter_FIX.c:34-90. x points at the reference's `x - predictLPCOrder` (pre_length = predictLPCOrder);
// LTP_res is written through `out` (sample index -> slot).
template <class XA, class OUT>
CA_DEV void silk_LTP_analysis_filter_dev(OUT out, XA x, const i16 *LTPCoef_Q14, const int *pitchL, const i32 *invGains_Q16, int subfr_length,
                                         int nb_subfr, int pre_length)
{
    for (int k = 0; k < nb_subfr; k++) {
        const XA x_ptr = x + k * subfr_length;
        const XA x_lag = x + (k * subfr_length - pitchL[k]);
        const i32 B0 = LTPCoef_Q14[k * LTP_ORDER], B1 = LTPCoef_Q14[k * LTP_ORDER + 1], B2 = LTPCoef_Q14[k * LTP_ORDER + 2],
                  B3 = LTPCoef_Q14[k * LTP_ORDER + 3], B4 = LTPCoef_Q14[k * LTP_ORDER + 4];
        const int n = subfr_length + pre_length;
#pragma unroll 4
        for (int i = 0; i < n; i++) {
            i32 LTP_est = __mul24((i32)x_lag[i + 2], B0);
            LTP_est = s_addw(LTP_est, __mul24((i32)x_lag[i + 1], B1));
            LTP_est = s_addw(LTP_est, __mul24((i32)x_lag[i], B2));
            LTP_est = s_addw(LTP_est, __mul24((i32)x_lag[i - 1], B3));
            LTP_est = s_addw(LTP_est, __mul24((i32)x_lag[i - 2], B4));
            LTP_est = s_rshift_round(LTP_est, 14);
            const i32 v = (i32)x_ptr[i] - LTP_est;
            const i32 r = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            out[k * n + i] = (i16)s_smulwb(invGains_Q16[k], r);
        }
    }
}

}  // namespace ca
