/* oracle/oracle_silk.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped, never timed as the product).
 *
 * Restates, on the flat function-boundary records of include/opusgpu_silk.h:
 *   silk_burg_modified_c        opus-fix/silk/fixed/burg_modified_FIX.c:45-275
 *   silk_NSQ_c                  opus-fix/silk/NSQ.c:74-180
 *   silk_noise_shape_quantizer  opus-fix/silk/NSQ.c:183-421
 *   silk_nsq_scale_states       opus-fix/silk/NSQ.c:423-496
 *   silk_LPC_analysis_filter    opus-fix/silk/LPC_analysis_filter.c:47-108 (FIXED_POINT branch -> celt_fir, celt/celt_lpc.c:93-150)
 *   silk_DIV32_varQ / silk_INVERSE32_varQ / silk_SQRT_APPROX   opus-fix/silk/Inlines.h:68-186
 *   silk_NSQ_del_dec_c (+ its quantizer and state scaling)  opus-fix/silk/NSQ_del_dec.c:112-724
 * with the 64-bit macro forms the x86-64 reference build uses (OPUS_FAST_INT64, silk/macros.h:47-102).
 * Pinned bit-exact against records captured from the compiled reference (tests/test_oracle_silk.py).
 */
#include <string.h>
#include "oracle_arith.h"
#include "../include/opusgpu_silk.h"

static inline i32 s_smulwb(i32 a, i32 b) { return (i32)(((i64)a * (i16)b) >> 16); }
static inline i32 s_smlawb(i32 a, i32 b, i32 c) { return (i32)((i64)a + (((i64)b * (i16)c) >> 16)); }
static inline i32 s_smlawt(i32 a, i32 b, i32 c) { return (i32)((i64)a + (((i64)b * ((i64)c >> 16)) >> 16)); }
static inline i32 s_smulww(i32 a, i32 b) { return (i32)(((i64)a * b) >> 16); }
static inline i32 s_smlaww(i32 a, i32 b, i32 c) { return (i32)((i64)a + (((i64)b * c) >> 16)); }
static inline i32 s_smulbb(i32 a, i32 b) { return (i32)(i16)a * (i32)(i16)b; }
static inline i32 s_smmul(i32 a, i32 b) { return (i32)(((i64)a * b) >> 32); }
static inline i32 s_rshift_round(i32 a, int s) { return s == 1 ? (a >> 1) + (a & 1) : ((a >> (s - 1)) + 1) >> 1; }
static inline i32 s_lshift(i32 a, int s) { return (i32)((u32)a << s); }
static inline i32 s_abs(i32 a) { return a > 0 ? a : -a; }
static inline int s_clz32(i32 x) { return x ? __builtin_clz((u32)x) : 32; }
static inline int s_clz64(i64 x) { i32 hi = (i32)(x >> 32); return hi == 0 ? 32 + s_clz32((i32)x) : s_clz32(hi); }
static inline i32 s_limit(i32 a, i32 l1, i32 l2) { return l1 > l2 ? (a > l1 ? l1 : (a < l2 ? l2 : a)) : (a > l2 ? l2 : (a < l1 ? l1 : a)); }
static inline i32 s_lshift_sat32(i32 a, int s) { return s_lshift(s_limit(a, (i32)0x80000000 >> s, 0x7FFFFFFF >> s), s); }

static i32 s_div32_varq(i32 a32, i32 b32, int Qres)                 /* Inlines.h:96-139 */
{
    int a_headrm = s_clz32(s_abs(a32)) - 1;
    i32 a32_nrm = s_lshift(a32, a_headrm);
    int b_headrm = s_clz32(s_abs(b32)) - 1;
    i32 b32_nrm = s_lshift(b32, b_headrm);
    i32 b32_inv = (0x7FFFFFFF >> 2) / (b32_nrm >> 16);
    i32 result = s_smulwb(a32_nrm, b32_inv);
    a32_nrm = (i32)((u32)a32_nrm - ((u32)s_smmul(b32_nrm, result) << 3));
    result = s_smlawb(result, a32_nrm, b32_inv);
    int lshift = 29 + a_headrm - b_headrm - Qres;
    if (lshift < 0) return s_lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

static i32 s_inverse32_varq(i32 b32, int Qres)                      /* Inlines.h:142-186 */
{
    int b_headrm = s_clz32(s_abs(b32)) - 1;
    i32 b32_nrm = s_lshift(b32, b_headrm);
    i32 b32_inv = (0x7FFFFFFF >> 2) / (b32_nrm >> 16);
    i32 result = s_lshift(b32_inv, 16);
    i32 err_Q32 = s_lshift(((i32)1 << 29) - s_smulwb(b32_nrm, b32_inv), 3);
    result = s_smlaww(result, err_Q32, b32_inv);
    int lshift = 61 - b_headrm - Qres;
    if (lshift <= 0) return s_lshift_sat32(result, -lshift);
    return lshift < 32 ? result >> lshift : 0;
}

static i32 s_sqrt_approx(i32 x)                                     /* Inlines.h:68-93 */
{
    if (x <= 0) return 0;
    int lz = s_clz32(x);
    int rot = 24 - lz;
    u32 ux = (u32)x;
    i32 frac_Q7 = (rot == 0 ? x : rot < 0 ? (i32)((ux << (u32)-rot) | (ux >> (32 - (u32)-rot))) : (i32)((ux << (32 - rot)) | (ux >> rot))) & 0x7f;
    i32 y = (lz & 1) ? 32768 : 46214;
    y >>= (lz >> 1);
    return s_smlawb(y, y, s_smulbb(213, frac_Q7));
}

#define QA 25
#define COND_FAC_Q32 42950                                           /* SILK_FIX_CONST(FIND_LPC_COND_FAC = 1e-5f, 32) */

void orc_silk_burg_modified(const opusgpu_burg_in *in, opusgpu_burg_out *out)
{
    const i16 *x = in->x;
    const int subfr_length = in->subfr_length, nb_subfr = in->nb_subfr, D = in->D;
    const i32 minInvGain_Q30 = in->minInvGain_Q30;
    i32 C_first_row[16], C_last_row[16], Af_QA[16], CAf[17], CAb[17], xcorr[16];
    i32 C0, num, nrg, rc_Q31, invGain_Q30, Atmp_QA, Atmp1, tmp1, tmp2, x1, x2;
    int k, n, s, lz, rshifts, reached_max_gain;
    i64 C0_64 = 0;
    for (k = 0; k < subfr_length * nb_subfr; k++) C0_64 += (i32)x[k] * (i32)x[k];
    lz = s_clz64(C0_64);
    rshifts = 32 + 1 + 2 - lz;
    if (rshifts > 32 - QA) rshifts = 32 - QA;
    if (rshifts < -16) rshifts = -16;
    C0 = rshifts > 0 ? (i32)(C0_64 >> rshifts) : s_lshift((i32)C0_64, -rshifts);
    CAb[0] = CAf[0] = C0 + s_smmul(COND_FAC_Q32, C0) + 1;
    memset(C_first_row, 0, sizeof(C_first_row));
    memset(Af_QA, 0, sizeof(Af_QA));
    if (rshifts > 0) {
        for (s = 0; s < nb_subfr; s++) {
            const i16 *xp = x + s * subfr_length;
            for (n = 1; n < D + 1; n++) {
                i64 acc = 0;
                for (k = 0; k < subfr_length - n; k++) acc += (i32)xp[k] * (i32)xp[k + n];
                C_first_row[n - 1] += (i32)(acc >> rshifts);
            }
        }
    } else {
        for (s = 0; s < nb_subfr; s++) {
            const i16 *xp = x + s * subfr_length;
            for (n = 1; n < D + 1; n++) {
                i32 d = 0;                                           /* celt_pitch_xcorr + tail = full lag-n product, 32-bit wrap */
                for (k = n; k < subfr_length; k++) d += (i32)xp[k] * (i32)xp[k - n];
                xcorr[n - 1] = d;
            }
            for (n = 1; n < D + 1; n++) C_first_row[n - 1] += s_lshift(xcorr[n - 1], -rshifts);
        }
    }
    memcpy(C_last_row, C_first_row, sizeof(C_first_row));
    CAb[0] = CAf[0] = C0 + s_smmul(COND_FAC_Q32, C0) + 1;
    invGain_Q30 = (i32)1 << 30;
    reached_max_gain = 0;
    for (n = 0; n < D; n++) {
        if (rshifts > -2) {
            for (s = 0; s < nb_subfr; s++) {
                const i16 *xp = x + s * subfr_length;
                x1 = -s_lshift((i32)xp[n], 16 - rshifts);
                x2 = -s_lshift((i32)xp[subfr_length - n - 1], 16 - rshifts);
                tmp1 = s_lshift((i32)xp[n], QA - 16);
                tmp2 = s_lshift((i32)xp[subfr_length - n - 1], QA - 16);
                for (k = 0; k < n; k++) {
                    C_first_row[k] = s_smlawb(C_first_row[k], x1, xp[n - k - 1]);
                    C_last_row[k] = s_smlawb(C_last_row[k], x2, xp[subfr_length - n + k]);
                    Atmp_QA = Af_QA[k];
                    tmp1 = s_smlawb(tmp1, Atmp_QA, xp[n - k - 1]);
                    tmp2 = s_smlawb(tmp2, Atmp_QA, xp[subfr_length - n + k]);
                }
                tmp1 = s_lshift(-tmp1, 32 - QA - rshifts);
                tmp2 = s_lshift(-tmp2, 32 - QA - rshifts);
                for (k = 0; k <= n; k++) {
                    CAf[k] = s_smlawb(CAf[k], tmp1, xp[n - k]);
                    CAb[k] = s_smlawb(CAb[k], tmp2, xp[subfr_length - n + k - 1]);
                }
            }
        } else {
            for (s = 0; s < nb_subfr; s++) {
                const i16 *xp = x + s * subfr_length;
                x1 = -s_lshift((i32)xp[n], -rshifts);
                x2 = -s_lshift((i32)xp[subfr_length - n - 1], -rshifts);
                tmp1 = s_lshift((i32)xp[n], 17);
                tmp2 = s_lshift((i32)xp[subfr_length - n - 1], 17);
                for (k = 0; k < n; k++) {
                    C_first_row[k] = C_first_row[k] + x1 * xp[n - k - 1];
                    C_last_row[k] = C_last_row[k] + x2 * xp[subfr_length - n + k];
                    Atmp1 = s_rshift_round(Af_QA[k], QA - 17);
                    tmp1 = tmp1 + xp[n - k - 1] * Atmp1;
                    tmp2 = tmp2 + xp[subfr_length - n + k] * Atmp1;
                }
                tmp1 = -tmp1;
                tmp2 = -tmp2;
                for (k = 0; k <= n; k++) {
                    CAf[k] = s_smlaww(CAf[k], tmp1, s_lshift((i32)xp[n - k], -rshifts - 1));
                    CAb[k] = s_smlaww(CAb[k], tmp2, s_lshift((i32)xp[subfr_length - n + k - 1], -rshifts - 1));
                }
            }
        }
        tmp1 = C_first_row[n];
        tmp2 = C_last_row[n];
        num = 0;
        nrg = CAb[0] + CAf[0];
        for (k = 0; k < n; k++) {
            Atmp_QA = Af_QA[k];
            lz = s_clz32(s_abs(Atmp_QA)) - 1;
            if (lz > 32 - QA) lz = 32 - QA;
            Atmp1 = s_lshift(Atmp_QA, lz);
            tmp1 = tmp1 + s_lshift(s_smmul(C_last_row[n - k - 1], Atmp1), 32 - QA - lz);
            tmp2 = tmp2 + s_lshift(s_smmul(C_first_row[n - k - 1], Atmp1), 32 - QA - lz);
            num = num + s_lshift(s_smmul(CAb[n - k], Atmp1), 32 - QA - lz);
            nrg = nrg + s_lshift(s_smmul(CAb[k + 1] + CAf[k + 1], Atmp1), 32 - QA - lz);
        }
        CAf[n + 1] = tmp1;
        CAb[n + 1] = tmp2;
        num = num + tmp2;
        num = s_lshift(-num, 1);
        if (s_abs(num) < nrg) rc_Q31 = s_div32_varq(num, nrg, 31);
        else rc_Q31 = (num > 0) ? 0x7FFFFFFF : (i32)0x80000000;
        tmp1 = ((i32)1 << 30) - s_smmul(rc_Q31, rc_Q31);
        tmp1 = s_lshift(s_smmul(invGain_Q30, tmp1), 2);
        if (tmp1 <= minInvGain_Q30) {
            tmp2 = ((i32)1 << 30) - s_div32_varq(minInvGain_Q30, invGain_Q30, 30);
            rc_Q31 = s_sqrt_approx(tmp2);
            rc_Q31 = (rc_Q31 + tmp2 / rc_Q31) >> 1;
            rc_Q31 = s_lshift(rc_Q31, 16);
            if (num < 0) rc_Q31 = -rc_Q31;
            invGain_Q30 = minInvGain_Q30;
            reached_max_gain = 1;
        } else {
            invGain_Q30 = tmp1;
        }
        for (k = 0; k < (n + 1) >> 1; k++) {
            tmp1 = Af_QA[k];
            tmp2 = Af_QA[n - k - 1];
            Af_QA[k] = tmp1 + s_lshift(s_smmul(tmp2, rc_Q31), 1);
            Af_QA[n - k - 1] = tmp2 + s_lshift(s_smmul(tmp1, rc_Q31), 1);
        }
        Af_QA[n] = rc_Q31 >> (31 - QA);
        if (reached_max_gain) {
            for (k = n + 1; k < D; k++) Af_QA[k] = 0;
            break;
        }
        for (k = 0; k <= n + 1; k++) {
            tmp1 = CAf[k];
            tmp2 = CAb[n - k + 1];
            CAf[k] = tmp1 + s_lshift(s_smmul(tmp2, rc_Q31), 1);
            CAb[n - k + 1] = tmp2 + s_lshift(s_smmul(tmp1, rc_Q31), 1);
        }
    }
    if (reached_max_gain) {
        for (k = 0; k < D; k++) out->A_Q16[k] = -s_rshift_round(Af_QA[k], QA - 16);
        if (rshifts > 0) {
            for (s = 0; s < nb_subfr; s++) {
                const i16 *xp = x + s * subfr_length;
                i64 acc = 0;
                for (k = 0; k < D; k++) acc += (i32)xp[k] * (i32)xp[k];
                C0 -= (i32)(acc >> rshifts);
            }
        } else {
            for (s = 0; s < nb_subfr; s++) {
                const i16 *xp = x + s * subfr_length;
                i32 acc = 0;
                for (k = 0; k < D; k++) acc += (i32)xp[k] * (i32)xp[k];
                C0 -= s_lshift(acc, -rshifts);
            }
        }
        out->res_nrg = s_lshift(s_smmul(invGain_Q30, C0), 2);
        out->res_nrg_Q = -rshifts;
    } else {
        nrg = CAf[0];
        tmp1 = (i32)1 << 16;
        for (k = 0; k < D; k++) {
            Atmp1 = s_rshift_round(Af_QA[k], QA - 16);
            nrg = s_smlaww(nrg, CAf[k + 1], Atmp1);
            tmp1 = s_smlaww(tmp1, Atmp1, Atmp1);
            out->A_Q16[k] = -Atmp1;
        }
        out->res_nrg = s_smlaww(nrg, s_smmul(COND_FAC_Q32, C0), -tmp1);
        out->res_nrg_Q = -rshifts;
    }
    for (k = D; k < 16; k++) out->A_Q16[k] = 0;
}

/* silk_LPC_analysis_filter, FIXED_POINT branch */
static void lpc_analysis_filter(i16 *outp, const i16 *inp, const i16 *B, int len, int d)
{
    for (int ix = d; ix < len; ix++) {
        i32 sum = 0;
        for (int m = 0; m < d; m++) sum += (i32)(i16)(-B[m]) * (i32)inp[ix - 1 - m];
        i32 v = (i32)inp[ix] + pshr32(sum, 12);
        outp[ix] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    }
    for (int j = 0; j < d; j++) outp[j] = 0;
}

static const i16 QUANT_OFFSETS_Q10[2][2] = {{100, 240}, {32, 100}};   /* silk/tables_other.c:95-97 */

void orc_silk_nsq(const opusgpu_nsq_in *in, opusgpu_nsq_state *NSQ, opusgpu_nsq_out *outp)
{
    i32 sLTP_Q15[640];
    i16 sLTP[640];
    i32 x_sc_Q10[80];
    const int nb_subfr = in->nb_subfr, subfr_length = in->subfr_length, frame_length = in->frame_length;
    const int ltp_mem_length = in->ltp_mem_length, predictLPCOrder = in->predictLPCOrder, shapingLPCOrder = in->shapingLPCOrder;
    const int signalType = in->signalType;
    const i32 *x_Q3 = in->x_Q3;
    i8 *pulses = outp->pulses;
    int lag, k;

    NSQ->rand_seed = in->Seed;
    lag = NSQ->lagPrev;
    const int offset_Q10 = QUANT_OFFSETS_Q10[signalType >> 1][in->quantOffsetType];
    const int LSF_interpolation_flag = in->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    NSQ->sLTP_shp_buf_idx = ltp_mem_length;
    NSQ->sLTP_buf_idx = ltp_mem_length;
    i16 *pxq = &NSQ->xq[ltp_mem_length];
    for (k = 0; k < nb_subfr; k++) {
        const i16 *A_Q12 = &in->PredCoef_Q12[((k >> 1) | (1 - LSF_interpolation_flag)) * 16];
        const i16 *B_Q14 = &in->LTPCoef_Q14[k * 5];
        const i16 *AR_shp_Q13 = &in->AR2_Q13[k * 16];
        i32 HarmShapeFIRPacked_Q14 = in->HarmShapeGain_Q14[k] >> 2;
        HarmShapeFIRPacked_Q14 |= s_lshift((i32)(in->HarmShapeGain_Q14[k] >> 1), 16);
        NSQ->rewhite_flag = 0;
        if (signalType == 2) {
            lag = in->pitchL[k];
            if ((k & (3 - (LSF_interpolation_flag << 1))) == 0) {
                int start_idx = ltp_mem_length - lag - predictLPCOrder - 5 / 2;
                lpc_analysis_filter(&sLTP[start_idx], &NSQ->xq[start_idx + k * subfr_length], A_Q12, ltp_mem_length - start_idx, predictLPCOrder);
                NSQ->rewhite_flag = 1;
                NSQ->sLTP_buf_idx = ltp_mem_length;
            }
        }
        /* ---- silk_nsq_scale_states ---- */
        {
            int i, lg = in->pitchL[k];
            i32 gain = in->Gains_Q16[k];
            i32 inv_gain_Q31 = s_inverse32_varq(gain > 1 ? gain : 1, 47);
            i32 gain_adj_Q16 = gain != NSQ->prev_gain_Q16 ? s_div32_varq(NSQ->prev_gain_Q16, gain, 16) : (i32)1 << 16;
            i32 inv_gain_Q23 = s_rshift_round(inv_gain_Q31, 8);
            for (i = 0; i < subfr_length; i++) x_sc_Q10[i] = s_smulww(x_Q3[i], inv_gain_Q23);
            NSQ->prev_gain_Q16 = gain;
            if (NSQ->rewhite_flag) {
                if (k == 0) inv_gain_Q31 = s_lshift(s_smulwb(inv_gain_Q31, in->LTP_scale_Q14), 2);
                for (i = NSQ->sLTP_buf_idx - lg - 5 / 2; i < NSQ->sLTP_buf_idx; i++) sLTP_Q15[i] = s_smulwb(inv_gain_Q31, sLTP[i]);
            }
            if (gain_adj_Q16 != (i32)1 << 16) {
                for (i = NSQ->sLTP_shp_buf_idx - ltp_mem_length; i < NSQ->sLTP_shp_buf_idx; i++)
                    NSQ->sLTP_shp_Q14[i] = s_smulww(gain_adj_Q16, NSQ->sLTP_shp_Q14[i]);
                if (signalType == 2 && NSQ->rewhite_flag == 0)
                    for (i = NSQ->sLTP_buf_idx - lg - 5 / 2; i < NSQ->sLTP_buf_idx; i++) sLTP_Q15[i] = s_smulww(gain_adj_Q16, sLTP_Q15[i]);
                NSQ->sLF_AR_shp_Q14 = s_smulww(gain_adj_Q16, NSQ->sLF_AR_shp_Q14);
                for (i = 0; i < 32; i++) NSQ->sLPC_Q14[i] = s_smulww(gain_adj_Q16, NSQ->sLPC_Q14[i]);
                for (i = 0; i < 16; i++) NSQ->sAR2_Q14[i] = s_smulww(gain_adj_Q16, NSQ->sAR2_Q14[i]);
            }
        }
        /* ---- silk_noise_shape_quantizer ---- */
        {
            const i32 Gain_Q10 = in->Gains_Q16[k] >> 6;
            const int Tilt_Q14 = in->Tilt_Q14[k], Lambda_Q10 = in->Lambda_Q10;
            const i32 LF_shp_Q14 = in->LF_shp_Q14[k];
            i32 *shp_lag_ptr = &NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - lag + 3 / 2];
            i32 *pred_lag_ptr = &sLTP_Q15[NSQ->sLTP_buf_idx - lag + 5 / 2];
            i32 *psLPC_Q14 = &NSQ->sLPC_Q14[32 - 1];
            for (int i = 0; i < subfr_length; i++) {
                i32 LTP_pred_Q13, LPC_pred_Q10, n_AR_Q12, n_LTP_Q13, n_LF_Q12, r_Q10, rr_Q10, q1_Q0, q1_Q10, q2_Q10, rd1_Q20, rd2_Q20;
                i32 exc_Q14, LPC_exc_Q14, xq_Q14, tmp1, tmp2, sLF_AR_shp_Q14;
                NSQ->rand_seed = (i32)(907633515u + (u32)NSQ->rand_seed * 196314165u);
                LPC_pred_Q10 = predictLPCOrder >> 1;
                for (int j = 0; j < predictLPCOrder; j++) LPC_pred_Q10 = s_smlawb(LPC_pred_Q10, psLPC_Q14[-j], A_Q12[j]);
                if (signalType == 2) {
                    LTP_pred_Q13 = 2;
                    for (int j = 0; j < 5; j++) LTP_pred_Q13 = s_smlawb(LTP_pred_Q13, pred_lag_ptr[-j], B_Q14[j]);
                    pred_lag_ptr++;
                } else {
                    LTP_pred_Q13 = 0;
                }
                tmp2 = psLPC_Q14[0];
                tmp1 = NSQ->sAR2_Q14[0];
                NSQ->sAR2_Q14[0] = tmp2;
                n_AR_Q12 = shapingLPCOrder >> 1;
                n_AR_Q12 = s_smlawb(n_AR_Q12, tmp2, AR_shp_Q13[0]);
                for (int j = 2; j < shapingLPCOrder; j += 2) {
                    tmp2 = NSQ->sAR2_Q14[j - 1];
                    NSQ->sAR2_Q14[j - 1] = tmp1;
                    n_AR_Q12 = s_smlawb(n_AR_Q12, tmp1, AR_shp_Q13[j - 1]);
                    tmp1 = NSQ->sAR2_Q14[j + 0];
                    NSQ->sAR2_Q14[j + 0] = tmp2;
                    n_AR_Q12 = s_smlawb(n_AR_Q12, tmp2, AR_shp_Q13[j]);
                }
                NSQ->sAR2_Q14[shapingLPCOrder - 1] = tmp1;
                n_AR_Q12 = s_smlawb(n_AR_Q12, tmp1, AR_shp_Q13[shapingLPCOrder - 1]);
                n_AR_Q12 = s_lshift(n_AR_Q12, 1);
                n_AR_Q12 = s_smlawb(n_AR_Q12, NSQ->sLF_AR_shp_Q14, Tilt_Q14);
                n_LF_Q12 = s_smulwb(NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - 1], LF_shp_Q14);
                n_LF_Q12 = s_smlawt(n_LF_Q12, NSQ->sLF_AR_shp_Q14, LF_shp_Q14);
                tmp1 = s_lshift(LPC_pred_Q10, 2) - n_AR_Q12;
                tmp1 = tmp1 - n_LF_Q12;
                if (lag > 0) {
                    n_LTP_Q13 = s_smulwb(shp_lag_ptr[0] + shp_lag_ptr[-2], HarmShapeFIRPacked_Q14);
                    n_LTP_Q13 = s_smlawt(n_LTP_Q13, shp_lag_ptr[-1], HarmShapeFIRPacked_Q14);
                    n_LTP_Q13 = s_lshift(n_LTP_Q13, 1);
                    shp_lag_ptr++;
                    tmp2 = LTP_pred_Q13 - n_LTP_Q13;
                    tmp1 = tmp2 + s_lshift(tmp1, 1);
                    tmp1 = s_rshift_round(tmp1, 3);
                } else {
                    tmp1 = s_rshift_round(tmp1, 2);
                }
                r_Q10 = x_sc_Q10[i] - tmp1;
                if (NSQ->rand_seed < 0) r_Q10 = -r_Q10;
                r_Q10 = s_limit(r_Q10, -(31 << 10), 30 << 10);
                q1_Q10 = r_Q10 - offset_Q10;
                q1_Q0 = q1_Q10 >> 10;
                if (q1_Q0 > 0) {
                    q1_Q10 = s_lshift(q1_Q0, 10) - 80;
                    q1_Q10 = q1_Q10 + offset_Q10;
                    q2_Q10 = q1_Q10 + 1024;
                    rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                    rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                } else if (q1_Q0 == 0) {
                    q1_Q10 = offset_Q10;
                    q2_Q10 = q1_Q10 + (1024 - 80);
                    rd1_Q20 = s_smulbb(q1_Q10, Lambda_Q10);
                    rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                } else if (q1_Q0 == -1) {
                    q2_Q10 = offset_Q10;
                    q1_Q10 = q2_Q10 - (1024 - 80);
                    rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                    rd2_Q20 = s_smulbb(q2_Q10, Lambda_Q10);
                } else {
                    q1_Q10 = s_lshift(q1_Q0, 10) + 80;
                    q1_Q10 = q1_Q10 + offset_Q10;
                    q2_Q10 = q1_Q10 + 1024;
                    rd1_Q20 = s_smulbb(-q1_Q10, Lambda_Q10);
                    rd2_Q20 = s_smulbb(-q2_Q10, Lambda_Q10);
                }
                rr_Q10 = r_Q10 - q1_Q10;
                rd1_Q20 = rd1_Q20 + s_smulbb(rr_Q10, rr_Q10);
                rr_Q10 = r_Q10 - q2_Q10;
                rd2_Q20 = rd2_Q20 + s_smulbb(rr_Q10, rr_Q10);
                if (rd2_Q20 < rd1_Q20) q1_Q10 = q2_Q10;
                pulses[i] = (i8)s_rshift_round(q1_Q10, 10);
                exc_Q14 = s_lshift(q1_Q10, 4);
                if (NSQ->rand_seed < 0) exc_Q14 = -exc_Q14;
                LPC_exc_Q14 = exc_Q14 + s_lshift(LTP_pred_Q13, 1);
                xq_Q14 = LPC_exc_Q14 + s_lshift(LPC_pred_Q10, 4);
                {   /* silk_SAT16(silk_RSHIFT_ROUND(silk_SMULWW(xq_Q14, Gain_Q10), 8)) evaluated in 64 bits */
                    i64 t = ((i64)xq_Q14 * Gain_Q10) >> 16;
                    t = ((t >> 7) + 1) >> 1;
                    pxq[i] = (i16)(t > 32767 ? 32767 : (t < -32768 ? -32768 : t));
                }
                psLPC_Q14++;
                *psLPC_Q14 = xq_Q14;
                sLF_AR_shp_Q14 = xq_Q14 - s_lshift(n_AR_Q12, 2);
                NSQ->sLF_AR_shp_Q14 = sLF_AR_shp_Q14;
                NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx] = sLF_AR_shp_Q14 - s_lshift(n_LF_Q12, 2);
                sLTP_Q15[NSQ->sLTP_buf_idx] = s_lshift(LPC_exc_Q14, 1);
                NSQ->sLTP_shp_buf_idx++;
                NSQ->sLTP_buf_idx++;
                NSQ->rand_seed = (i32)((u32)NSQ->rand_seed + (u32)(i32)pulses[i]);
            }
            memcpy(NSQ->sLPC_Q14, &NSQ->sLPC_Q14[subfr_length], 32 * sizeof(i32));
        }
        x_Q3 += subfr_length;
        pulses += subfr_length;
        pxq += subfr_length;
    }
    NSQ->lagPrev = in->pitchL[nb_subfr - 1];
    memmove(NSQ->xq, &NSQ->xq[frame_length], ltp_mem_length * sizeof(i16));
    memmove(NSQ->sLTP_shp_Q14, &NSQ->sLTP_shp_Q14[frame_length], ltp_mem_length * sizeof(i32));
}

/* ---- silk_NSQ_del_dec_c and its helpers: opus-fix/silk/NSQ_del_dec.c:112-318 (driver), :324-630 (quantizer of one
 * subframe), :632-724 (state scaling). One candidate path ("delayed-decision state") is a DdState; the field order is the
 * reference's NSQ_del_dec_struct (:35-47) because the survivor copy at :583-584 is a memcpy from int32 offset i to the
 * end of the struct. ---- */
#define DD_DELAY 32                                     /* DECISION_DELAY, silk/define.h:157 */
typedef struct {
    i32 lpc[80 + 32];                                   /* sLPC_Q14 */
    i32 rnd[DD_DELAY], q[DD_DELAY], xq[DD_DELAY], pred[DD_DELAY], shape[DD_DELAY];
    i32 ar2[16];
    i32 lf_ar, seed, seed0, rd;
} DdState;
typedef struct { i32 q, rd, xq, lf_ar, shp, exc; } DdCand;      /* NSQ_sample_struct :49-56 */

static int dd_winner(const DdState *dd, int n)          /* :193-201, :285-292: first strict minimum */
{
    int w = 0;
    for (int k = 1; k < n; k++) if (dd[k].rd < dd[w].rd) w = k;
    return w;
}

void orc_silk_nsq_del_dec(const opusgpu_nsq_dd_in *din, opusgpu_nsq_state *NSQ, opusgpu_nsq_dd_out *outp)
{
    const opusgpu_nsq_in *in = &din->base;
    const int nst = din->nStatesDelayedDecision, warping_Q16 = din->warping_Q16;
    const int nb_subfr = in->nb_subfr, L = in->subfr_length, frame_length = in->frame_length;
    const int ltp_mem = in->ltp_mem_length, pord = in->predictLPCOrder, sord = in->shapingLPCOrder;
    const int voiced = in->signalType == 2;
    i32 sLTP_Q15[640], x_sc_Q10[80], delayedGain_Q10[DD_DELAY];
    i16 sLTP[640];
    DdState dd[4];
    DdCand cand[4][2];
    const i32 *x_Q3 = in->x_Q3;
    i8 *pulses = outp->pulses;
    int lag = NSQ->lagPrev, k, i, j;

    memset(dd, 0, sizeof(dd));
    memset(delayedGain_Q10, 0, sizeof(delayedGain_Q10));
    for (k = 0; k < nst; k++) {
        dd[k].seed = dd[k].seed0 = (k + in->Seed) & 3;
        dd[k].lf_ar = NSQ->sLF_AR_shp_Q14;
        dd[k].shape[0] = NSQ->sLTP_shp_Q14[ltp_mem - 1];
        memcpy(dd[k].lpc, NSQ->sLPC_Q14, 32 * sizeof(i32));
        memcpy(dd[k].ar2, NSQ->sAR2_Q14, sizeof(dd[k].ar2));
    }
    const int offset_Q10 = QUANT_OFFSETS_Q10[in->signalType >> 1][in->quantOffsetType];
    int smpl = 0;                                       /* index of the oldest entry of the 32-deep rings */
    int delay = L < DD_DELAY ? L : DD_DELAY;
    if (voiced) {
        for (k = 0; k < nb_subfr; k++) if (in->pitchL[k] - 5 / 2 - 1 < delay) delay = in->pitchL[k] - 5 / 2 - 1;
    } else if (lag > 0 && lag - 5 / 2 - 1 < delay) {
        delay = lag - 5 / 2 - 1;
    }
    const int interp = in->NLSFInterpCoef_Q2 == 4 ? 0 : 1;
    i16 *pxq = &NSQ->xq[ltp_mem];
    NSQ->sLTP_shp_buf_idx = ltp_mem;
    NSQ->sLTP_buf_idx = ltp_mem;
    int subfr = 0;
    for (k = 0; k < nb_subfr; k++) {
        const i16 *A_Q12 = &in->PredCoef_Q12[((k >> 1) | (1 - interp)) * 16];
        const i16 *B_Q14 = &in->LTPCoef_Q14[k * 5];
        const i16 *AR_Q13 = &in->AR2_Q13[k * 16];
        i32 harm = in->HarmShapeGain_Q14[k] >> 2;
        harm |= s_lshift((i32)(in->HarmShapeGain_Q14[k] >> 1), 16);
        NSQ->rewhite_flag = 0;
        if (voiced) {
            lag = in->pitchL[k];
            if ((k & (3 - (interp << 1))) == 0) {
                if (k == 2) {                           /* :190-221: flush the survivor before the filters change */
                    int w = dd_winner(dd, nst);
                    for (i = 0; i < nst; i++) if (i != w) dd[i].rd = (i32)((u32)dd[i].rd + (0x7FFFFFFF >> 4));
                    int last = smpl + delay;
                    for (i = 0; i < delay; i++) {
                        last = (last - 1) & (DD_DELAY - 1);
                        pulses[i - delay] = (i8)s_rshift_round(dd[w].q[last], 10);
                        i32 v = s_rshift_round(s_smulww(dd[w].xq[last], in->Gains_Q16[1]), 14);
                        pxq[i - delay] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
                        NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - delay + i] = dd[w].shape[last];
                    }
                    subfr = 0;
                }
                int start_idx = ltp_mem - lag - pord - 5 / 2;
                lpc_analysis_filter(&sLTP[start_idx], &NSQ->xq[start_idx + k * L], A_Q12, ltp_mem - start_idx, pord);
                NSQ->sLTP_buf_idx = ltp_mem;
                NSQ->rewhite_flag = 1;
            }
        }
        /* ---- silk_nsq_del_dec_scale_states :632-724 ---- */
        {
            const int lg = in->pitchL[k];
            const i32 gain = in->Gains_Q16[k];
            i32 inv_gain_Q31 = s_inverse32_varq(gain > 1 ? gain : 1, 47);
            const i32 adj = gain != NSQ->prev_gain_Q16 ? s_div32_varq(NSQ->prev_gain_Q16, gain, 16) : (i32)1 << 16;
            const i32 inv_gain_Q23 = s_rshift_round(inv_gain_Q31, 8);
            for (i = 0; i < L; i++) x_sc_Q10[i] = s_smulww(x_Q3[i], inv_gain_Q23);
            NSQ->prev_gain_Q16 = gain;
            if (NSQ->rewhite_flag) {
                if (k == 0) inv_gain_Q31 = s_lshift(s_smulwb(inv_gain_Q31, in->LTP_scale_Q14), 2);
                for (i = NSQ->sLTP_buf_idx - lg - 5 / 2; i < NSQ->sLTP_buf_idx; i++) sLTP_Q15[i] = s_smulwb(inv_gain_Q31, sLTP[i]);
            }
            if (adj != (i32)1 << 16) {
                for (i = NSQ->sLTP_shp_buf_idx - ltp_mem; i < NSQ->sLTP_shp_buf_idx; i++)
                    NSQ->sLTP_shp_Q14[i] = s_smulww(adj, NSQ->sLTP_shp_Q14[i]);
                if (voiced && NSQ->rewhite_flag == 0)
                    for (i = NSQ->sLTP_buf_idx - lg - 5 / 2; i < NSQ->sLTP_buf_idx - delay; i++) sLTP_Q15[i] = s_smulww(adj, sLTP_Q15[i]);
                for (j = 0; j < nst; j++) {
                    dd[j].lf_ar = s_smulww(adj, dd[j].lf_ar);
                    for (i = 0; i < 32; i++) dd[j].lpc[i] = s_smulww(adj, dd[j].lpc[i]);
                    for (i = 0; i < 16; i++) dd[j].ar2[i] = s_smulww(adj, dd[j].ar2[i]);
                    for (i = 0; i < DD_DELAY; i++) {
                        dd[j].pred[i] = s_smulww(adj, dd[j].pred[i]);
                        dd[j].shape[i] = s_smulww(adj, dd[j].shape[i]);
                    }
                }
            }
        }
        /* ---- silk_noise_shape_quantizer_del_dec :324-630 ---- */
        {
            const i32 Gain_Q10 = in->Gains_Q16[k] >> 6;
            const int Tilt_Q14 = in->Tilt_Q14[k], Lambda_Q10 = in->Lambda_Q10;
            const i32 LF_shp_Q14 = in->LF_shp_Q14[k];
            const i32 *shp_lag = &NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - lag + 3 / 2];
            const i32 *pred_lag = &sLTP_Q15[NSQ->sLTP_buf_idx - lag + 5 / 2];
            for (i = 0; i < L; i++) {
                i32 LTP_pred_Q14 = 0, n_LTP_Q14 = 0;
                if (voiced) {
                    LTP_pred_Q14 = 2;
                    for (j = 0; j < 5; j++) LTP_pred_Q14 = s_smlawb(LTP_pred_Q14, pred_lag[-j], B_Q14[j]);
                    LTP_pred_Q14 = s_lshift(LTP_pred_Q14, 1);
                    pred_lag++;
                }
                if (lag > 0) {
                    n_LTP_Q14 = s_smulwb((i32)((u32)shp_lag[0] + (u32)shp_lag[-2]), harm);
                    n_LTP_Q14 = s_smlawt(n_LTP_Q14, shp_lag[-1], harm);
                    n_LTP_Q14 = (i32)((u32)LTP_pred_Q14 - ((u32)n_LTP_Q14 << 2));
                    shp_lag++;
                }
                for (int s = 0; s < nst; s++) {
                    DdState *d = &dd[s];
                    DdCand *c = cand[s];
                    d->seed = (i32)(907633515u + (u32)d->seed * 196314165u);
                    const i32 *lp = &d->lpc[32 - 1 + i];
                    i32 LPC_pred_Q14 = pord >> 1;
                    for (j = 0; j < pord; j++) LPC_pred_Q14 = s_smlawb(LPC_pred_Q14, lp[-j], A_Q12[j]);
                    LPC_pred_Q14 = s_lshift(LPC_pred_Q14, 4);
                    /* warped noise-shaping filter: a chain of first-order all-pass sections (:419-440) */
                    i32 tmp2 = s_smlawb(lp[0], d->ar2[0], warping_Q16);
                    i32 tmp1 = s_smlawb(d->ar2[0], (i32)((u32)d->ar2[1] - (u32)tmp2), warping_Q16);
                    d->ar2[0] = tmp2;
                    i32 n_AR_Q14 = sord >> 1;
                    n_AR_Q14 = s_smlawb(n_AR_Q14, tmp2, AR_Q13[0]);
                    for (j = 2; j < sord; j += 2) {
                        tmp2 = s_smlawb(d->ar2[j - 1], (i32)((u32)d->ar2[j] - (u32)tmp1), warping_Q16);
                        d->ar2[j - 1] = tmp1;
                        n_AR_Q14 = s_smlawb(n_AR_Q14, tmp1, AR_Q13[j - 1]);
                        tmp1 = s_smlawb(d->ar2[j], (i32)((u32)d->ar2[j + 1] - (u32)tmp2), warping_Q16);
                        d->ar2[j] = tmp2;
                        n_AR_Q14 = s_smlawb(n_AR_Q14, tmp2, AR_Q13[j]);
                    }
                    d->ar2[sord - 1] = tmp1;
                    n_AR_Q14 = s_smlawb(n_AR_Q14, tmp1, AR_Q13[sord - 1]);
                    n_AR_Q14 = s_lshift(n_AR_Q14, 1);
                    n_AR_Q14 = s_smlawb(n_AR_Q14, d->lf_ar, Tilt_Q14);
                    n_AR_Q14 = s_lshift(n_AR_Q14, 2);
                    i32 n_LF_Q14 = s_smulwb(d->shape[smpl], LF_shp_Q14);
                    n_LF_Q14 = s_smlawt(n_LF_Q14, d->lf_ar, LF_shp_Q14);
                    n_LF_Q14 = s_lshift(n_LF_Q14, 2);
                    tmp1 = (i32)((u32)n_AR_Q14 + (u32)n_LF_Q14);
                    tmp2 = (i32)((u32)n_LTP_Q14 + (u32)LPC_pred_Q14);
                    tmp1 = (i32)((u32)tmp2 - (u32)tmp1);
                    tmp1 = s_rshift_round(tmp1, 4);
                    i32 r_Q10 = (i32)((u32)x_sc_Q10[i] - (u32)tmp1);
                    if (d->seed < 0) r_Q10 = (i32)(0u - (u32)r_Q10);
                    r_Q10 = s_limit(r_Q10, -(31 << 10), 30 << 10);
                    i32 q1_Q10 = r_Q10 - offset_Q10, q2_Q10, rd1, rd2;
                    i32 q1_Q0 = q1_Q10 >> 10;
                    if (q1_Q0 > 0) {
                        q1_Q10 = s_lshift(q1_Q0, 10) - 80 + offset_Q10;
                        q2_Q10 = q1_Q10 + 1024;
                        rd1 = s_smulbb(q1_Q10, Lambda_Q10);
                        rd2 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else if (q1_Q0 == 0) {
                        q1_Q10 = offset_Q10;
                        q2_Q10 = q1_Q10 + (1024 - 80);
                        rd1 = s_smulbb(q1_Q10, Lambda_Q10);
                        rd2 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else if (q1_Q0 == -1) {
                        q2_Q10 = offset_Q10;
                        q1_Q10 = q2_Q10 - (1024 - 80);
                        rd1 = s_smulbb(-q1_Q10, Lambda_Q10);
                        rd2 = s_smulbb(q2_Q10, Lambda_Q10);
                    } else {
                        q1_Q10 = s_lshift(q1_Q0, 10) + 80 + offset_Q10;
                        q2_Q10 = q1_Q10 + 1024;
                        rd1 = s_smulbb(-q1_Q10, Lambda_Q10);
                        rd2 = s_smulbb(-q2_Q10, Lambda_Q10);
                    }
                    i32 rr = r_Q10 - q1_Q10;
                    rd1 = (rd1 + s_smulbb(rr, rr)) >> 10;
                    rr = r_Q10 - q2_Q10;
                    rd2 = (rd2 + s_smulbb(rr, rr)) >> 10;
                    const int first_is_q1 = rd1 < rd2;
                    c[0].rd = (i32)((u32)d->rd + (u32)(first_is_q1 ? rd1 : rd2));
                    c[1].rd = (i32)((u32)d->rd + (u32)(first_is_q1 ? rd2 : rd1));
                    c[0].q = first_is_q1 ? q1_Q10 : q2_Q10;
                    c[1].q = first_is_q1 ? q2_Q10 : q1_Q10;
                    for (j = 0; j < 2; j++) {
                        i32 exc_Q14 = s_lshift(c[j].q, 4);
                        if (d->seed < 0) exc_Q14 = -exc_Q14;
                        c[j].exc = (i32)((u32)exc_Q14 + (u32)LTP_pred_Q14);
                        c[j].xq = (i32)((u32)c[j].exc + (u32)LPC_pred_Q14);
                        c[j].lf_ar = (i32)((u32)c[j].xq - (u32)n_AR_Q14);
                        c[j].shp = (i32)((u32)c[j].lf_ar - (u32)n_LF_Q14);
                    }
                }
                smpl = (smpl - 1) & (DD_DELAY - 1);
                const int last = (smpl + delay) & (DD_DELAY - 1);
                int w = 0;
                for (j = 1; j < nst; j++) if (cand[j][0].rd < cand[w][0].rd) w = j;
                const i32 wrand = dd[w].rnd[last];
                for (j = 0; j < nst; j++) {
                    if (dd[j].rnd[last] != wrand) {      /* paths that disagree with the winner on the expiring sample (:544-552) */
                        cand[j][0].rd = (i32)((u32)cand[j][0].rd + (0x7FFFFFFF >> 4));
                        cand[j][1].rd = (i32)((u32)cand[j][1].rd + (0x7FFFFFFF >> 4));
                    }
                }
                int worst = 0, best2 = 0;
                for (j = 1; j < nst; j++) {
                    if (cand[j][0].rd > cand[worst][0].rd) worst = j;
                    if (cand[j][1].rd < cand[best2][1].rd) best2 = j;
                }
                if (cand[best2][1].rd < cand[worst][0].rd) {
                    memcpy((i32 *)&dd[worst] + i, (i32 *)&dd[best2] + i, sizeof(DdState) - i * sizeof(i32));
                    cand[worst][0] = cand[best2][1];
                }
                if (subfr > 0 || i >= delay) {
                    const DdState *d = &dd[w];
                    pulses[i - delay] = (i8)s_rshift_round(d->q[last], 10);
                    i32 v = s_rshift_round(s_smulww(d->xq[last], delayedGain_Q10[last]), 8);
                    pxq[i - delay] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
                    NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - delay] = d->shape[last];
                    sLTP_Q15[NSQ->sLTP_buf_idx - delay] = d->pred[last];
                }
                NSQ->sLTP_shp_buf_idx++;
                NSQ->sLTP_buf_idx++;
                for (j = 0; j < nst; j++) {
                    DdState *d = &dd[j];
                    const DdCand *c = &cand[j][0];
                    d->lf_ar = c->lf_ar;
                    d->lpc[32 + i] = c->xq;
                    d->xq[smpl] = c->xq;
                    d->q[smpl] = c->q;
                    d->pred[smpl] = s_lshift(c->exc, 1);
                    d->shape[smpl] = c->shp;
                    d->seed = (i32)((u32)d->seed + (u32)s_rshift_round(c->q, 10));
                    d->rnd[smpl] = d->seed;
                    d->rd = c->rd;
                }
                delayedGain_Q10[smpl] = Gain_Q10;
            }
            for (j = 0; j < nst; j++) memcpy(dd[j].lpc, &dd[j].lpc[L], 32 * sizeof(i32));
        }
        subfr++;
        x_Q3 += L;
        pulses += L;
        pxq += L;
    }
    {   /* :285-309: flush the last `delay` samples of the winning path */
        const int w = dd_winner(dd, nst);
        const DdState *d = &dd[w];
        outp->Seed = d->seed0;
        int last = smpl + delay;
        const i32 Gain_Q10 = in->Gains_Q16[nb_subfr - 1] >> 6;
        for (i = 0; i < delay; i++) {
            last = (last - 1) & (DD_DELAY - 1);
            pulses[i - delay] = (i8)s_rshift_round(d->q[last], 10);
            i32 v = s_rshift_round(s_smulww(d->xq[last], Gain_Q10), 8);
            pxq[i - delay] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            NSQ->sLTP_shp_Q14[NSQ->sLTP_shp_buf_idx - delay + i] = d->shape[last];
        }
        memcpy(NSQ->sLPC_Q14, &d->lpc[L], 32 * sizeof(i32));
        memcpy(NSQ->sAR2_Q14, d->ar2, sizeof(d->ar2));
        NSQ->sLF_AR_shp_Q14 = d->lf_ar;
    }
    NSQ->lagPrev = in->pitchL[nb_subfr - 1];
    memmove(NSQ->xq, &NSQ->xq[frame_length], ltp_mem * sizeof(i16));
    memmove(NSQ->sLTP_shp_Q14, &NSQ->sLTP_shp_Q14[frame_length], ltp_mem * sizeof(i32));
}

void orc_silk_nsq_del_dec_batch(const opusgpu_nsq_dd_in *in, opusgpu_nsq_state *st, opusgpu_nsq_dd_out *out, int n)
{
    for (int i = 0; i < n; i++) orc_silk_nsq_del_dec(&in[i], &st[i], &out[i]);
}

void orc_silk_burg_batch(const opusgpu_burg_in *in, opusgpu_burg_out *out, int n)
{
    for (int i = 0; i < n; i++) orc_silk_burg_modified(&in[i], &out[i]);
}

void orc_silk_nsq_batch(const opusgpu_nsq_in *in, opusgpu_nsq_state *st, opusgpu_nsq_out *out, int n)
{
    for (int i = 0; i < n; i++) orc_silk_nsq(&in[i], &st[i], &out[i]);
}
