// silk_prefilter_dev.h -- silk_prefilter_FIX (opus-fix/silk/fixed/prefilter_FIX.c:102-184): the noise-shaping prefilter that turns
// the input frame into the quantiser's input xw_Q3 for silk_NSQ / silk_NSQ_del_dec (sixth slice).
//
//   silk_prefilter_FIX                       opus-fix/silk/fixed/prefilter_FIX.c:102-184
//   silk_warped_LPC_analysis_filter_FIX_c    opus-fix/silk/fixed/prefilter_FIX.c:58-100
//   silk_prefilt_FIX                         opus-fix/silk/fixed/prefilter_FIX.c:186-241
//
// One lane owns one frame and its silk_prefilter_state_FIX; the 512-entry harmonic-shaping ring buffer is reached through
// an accessor (LDS column on the device).
#pragma once
#include "silk_shape_dev.h"

namespace ca {

enum { LTP_BUF_LENGTH = 512, LTP_MASK = LTP_BUF_LENGTH - 1 };

struct PrefilterState {                                 // silk_prefilter_state_FIX minus sLTP_shp[] (structs_FIX.h:53-62)
    i32 sAR_shp[MAX_SHAPE_LPC_ORDER + 1];
    int sLTP_shp_buf_idx;
    i32 sLF_AR_shp_Q12, sLF_MA_shp_Q12, sHarmHP_Q2, rand_seed;
    int lagPrev;
};

struct PrefilterCtrl {                                  // the psEncCtrl / psEnc->sCmn fields read
    int pitchL[4], HarmShapeGain_Q14[4], HarmBoost_Q14[4], Tilt_Q14[4], GainsPre_Q14[4];
    i32 LF_shp_Q14[4];
    i16 AR1_Q13[4 * MAX_SHAPE_LPC_ORDER];
    int coding_quality_Q14, nb_subfr, subfr_length, signalType, warping_Q16, shapingLPCOrder;
};

CA_DEV i32 s_smulwt(i32 a, i32 b) { return s_smulwb(a, b >> 16); }

// x: the frame; xw: accessor for xw_Q3[] (i32 out); ltp: accessor for P->sLTP_shp[512]
template <class XG, class XW, class LTP>
CA_DEV void silk_prefilter_dev(PrefilterState &P, const PrefilterCtrl &c, XG x, XW xw_Q3, LTP ltp)
{
    const int L = c.subfr_length, order = c.shapingLPCOrder, lambda_Q16 = (i16)c.warping_Q16;
    int lag = P.lagPrev;
    for (int k = 0; k < c.nb_subfr; k++) {
        if (c.signalType == 2) lag = c.pitchL[k];
        const int HarmShapeGain_Q12 = s_smulwb((i32)c.HarmShapeGain_Q14[k], 16384 - c.HarmBoost_Q14[k]);
        i32 HarmShapeFIRPacked_Q12 = HarmShapeGain_Q12 >> 2;
        HarmShapeFIRPacked_Q12 |= shl32((i32)(HarmShapeGain_Q12 >> 1), 16);
        const int Tilt_Q14 = c.Tilt_Q14[k];
        const i32 LF_shp_Q14 = c.LF_shp_Q14[k];
        i32 coef_Q13[MAX_SHAPE_LPC_ORDER];                                                  // this subframe's filter, in registers
#pragma unroll
        for (int i = 0; i < MAX_SHAPE_LPC_ORDER; i++) coef_Q13[i] = c.AR1_Q13[k * MAX_SHAPE_LPC_ORDER + i];
        i32 coef_last = 0;                                                                  // coef_Q13[order - 1]
#pragma unroll
        for (int i = 1; i < MAX_SHAPE_LPC_ORDER; i += 2)
            if (i == order - 1) coef_last = coef_Q13[i];
        const i32 B0 = (i16)s_rshift_round(c.GainsPre_Q14[k], 4);
        i32 tmp_32 = s_addw(3355443, s_smulbb(c.HarmBoost_Q14[k], HarmShapeGain_Q12));        // SILK_FIX_CONST(INPUT_TILT, 26)
        tmp_32 = s_addw(tmp_32, s_smulbb(c.coding_quality_Q14, 410));                           // SILK_FIX_CONST(HIGH_RATE_INPUT_TILT, 12)
        tmp_32 = s_smulwb(tmp_32, -c.GainsPre_Q14[k]);
        tmp_32 = s_rshift_round(tmp_32, 14);
        const i32 B1 = (i16)(tmp_32 > 32767 ? 32767 : (tmp_32 < -32768 ? -32768 : tmp_32));
        int LTP_shp_buf_idx = P.sLTP_shp_buf_idx;
        i32 sLF_AR_shp_Q12 = P.sLF_AR_shp_Q12, sLF_MA_shp_Q12 = P.sLF_MA_shp_Q12;
        i32 prev_res_Q2 = P.sHarmHP_Q2;
        // one sample: silk_warped_LPC_analysis_filter_FIX_c (:58-100), the harmonic high-pass, silk_prefilt_FIX; returns xw_Q3
        auto sample = [&](const i32 in) -> i32 {
            i32 tmp2 = s_smlawb(P.sAR_shp[0], P.sAR_shp[1], lambda_Q16);
            P.sAR_shp[0] = shl32(in, 14);
            i32 tmp1 = s_smlawb(P.sAR_shp[1], s_subw(P.sAR_shp[2], tmp2), lambda_Q16);
            P.sAR_shp[1] = tmp2;
            i32 acc_Q11 = order >> 1;
            acc_Q11 = s_smlawb(acc_Q11, tmp2, coef_Q13[0]);
#pragma unroll
            for (int i = 2; i < MAX_SHAPE_LPC_ORDER; i += 2) {                              // unrolled: sAR_shp[] stays in registers
                if (i < order) {
                    tmp2 = s_smlawb(P.sAR_shp[i], s_subw(P.sAR_shp[i + 1], tmp1), lambda_Q16);
                    P.sAR_shp[i] = tmp1;
                    acc_Q11 = s_smlawb(acc_Q11, tmp1, coef_Q13[i - 1]);
                    tmp1 = s_smlawb(P.sAR_shp[i + 1], s_subw(P.sAR_shp[i + 2], tmp2), lambda_Q16);
                    P.sAR_shp[i + 1] = tmp2;
                    acc_Q11 = s_smlawb(acc_Q11, tmp2, coef_Q13[i]);
                }
            }
#pragma unroll
            for (int i = 2; i <= MAX_SHAPE_LPC_ORDER; i += 2)
                if (i == order) P.sAR_shp[i] = tmp1;
            acc_Q11 = s_smlawb(acc_Q11, tmp1, coef_last);
            const i32 st_res_Q2 = s_subw(shl32(in, 2), s_rshift_round(acc_Q11, 9));
            // harmonic high-pass (:158-164)
            const i32 x_filt_Q12 = s_addw(s_mulw(st_res_Q2, B0), s_mulw(prev_res_Q2, B1));
            prev_res_Q2 = st_res_Q2;
            // silk_prefilt_FIX (:186-241), one sample
            i32 n_LTP_Q12 = 0;
            if (lag > 0) {
                const int idx = lag + LTP_shp_buf_idx;
                n_LTP_Q12 = s_smulbb((i32)ltp[(idx - 2) & LTP_MASK], HarmShapeFIRPacked_Q12);
                n_LTP_Q12 = s_addw(n_LTP_Q12, __mul24((i32)ltp[(idx - 1) & LTP_MASK], HarmShapeFIRPacked_Q12 >> 16));
                n_LTP_Q12 = s_addw(n_LTP_Q12, s_smulbb((i32)ltp[idx & LTP_MASK], HarmShapeFIRPacked_Q12));
            }
            const i32 n_Tilt_Q10 = s_smulwb(sLF_AR_shp_Q12, Tilt_Q14);
            const i32 n_LF_Q10 = s_smlawb(s_smulwt(sLF_AR_shp_Q12, LF_shp_Q14), sLF_MA_shp_Q12, LF_shp_Q14);
            sLF_AR_shp_Q12 = s_subw(x_filt_Q12, shl32(n_Tilt_Q10, 2));
            sLF_MA_shp_Q12 = s_subw(sLF_AR_shp_Q12, shl32(n_LF_Q10, 2));
            LTP_shp_buf_idx = (LTP_shp_buf_idx - 1) & LTP_MASK;
            const i32 v = s_rshift_round(sLF_MA_shp_Q12, 12);
            ltp[LTP_shp_buf_idx] = (i16)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
            return s_rshift_round(s_subw(sLF_MA_shp_Q12, n_LTP_Q12), 9);
        };
        // Four samples per group: one 8-byte load of the input, issued AHEAD of the previous group's 16-byte store (loads queue
        // behind stores: a load and a store per sample would be an exposed memory round trip per sample)
        int n = 0;
        struct In4 { i16 v[4]; } in4;
        struct Out4 { i32 v[4]; } out4;
        if (L >= 4) __builtin_memcpy(&in4, &x[k * L], sizeof(in4));
        for (; n + 4 <= L; n += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) out4.v[u] = sample((i32)in4.v[u]);
            if (n + 8 <= L) __builtin_memcpy(&in4, &x[k * L + n + 4], sizeof(in4));
            __builtin_memcpy(&xw_Q3[k * L + n], &out4, sizeof(out4));
        }
        for (; n < L; n++) xw_Q3[k * L + n] = sample((i32)x[k * L + n]);
        P.sHarmHP_Q2 = prev_res_Q2;
        P.sLF_AR_shp_Q12 = sLF_AR_shp_Q12;
        P.sLF_MA_shp_Q12 = sLF_MA_shp_Q12;
        P.sLTP_shp_buf_idx = LTP_shp_buf_idx;
    }
    P.lagPrev = c.pitchL[c.nb_subfr - 1];
}

}  // namespace ca
