// silk_pred_dev.h -- silk_find_pred_coefs_FIX (opus-fix/silk/fixed/find_pred_coefs_FIX.c:35-148), whole: gain weighting,
// the voiced branch (LTP analysis, LTP gain quantisation, LTP scaling control, LTP residual) or the unvoiced one
// (gain-scaled copy), then silk_find_LPC_FIX, silk_process_NLSFs and silk_residual_energy_FIX.
// One lane owns one frame. res_pitch / x are read where the record holds them; LPC_in_pre, which the Burg analyses and the
// residual filters read many times, is built in the caller's fast storage (`pre`).
#pragma once
#include "silk_ltp_dev.h"

namespace ca {

struct PredCoefsOut {                                   // what the call leaves in psEnc / psEncCtrl
    i16 PredCoef_Q12[2][SILK_MAX_LPC];
    i16 LTPCoef_Q14[4 * LTP_ORDER];
    i16 NLSF_Q15[SILK_MAX_LPC];                         // -> psEnc->sCmn.prev_NLSFq_Q15
    i32 ResNrg[4], ResNrgQ[4];
    i32 LTPredCodGain_Q7, LTP_scale_Q14, sum_log_gain_Q7;
    i8 NLSFIndices[SILK_MAX_LPC + 1];
    int NLSFInterpCoef_Q2, PERIndex, LTP_scaleIndex;
    i8 LTPIndex[4];
};

struct PredCoefsCfg {                                   // the psEnc / psEncCtrl fields the call reads
    i32 Gains_Q16[4];
    int pitchL[4];
    i16 prev_NLSFq_Q15[SILK_MAX_LPC];
    int nb_subfr, subfr_length, predictLPCOrder, ltp_mem_length, signalType, condCoding, first_frame_after_reset, useInterpolatedNLSFs,
        speech_activity_Q8, NLSF_MSVQ_Survivors, mu_LTP_Q9, LTPQuantLowComplexity, sum_log_gain_Q7, coding_quality_Q14, PacketLoss_perc,
        nFramesPerPacket;
};

// res_pitch: index 0 = res_pitch[0]; x: index 0 = the reference's x[0] (the frame), negative indices reach into x_buf.
// pre: storage of LPC_in_pre (nb * (subfr_length + order) samples, each read a handful of times, in order); e: the Burg recursion's
// edge accessor over it (silk_burg_dev.h), staged here once LPC_in_pre exists; tables / enc: the workgroup's NLSF codebook copies and
// derived tables (silk_nlsf_dev.h).
template <class XG, class PRE, class XE>
CA_DEV void silk_find_pred_coefs_dev(const PredCoefsCfg &c, XG res_pitch, XG x, PRE pre, XE e, PredCoefsOut &o, const NlsfTablesLds *tables = nullptr,
                                     const NlsfEncTables *enc = nullptr)
{
    const int order = c.predictLPCOrder, nb = c.nb_subfr, L = c.subfr_length;
    i32 invGains_Q16[4], local_gains[4], Wght_Q15[4];
    i32 min_gain_Q16 = 0x7FFFFFFF >> 6;
    for (int i = 0; i < nb; i++) min_gain_Q16 = imin(min_gain_Q16, c.Gains_Q16[i]);
    for (int i = 0; i < nb; i++) {
        invGains_Q16[i] = imax(s_div32_varq(min_gain_Q16, c.Gains_Q16[i], 16 - 2), 363);
        Wght_Q15[i] = s_smulwb(invGains_Q16[i], invGains_Q16[i]) >> 1;
        local_gains[i] = ((i32)1 << 16) / invGains_Q16[i];
    }
    o.sum_log_gain_Q7 = c.sum_log_gain_Q7;
    o.LTP_scale_Q14 = 0;
    o.LTP_scaleIndex = -1;                              // unvoiced: psEnc->sCmn.indices.LTP_scaleIndex / LTP_scale_Q14 are not written
    o.PERIndex = 0;
    for (int i = 0; i < 4; i++) o.LTPIndex[i] = 0;
    if (c.signalType == 2) {                            // TYPE_VOICED
        i32 WLTP[4 * LTP_ORDER * LTP_ORDER];
        int LTP_corrs_rshift[4], gain_Q7 = 0;
        silk_find_LTP_dev(o.LTPCoef_Q14, WLTP, &gain_Q7, res_pitch, c.pitchL, Wght_Q15, L, nb, c.ltp_mem_length, LTP_corrs_rshift);
        o.LTPredCodGain_Q7 = gain_Q7;
        silk_quant_LTP_gains_dev(o.LTPCoef_Q14, o.LTPIndex, &o.PERIndex, &o.sum_log_gain_Q7, WLTP, c.mu_LTP_Q9, c.LTPQuantLowComplexity, nb);
        // silk_LTP_scale_ctrl_FIX (LTP_scale_ctrl_FIX.c:35-53); SILK_FIX_CONST(0.1, 9) = 51
        if (c.condCoding == 0) {                        // CODE_INDEPENDENTLY
            const int round_loss = c.PacketLoss_perc + c.nFramesPerPacket;
            o.LTP_scaleIndex = s_limit(s_smulwb(s_smulbb(round_loss, o.LTPredCodGain_Q7), 51), 0, 2);
        } else {
            o.LTP_scaleIndex = 0;
        }
        o.LTP_scale_Q14 = SILK_LTPScales_table_Q14[o.LTP_scaleIndex];
        silk_LTP_analysis_filter_dev(pre, x + (-order), o.LTPCoef_Q14, c.pitchL, invGains_Q16, L, nb, order);
    } else {
        for (int i = 0; i < nb; i++) {                  // silk_scale_copy_vector16 per subframe, order samples prepended
            const int n = L + order;
            for (int k = 0; k < n; k++) pre[i * n + k] = (i16)s_smulwb(invGains_Q16[i], (i32)x[i * L - order + k]);
        }
        for (int i = 0; i < nb * LTP_ORDER; i++) o.LTPCoef_Q14[i] = 0;
        o.LTPredCodGain_Q7 = 0;
        o.sum_log_gain_Q7 = 0;
    }
    i32 minInvGain_Q30;
    if (c.first_frame_after_reset) {
        minInvGain_Q30 = 10737418;                      // SILK_FIX_CONST(1.0f / MAX_PREDICTION_POWER_GAIN_AFTER_RESET, 30)
    } else {
        // SILK_FIX_CONST(1.0 / 3, 16) = 21845; MAX_PREDICTION_POWER_GAIN = 10000; SILK_FIX_CONST(0.25 / 0.75, 18) = 65536 / 196608
        minInvGain_Q30 = s_log2lin(s_smlawb(16 << 7, o.LTPredCodGain_Q7, 21845));
        minInvGain_Q30 = s_div32_varq(minInvGain_Q30, s_smulww(10000, s_smlawb(65536, 196608, c.coding_quality_Q14)), 14);
    }
    e.stage(pre, L + order, nb);
    o.NLSFInterpCoef_Q2 = silk_find_LPC_dev(pre, e, minInvGain_Q30, L, nb, order, c.useInterpolatedNLSFs, c.first_frame_after_reset,
                                            c.prev_NLSFq_Q15, o.NLSF_Q15);
    silk_process_NLSFs_dev(o.PredCoef_Q12, o.NLSFIndices, o.NLSF_Q15, c.prev_NLSFq_Q15, c.speech_activity_Q8, nb, order,
                           c.useInterpolatedNLSFs, o.NLSFInterpCoef_Q2, c.NLSF_MSVQ_Survivors, c.signalType, tables, enc);
    silk_residual_energy_dev(o.ResNrg, o.ResNrgQ, pre, o.PredCoef_Q12, local_gains, L, nb, order);
}

template <class XG, class PRE>
CA_DEV void silk_find_pred_coefs_dev(const PredCoefsCfg &c, XG res_pitch, XG x, PRE pre, PredCoefsOut &o, const NlsfTablesLds *tables = nullptr)
{
    BurgEdgesOf<PRE> e;
    e.x = pre; e.L = c.subfr_length + c.predictLPCOrder;
    silk_find_pred_coefs_dev(c, res_pitch, x, pre, e, o, tables);
}

}  // namespace ca
