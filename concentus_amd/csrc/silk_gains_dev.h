// silk_gains_dev.h -- silk_process_gains_FIX (opus-fix/silk/fixed/process_gains_FIX.c:37-125): the step between
// silk_find_pred_coefs_FIX and the noise-shaping quantiser in silk_encode_frame_FIX (fourth slice).
//
//   silk_process_gains_FIX     opus-fix/silk/fixed/process_gains_FIX.c:37-125
//   silk_gains_quant           opus-fix/silk/gain_quant.c:41-103
//   silk_sigm_Q15              opus-fix/silk/sigm_Q15.c:49-75
#pragma once
#include "silk_ltp_dev.h"

namespace ca {

// round(1024 * diff(sigmoid(0..5, 1)))), round(32767 * sigmoid(0..5)), round(32767 * sigmoid(-(0..5)))  (sigm_Q15.c:36-47;
// tests/test_tables.py re-derives them and compares with the compiled reference)
CA_DEVICE_CONST i32 SILK_sigm_LUT_slope_Q10[6] = {237, 153, 73, 30, 12, 7};
CA_DEVICE_CONST i32 SILK_sigm_LUT_pos_Q15[6] = {16384, 23955, 28861, 31213, 32178, 32548};
CA_DEVICE_CONST i32 SILK_sigm_LUT_neg_Q15[6] = {16384, 8812, 3906, 1554, 589, 219};
// silk_Quantization_Offsets_Q10[signalType >> 1][quantOffsetType] (tables_other.c:95-97; define.h:125-128)
CA_DEVICE_CONST i16 SILK_Quantization_Offsets_Q10[4] = {100, 240, 32, 100};

CA_DEV int silk_sigm_Q15_dev(int in_Q5)                                                      // sigm_Q15.c:49-75
{
    if (in_Q5 < 0) {
        in_Q5 = -in_Q5;
        if (in_Q5 >= 6 * 32) return 0;
        const int ind = in_Q5 >> 5;
        return SILK_sigm_LUT_neg_Q15[ind] - s_smulbb(SILK_sigm_LUT_slope_Q10[ind], in_Q5 & 0x1F);
    }
    if (in_Q5 >= 6 * 32) return 32767;
    const int ind = in_Q5 >> 5;
    return SILK_sigm_LUT_pos_Q15[ind] + s_smulbb(SILK_sigm_LUT_slope_Q10[ind], in_Q5 & 0x1F);
}

enum { GQ_OFFSET = 2090, GQ_SCALE_Q16 = 2251, GQ_INV_SCALE_Q16 = 1907825, GQ_N_LEVELS = 64, GQ_MAX_DELTA = 36, GQ_MIN_DELTA = -4 };

CA_DEV void silk_gains_quant_dev(i8 *ind, i32 *gain_Q16, int *prev_ind_io, int conditional, int nb_subfr)   // gain_quant.c:41-103
{
    int prev_ind = *prev_ind_io;
    for (int k = 0; k < nb_subfr; k++) {
        int v = (i8)s_smulwb(GQ_SCALE_Q16, s_lin2log(gain_Q16[k]) - GQ_OFFSET);
        if (v < prev_ind) v = (i8)(v + 1);
        v = s_limit(v, 0, GQ_N_LEVELS - 1);
        if (k == 0 && conditional == 0) {
            v = s_limit(v, prev_ind + GQ_MIN_DELTA, GQ_N_LEVELS - 1);
            prev_ind = (i8)v;
        } else {
            v = (i8)(v - prev_ind);
            const int thr = 2 * GQ_MAX_DELTA - GQ_N_LEVELS + prev_ind;
            if (v > thr) v = (i8)(thr + ((v - thr + 1) >> 1));
            v = s_limit(v, GQ_MIN_DELTA, GQ_MAX_DELTA);
            if (v > thr) prev_ind = (i8)(prev_ind + shl32(v, 1) - thr);
            else prev_ind = (i8)(prev_ind + v);
            v = (i8)(v - GQ_MIN_DELTA);
        }
        ind[k] = (i8)v;
        gain_Q16[k] = s_log2lin(imin(s_smulwb(GQ_INV_SCALE_Q16, prev_ind) + GQ_OFFSET, 3967));
    }
    *prev_ind_io = prev_ind;
}

struct ProcessGainsIO {
    i32 Gains_Q16[4];            // I/O (quantised out)
    i32 GainsUnq_Q16[4];         // O
    i32 ResNrg[4];               // I
    int ResNrgQ[4];              // I
    i8 GainsIndices[4];          // O
    int LastGainIndex;           // I/O  (psEnc->sShape.LastGainIndex)
    int lastGainIndexPrev;       // O
    int quantOffsetType;         // I/O  (written for voiced frames)
    int Lambda_Q10;              // O
};

// process_gains_FIX.c:37-125
CA_DEV void silk_process_gains_dev(ProcessGainsIO &g, int signalType, int nb_subfr, int subfr_length, int LTPredCodGain_Q7, int SNR_dB_Q7,
                                   int condCoding, int input_tilt_Q15, int nStatesDelayedDecision, int speech_activity_Q8,
                                   int input_quality_Q14, int coding_quality_Q14)
{
    if (signalType == 2) {
        const i32 s_Q16 = -silk_sigm_Q15_dev(s_rshift_round(LTPredCodGain_Q7 - 1536, 4));     // SILK_FIX_CONST(12.0, 7)
        for (int k = 0; k < nb_subfr; k++) g.Gains_Q16[k] = s_smlawb(g.Gains_Q16[k], g.Gains_Q16[k], s_Q16);
    }
    // SILK_FIX_CONST(21 + 16 / 0.33, 7) = 8894, SILK_FIX_CONST(0.33, 16) = 21627
    const i32 InvMaxSqrVal_Q16 = s_log2lin(s_smulwb(8894 - SNR_dB_Q7, 21627)) / subfr_length;
    for (int k = 0; k < nb_subfr; k++) {
        const i32 ResNrg = g.ResNrg[k];
        i32 ResNrgPart = s_smulww(ResNrg, InvMaxSqrVal_Q16);
        if (g.ResNrgQ[k] > 0) {
            ResNrgPart = s_rshift_round(ResNrgPart, g.ResNrgQ[k]);
        } else if (ResNrgPart >= (0x7FFFFFFF >> (-g.ResNrgQ[k]))) {
            ResNrgPart = 0x7FFFFFFF;
        } else {
            ResNrgPart = shl32(ResNrgPart, -g.ResNrgQ[k]);
        }
        i32 gain = g.Gains_Q16[k];
        i32 gain_squared = s_add_sat32(ResNrgPart, s_smmul(gain, gain));
        if (gain_squared < 32767) {
            gain_squared = s_smlaww(shl32(ResNrgPart, 16), gain, gain);
            gain = s_sqrt_approx(gain_squared);
            gain = imin(gain, 0x7FFFFFFF >> 8);
            g.Gains_Q16[k] = s_lshift_sat32(gain, 8);
        } else {
            gain = s_sqrt_approx(gain_squared);
            gain = imin(gain, 0x7FFFFFFF >> 16);
            g.Gains_Q16[k] = s_lshift_sat32(gain, 16);
        }
    }
    for (int k = 0; k < nb_subfr; k++) g.GainsUnq_Q16[k] = g.Gains_Q16[k];
    g.lastGainIndexPrev = g.LastGainIndex;
    silk_gains_quant_dev(g.GainsIndices, g.Gains_Q16, &g.LastGainIndex, condCoding == 2 /* CODE_CONDITIONALLY */, nb_subfr);
    if (signalType == 2) g.quantOffsetType = (LTPredCodGain_Q7 + (input_tilt_Q15 >> 8) > 128) ? 0 : 1;   // SILK_FIX_CONST(1.0, 7)
    const i32 quant_offset_Q10 = SILK_Quantization_Offsets_Q10[(signalType >> 1) * 2 + g.quantOffsetType];
    // LAMBDA_OFFSET Q10 1229, LAMBDA_DELAYED_DECISIONS Q10 -50, LAMBDA_SPEECH_ACT Q18 -52428, LAMBDA_INPUT_QUALITY Q12 -409,
    // LAMBDA_CODING_QUALITY Q12 -818, LAMBDA_QUANT_OFFSET Q16 52429 (single-precision constants through SILK_FIX_CONST)
    g.Lambda_Q10 = 1229 + s_smulbb(-50, nStatesDelayedDecision) + s_smulwb(-52428, speech_activity_Q8) + s_smulwb(-409, input_quality_Q14)
                   + s_smulwb(-818, coding_quality_Q14) + s_smulwb(52429, quant_offset_Q10);
}

}  // namespace ca
