// silk_vad_dev.h -- silk_VAD_GetSA_Q8_c (opus-fix/silk/VAD.c:82-312): the voice-activity detector that opens silk_encode_frame_FIX
// (silk/fixed/encode_frame_FIX.c:58) and produces speech_activity_Q8, input_tilt_Q15 and input_quality_bands_Q15[] -- inputs of the
// analysis chain (ninth slice).
//
//   silk_VAD_GetSA_Q8_c        opus-fix/silk/VAD.c:82-312
//   silk_VAD_GetNoiseLevels    opus-fix/silk/VAD.c:317-390
//   silk_ana_filt_bank_1       opus-fix/silk/ana_filt_bank_1.c:39-74
//
// One lane owns one frame and its silk_VAD_state; X: scratch of 5/4 frame_length samples in the caller's storage.
#pragma once
#include "silk_gains_dev.h"

namespace ca {

enum { VAD_N_BANDS = 4, VAD_INTERNAL_SUBFRAMES_LOG2 = 2, VAD_INTERNAL_SUBFRAMES = 4, VAD_NOISE_LEVEL_SMOOTH_COEF_Q16 = 1024,
       VAD_NEGATIVE_OFFSET_Q5 = 128, VAD_SNR_FACTOR_Q16 = 45000, VAD_SNR_SMOOTH_COEF_Q18 = 4096 };

struct VadState {               // silk_VAD_state (opus-fix/silk/structs.h:60-73)
    i32 AnaState[2], AnaState1[2], AnaState2[2], XnrgSubfr[VAD_N_BANDS], NrgRatioSmth_Q8[VAD_N_BANDS];
    i16 HPstate;
    i32 NL[VAD_N_BANDS], inv_NL[VAD_N_BANDS], NoiseLevelBias[VAD_N_BANDS], counter;
};

// ana_filt_bank_1.c:39-74; in / outL may be the same buffer (outL[k] is written after in[2k], in[2k+1] were read)
template <class IN, class XS>
CA_DEV void silk_ana_filt_bank_1_dev(IN in, i32 *S, XS outL, XS outH, int N)
{
    const int N2 = N >> 1;
    for (int k = 0; k < N2; k++) {
        i32 in32 = shl32((i32)in[2 * k], 10);
        i32 Y = s_subw(in32, S[0]);
        i32 X = s_smlawb(Y, Y, -24290);                                                     // A_fb1_21
        const i32 out_1 = s_addw(S[0], X);
        S[0] = s_addw(in32, X);
        in32 = shl32((i32)in[2 * k + 1], 10);
        Y = s_subw(in32, S[1]);
        X = s_smulwb(Y, 5394 << 1);                                                         // A_fb1_20
        const i32 out_2 = s_addw(S[1], X);
        S[1] = s_addw(in32, X);
        const i32 lo = s_rshift_round(s_addw(out_2, out_1), 11), hi = s_rshift_round(s_subw(out_2, out_1), 11);
        outL[k] = (i16)(lo > 32767 ? 32767 : (lo < -32768 ? -32768 : lo));
        outH[k] = (i16)(hi > 32767 ? 32767 : (hi < -32768 ? -32768 : hi));
    }
}

CA_DEV i32 s_add_pos_sat32(i32 a, i32 b) { const i32 s = s_addw(a, b); return (s & 0x80000000) ? 0x7FFFFFFF : s; }

CA_DEV void silk_VAD_GetNoiseLevels_dev(const i32 *pX, VadState &V)                         // VAD.c:317-390
{
    const int min_coef = V.counter < 1000 ? 32767 / ((V.counter >> 4) + 1) : 0;
    for (int k = 0; k < VAD_N_BANDS; k++) {
        i32 nl = V.NL[k];
        const i32 nrg = s_add_pos_sat32(pX[k], V.NoiseLevelBias[k]);
        const i32 inv_nrg = 0x7FFFFFFF / nrg;
        int coef;
        if (nrg > shl32(nl, 3)) coef = VAD_NOISE_LEVEL_SMOOTH_COEF_Q16 >> 3;
        else if (nrg < nl) coef = VAD_NOISE_LEVEL_SMOOTH_COEF_Q16;
        else coef = s_smulwb(s_smulww(inv_nrg, nl), VAD_NOISE_LEVEL_SMOOTH_COEF_Q16 << 1);
        coef = imax(coef, min_coef);
        V.inv_NL[k] = s_smlawb(V.inv_NL[k], inv_nrg - V.inv_NL[k], coef);
        nl = 0x7FFFFFFF / V.inv_NL[k];
        nl = imin(nl, 0x00FFFFFF);
        V.NL[k] = nl;
    }
    V.counter++;
}

struct VadOut { int speech_activity_Q8, input_tilt_Q15, input_quality_bands_Q15[VAD_N_BANDS]; };

// VAD.c:82-312
template <class IN, class XS>
CA_DEV void silk_VAD_GetSA_Q8_dev(VadState &V, VadOut &o, IN pIn, XS X, int frame_length, int fs_kHz)
{
    const i32 tiltWeights[VAD_N_BANDS] = {30000, 6000, -12000, -12000};
    const int dl1 = frame_length >> 1, dl2 = frame_length >> 2, dl = frame_length >> 3;
    int X_offset[VAD_N_BANDS];
    X_offset[0] = 0;
    X_offset[1] = dl + dl2;
    X_offset[2] = X_offset[1] + dl;
    X_offset[3] = X_offset[2] + dl2;
    silk_ana_filt_bank_1_dev(pIn, V.AnaState, X, X + X_offset[3], frame_length);
    silk_ana_filt_bank_1_dev(X, V.AnaState1, X, X + X_offset[2], dl1);
    silk_ana_filt_bank_1_dev(X, V.AnaState2, X, X + X_offset[1], dl2);
    // HP filter on the lowest band (:150-164)
    X[dl - 1] = (i16)(X[dl - 1] >> 1);
    const i16 HPstateTmp = X[dl - 1];
    for (int i = dl - 1; i > 0; i--) {
        X[i - 1] = (i16)(X[i - 1] >> 1);
        X[i] = (i16)(X[i] - X[i - 1]);
    }
    X[0] = (i16)(X[0] - V.HPstate);
    V.HPstate = HPstateTmp;
    // energy per band (:166-206)
    i32 Xnrg[VAD_N_BANDS], NrgToNoiseRatio_Q8[VAD_N_BANDS];
    for (int b = 0; b < VAD_N_BANDS; b++) {
        const int dfl = frame_length >> imin(VAD_N_BANDS - b, VAD_N_BANDS - 1);
        const int dec_subframe_length = dfl >> VAD_INTERNAL_SUBFRAMES_LOG2;
        int dec_subframe_offset = 0;
        i32 sumSquared = 0;
        Xnrg[b] = V.XnrgSubfr[b];
        for (int s = 0; s < VAD_INTERNAL_SUBFRAMES; s++) {
            sumSquared = 0;
            for (int i = 0; i < dec_subframe_length; i++) {
                const i32 x_tmp = (i32)X[X_offset[b] + i + dec_subframe_offset] >> 3;
                sumSquared = s_addw(sumSquared, s_smulbb(x_tmp, x_tmp));
            }
            if (s < VAD_INTERNAL_SUBFRAMES - 1) Xnrg[b] = s_add_pos_sat32(Xnrg[b], sumSquared);
            else Xnrg[b] = s_add_pos_sat32(Xnrg[b], sumSquared >> 1);
            dec_subframe_offset += dec_subframe_length;
        }
        V.XnrgSubfr[b] = sumSquared;
    }
    silk_VAD_GetNoiseLevels_dev(Xnrg, V);
    // SNR per band, tilt (:213-247)
    i32 sumSquared = 0, input_tilt = 0;
    for (int b = 0; b < VAD_N_BANDS; b++) {
        i32 speech_nrg = Xnrg[b] - V.NL[b];
        if (speech_nrg > 0) {
            if ((Xnrg[b] & 0xFF800000) == 0) NrgToNoiseRatio_Q8[b] = shl32(Xnrg[b], 8) / (V.NL[b] + 1);
            else NrgToNoiseRatio_Q8[b] = Xnrg[b] / ((V.NL[b] >> 8) + 1);
            i32 SNR_Q7 = s_lin2log(NrgToNoiseRatio_Q8[b]) - 8 * 128;
            sumSquared = s_addw(sumSquared, s_smulbb(SNR_Q7, SNR_Q7));
            if (speech_nrg < ((i32)1 << 20)) SNR_Q7 = s_smulwb(shl32(s_sqrt_approx(speech_nrg), 6), SNR_Q7);
            input_tilt = s_smlawb(input_tilt, tiltWeights[b], SNR_Q7);
        } else {
            NrgToNoiseRatio_Q8[b] = 256;
        }
    }
    sumSquared = sumSquared / VAD_N_BANDS;
    const int pSNR_dB_Q7 = (i16)(3 * s_sqrt_approx(sumSquared));
    int SA_Q15 = silk_sigm_Q15_dev(s_smulwb(VAD_SNR_FACTOR_Q16, pSNR_dB_Q7) - VAD_NEGATIVE_OFFSET_Q5);
    o.input_tilt_Q15 = shl32(silk_sigm_Q15_dev(input_tilt) - 16384, 1);
    // power scaling (:262-284)
    i32 speech_nrg = 0;
    for (int b = 0; b < VAD_N_BANDS; b++) speech_nrg += (b + 1) * ((Xnrg[b] - V.NL[b]) >> 4);
    if (speech_nrg <= 0) {
        SA_Q15 >>= 1;
    } else if (speech_nrg < 32768) {
        speech_nrg = s_lshift_sat32(speech_nrg, frame_length == 10 * fs_kHz ? 16 : 15);
        speech_nrg = s_sqrt_approx(speech_nrg);
        SA_Q15 = s_smulwb(32768 + speech_nrg, SA_Q15);
    }
    o.speech_activity_Q8 = imin(SA_Q15 >> 7, 255);
    // smoothed energy-to-noise ratios, input quality (:289-307)
    i32 smooth_coef_Q16 = s_smulwb(VAD_SNR_SMOOTH_COEF_Q18, s_smulwb((i32)SA_Q15, SA_Q15));
    if (frame_length == 10 * fs_kHz) smooth_coef_Q16 >>= 1;
    for (int b = 0; b < VAD_N_BANDS; b++) {
        V.NrgRatioSmth_Q8[b] = s_smlawb(V.NrgRatioSmth_Q8[b], NrgToNoiseRatio_Q8[b] - V.NrgRatioSmth_Q8[b], smooth_coef_Q16);
        const i32 SNR_Q7 = 3 * (s_lin2log(V.NrgRatioSmth_Q8[b]) - 8 * 128);
        o.input_quality_bands_Q15[b] = silk_sigm_Q15_dev((SNR_Q7 - 16 * 128) >> 4);
    }
}

}  // namespace ca
