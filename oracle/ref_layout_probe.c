/* oracle/ref_layout_probe.c -- TEST INFRASTRUCTURE ONLY.
 * Prints, as JSON, the x86-64 offsets of the reference's struct fields that the per-call SILK hooks read through the
 * reference's own pointer types (silk_encoder_state / SideInfoIndices, opus-fix/silk/structs.h), so that
 * tests/test_hooks_layout.py can pin the constants hard-coded in include/opusgpu_hooks.h. Compiled against the
 * reference's headers where they lie (oracle/Makefile -> oracle/_ref/layout_probe). */
#include <stddef.h>
#include <stdio.h>
#include "main_FIX.h"

#define F(T, f) printf("  \"%s.%s\": %zu,\n", #T, #f, offsetof(T, f))

int main(void)
{
    printf("{\n");
    F(silk_encoder_state, nb_subfr); F(silk_encoder_state, subfr_length); F(silk_encoder_state, frame_length);
    F(silk_encoder_state, ltp_mem_length); F(silk_encoder_state, predictLPCOrder); F(silk_encoder_state, shapingLPCOrder);
    F(silk_encoder_state, nStatesDelayedDecision); F(silk_encoder_state, warping_Q16); F(silk_encoder_state, arch);
    F(silk_encoder_state, useInterpolatedNLSFs); F(silk_encoder_state, first_frame_after_reset); F(silk_encoder_state, prev_NLSFq_Q15);
    F(silk_encoder_state, indices); F(silk_encoder_state, speech_activity_Q8); F(silk_encoder_state, NLSF_MSVQ_Survivors);
    F(silk_encoder_state, psNLSF_CB);
    F(silk_encoder_state, mu_LTP_Q9); F(silk_encoder_state, LTPQuantLowComplexity); F(silk_encoder_state, sum_log_gain_Q7);
    F(silk_encoder_state, PacketLoss_perc); F(silk_encoder_state, nFramesPerPacket);
    F(silk_encoder_state_FIX, sCmn);
    F(silk_encoder_control_FIX, Gains_Q16); F(silk_encoder_control_FIX, PredCoef_Q12); F(silk_encoder_control_FIX, LTPCoef_Q14);
    F(silk_encoder_control_FIX, LTP_scale_Q14); F(silk_encoder_control_FIX, pitchL); F(silk_encoder_control_FIX, LTPredCodGain_Q7);
    F(silk_encoder_control_FIX, ResNrg); F(silk_encoder_control_FIX, ResNrgQ); F(silk_encoder_control_FIX, coding_quality_Q14);
    F(silk_encoder_state, fs_kHz);
    F(silk_encoder_state, la_pitch);
    F(silk_encoder_state, la_shape);
    F(silk_encoder_state, shapeWinLength);
    F(silk_encoder_state, pitch_LPC_win_length);
    F(silk_encoder_state, pitchEstimationLPCOrder);
    F(silk_encoder_state, pitchEstimationComplexity);
    F(silk_encoder_state, pitchEstimationThreshold_Q16);
    F(silk_encoder_state, SNR_dB_Q7);
    F(silk_encoder_state, useCBR);
    F(silk_encoder_state, input_quality_bands_Q15);
    F(silk_encoder_state, input_tilt_Q15);
    F(silk_encoder_state, prevSignalType);
    F(silk_encoder_state, prevLag);
    F(SideInfoIndices, GainsIndices);
    F(SideInfoIndices, lagIndex);
    F(SideInfoIndices, contourIndex);
    F(silk_encoder_state_FIX, sShape);
    F(silk_encoder_state_FIX, sPrefilt);
    F(silk_encoder_state_FIX, LTPCorr_Q15);
    F(silk_shape_state_FIX, LastGainIndex);
    F(silk_shape_state_FIX, HarmBoost_smth_Q16);
    F(silk_shape_state_FIX, HarmShapeGain_smth_Q16);
    F(silk_shape_state_FIX, Tilt_smth_Q16);
    F(silk_encoder_control_FIX, AR1_Q13);
    F(silk_encoder_control_FIX, AR2_Q13);
    F(silk_encoder_control_FIX, LF_shp_Q14);
    F(silk_encoder_control_FIX, GainsPre_Q14);
    F(silk_encoder_control_FIX, HarmBoost_Q14);
    F(silk_encoder_control_FIX, Tilt_Q14);
    F(silk_encoder_control_FIX, HarmShapeGain_Q14);
    F(silk_encoder_control_FIX, Lambda_Q10);
    F(silk_encoder_control_FIX, input_quality_Q14);
    F(silk_encoder_control_FIX, sparseness_Q8);
    F(silk_encoder_control_FIX, predGain_Q16);
    F(silk_encoder_control_FIX, GainsUnq_Q16);
    F(silk_encoder_control_FIX, lastGainIndexPrev);
    F(silk_encoder_state, sVAD);
    F(SideInfoIndices, signalType); F(SideInfoIndices, quantOffsetType); F(SideInfoIndices, NLSFInterpCoef_Q2);
    F(SideInfoIndices, Seed); F(SideInfoIndices, NLSFIndices);
    F(SideInfoIndices, LTPIndex); F(SideInfoIndices, PERIndex); F(SideInfoIndices, LTP_scaleIndex);
    F(silk_encoder_state, inputBuf); F(silk_encoder_state, frameCounter); F(silk_encoder_state, prefillFlag); F(silk_encoder_state, sLP);
    F(silk_LP_state, mode); F(silk_encoder_state, LBRR_enabled); F(silk_encoder_state, pulses); F(silk_encoder_state, sNSQ);
    F(silk_encoder_state, nFramesEncoded); F(silk_encoder_state, ec_prevLagIndex); F(silk_encoder_state, ec_prevSignalType);
    F(silk_encoder_state_FIX, x_buf);
    printf("  \"sizeof.silk_encoder_state\": %zu,\n  \"sizeof.SideInfoIndices\": %zu,\n  \"sizeof.silk_nsq_state\": %zu,\n  \"sizeof.silk_encoder_control_FIX\": %zu,\n  \"sizeof.silk_prefilter_state_FIX\": %zu,\n  \"sizeof.silk_encoder_state_FIX\": %zu,\n  \"sizeof.silk_VAD_state\": %zu\n}\n",
           sizeof(silk_encoder_state), sizeof(SideInfoIndices), sizeof(silk_nsq_state), sizeof(silk_encoder_control_FIX),
           sizeof(silk_prefilter_state_FIX), sizeof(silk_encoder_state_FIX), sizeof(silk_VAD_state));
    return 0;
}
