"""CPU tier: the struct offsets hard-coded in include/opusgpu_hooks.h (the SILK per-call hooks read a few fields through
the reference's own silk_encoder_state / SideInfoIndices pointers) against the reference's headers: oracle/Makefile
compiles oracle/ref_layout_probe.c against opus-fix/silk/structs.h and writes oracle/_ref/layout.json."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYOUT = os.path.join(ROOT, "oracle", "_ref", "layout.json")

PAIRS = {
    "OPUSGPU_REF_OFF_INPUT_BUF": "silk_encoder_state.inputBuf",
    "OPUSGPU_REF_OFF_FRAME_COUNTER": "silk_encoder_state.frameCounter",
    "OPUSGPU_REF_OFF_PREFILL_FLAG": "silk_encoder_state.prefillFlag",
    "OPUSGPU_REF_OFF_SLP": "silk_encoder_state.sLP",
    "OPUSGPU_REF_OFF_LP_MODE": "silk_LP_state.mode",
    "OPUSGPU_REF_OFF_LBRR_ENABLED": "silk_encoder_state.LBRR_enabled",
    "OPUSGPU_REF_OFF_PULSES": "silk_encoder_state.pulses",
    "OPUSGPU_REF_OFF_SNSQ": "silk_encoder_state.sNSQ",
    "OPUSGPU_REF_OFF_N_FRAMES_ENCODED": "silk_encoder_state.nFramesEncoded",
    "OPUSGPU_REF_OFF_EC_PREV_LAG_INDEX": "silk_encoder_state.ec_prevLagIndex",
    "OPUSGPU_REF_OFF_EC_PREV_SIGNAL_TYPE": "silk_encoder_state.ec_prevSignalType",
    "OPUSGPU_REF_OFF_FIX_X_BUF": "silk_encoder_state_FIX.x_buf",
    "OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE": "sizeof.silk_encoder_state",
    "OPUSGPU_REF_OFF_NB_SUBFR": "silk_encoder_state.nb_subfr",
    "OPUSGPU_REF_OFF_FRAME_LENGTH": "silk_encoder_state.frame_length",
    "OPUSGPU_REF_OFF_SUBFR_LENGTH": "silk_encoder_state.subfr_length",
    "OPUSGPU_REF_OFF_LTP_MEM_LENGTH": "silk_encoder_state.ltp_mem_length",
    "OPUSGPU_REF_OFF_N_STATES_DEL_DEC": "silk_encoder_state.nStatesDelayedDecision",
    "OPUSGPU_REF_OFF_SHAPING_LPC_ORDER": "silk_encoder_state.shapingLPCOrder",
    "OPUSGPU_REF_OFF_PREDICT_LPC_ORDER": "silk_encoder_state.predictLPCOrder",
    "OPUSGPU_REF_OFF_WARPING_Q16": "silk_encoder_state.warping_Q16",
    "OPUSGPU_REF_OFF_PREV_NLSFQ_Q15": "silk_encoder_state.prev_NLSFq_Q15",
    "OPUSGPU_REF_OFF_USE_INTERPOLATED_NLSFS": "silk_encoder_state.useInterpolatedNLSFs",
    "OPUSGPU_REF_OFF_FIRST_FRAME_AFTER_RESET": "silk_encoder_state.first_frame_after_reset",
    "OPUSGPU_REF_OFF_INDICES": "silk_encoder_state.indices",
    "OPUSGPU_REF_SIZEOF_SIDE_INFO_INDICES": "sizeof.SideInfoIndices",
    "OPUSGPU_REF_OFF_SIGNAL_TYPE": "SideInfoIndices.signalType",
    "OPUSGPU_REF_OFF_QUANT_OFFSET_TYPE": "SideInfoIndices.quantOffsetType",
    "OPUSGPU_REF_OFF_NLSF_INTERP_COEF_Q2": "SideInfoIndices.NLSFInterpCoef_Q2",
    "OPUSGPU_REF_OFF_SEED": "SideInfoIndices.Seed",
    "OPUSGPU_REF_OFF_NLSF_INDICES": "SideInfoIndices.NLSFIndices",
    "OPUSGPU_REF_OFF_SPEECH_ACTIVITY_Q8": "silk_encoder_state.speech_activity_Q8",
    "OPUSGPU_REF_OFF_NLSF_MSVQ_SURVIVORS": "silk_encoder_state.NLSF_MSVQ_Survivors",
    "OPUSGPU_REF_OFF_FIX_SCMN": "silk_encoder_state_FIX.sCmn",
    "OPUSGPU_REF_OFF_MU_LTP_Q9": "silk_encoder_state.mu_LTP_Q9",
    "OPUSGPU_REF_OFF_LTP_QUANT_LOW_COMPLEXITY": "silk_encoder_state.LTPQuantLowComplexity",
    "OPUSGPU_REF_OFF_SUM_LOG_GAIN_Q7": "silk_encoder_state.sum_log_gain_Q7",
    "OPUSGPU_REF_OFF_PACKET_LOSS_PERC": "silk_encoder_state.PacketLoss_perc",
    "OPUSGPU_REF_OFF_N_FRAMES_PER_PACKET": "silk_encoder_state.nFramesPerPacket",
    "OPUSGPU_REF_OFF_LTP_INDEX": "SideInfoIndices.LTPIndex",
    "OPUSGPU_REF_OFF_PER_INDEX": "SideInfoIndices.PERIndex",
    "OPUSGPU_REF_OFF_LTP_SCALE_INDEX": "SideInfoIndices.LTP_scaleIndex",
    "OPUSGPU_REF_SIZEOF_SILK_ENCODER_CONTROL_FIX": "sizeof.silk_encoder_control_FIX",
    "OPUSGPU_REF_OFF_CTRL_GAINS_Q16": "silk_encoder_control_FIX.Gains_Q16",
    "OPUSGPU_REF_OFF_CTRL_PRED_COEF_Q12": "silk_encoder_control_FIX.PredCoef_Q12",
    "OPUSGPU_REF_OFF_CTRL_LTP_COEF_Q14": "silk_encoder_control_FIX.LTPCoef_Q14",
    "OPUSGPU_REF_OFF_CTRL_LTP_SCALE_Q14": "silk_encoder_control_FIX.LTP_scale_Q14",
    "OPUSGPU_REF_OFF_CTRL_PITCHL": "silk_encoder_control_FIX.pitchL",
    "OPUSGPU_REF_OFF_CTRL_CODING_QUALITY_Q14": "silk_encoder_control_FIX.coding_quality_Q14",
    "OPUSGPU_REF_OFF_CTRL_LTP_RED_COD_GAIN_Q7": "silk_encoder_control_FIX.LTPredCodGain_Q7",
    "OPUSGPU_REF_OFF_CTRL_RES_NRG": "silk_encoder_control_FIX.ResNrg",
    "OPUSGPU_REF_OFF_CTRL_RES_NRG_Q": "silk_encoder_control_FIX.ResNrgQ",
    "OPUSGPU_REF_OFF_FS_KHZ": "silk_encoder_state.fs_kHz",
    "OPUSGPU_REF_OFF_LA_PITCH": "silk_encoder_state.la_pitch",
    "OPUSGPU_REF_OFF_LA_SHAPE": "silk_encoder_state.la_shape",
    "OPUSGPU_REF_OFF_SHAPE_WIN_LENGTH": "silk_encoder_state.shapeWinLength",
    "OPUSGPU_REF_OFF_PITCH_LPC_WIN_LENGTH": "silk_encoder_state.pitch_LPC_win_length",
    "OPUSGPU_REF_OFF_PITCH_EST_LPC_ORDER": "silk_encoder_state.pitchEstimationLPCOrder",
    "OPUSGPU_REF_OFF_PITCH_EST_COMPLEXITY": "silk_encoder_state.pitchEstimationComplexity",
    "OPUSGPU_REF_OFF_PITCH_EST_THRESHOLD_Q16": "silk_encoder_state.pitchEstimationThreshold_Q16",
    "OPUSGPU_REF_OFF_SNR_DB_Q7": "silk_encoder_state.SNR_dB_Q7",
    "OPUSGPU_REF_OFF_USE_CBR": "silk_encoder_state.useCBR",
    "OPUSGPU_REF_OFF_INPUT_QUALITY_BANDS_Q15": "silk_encoder_state.input_quality_bands_Q15",
    "OPUSGPU_REF_OFF_INPUT_TILT_Q15": "silk_encoder_state.input_tilt_Q15",
    "OPUSGPU_REF_OFF_PREV_SIGNAL_TYPE": "silk_encoder_state.prevSignalType",
    "OPUSGPU_REF_OFF_PREV_LAG": "silk_encoder_state.prevLag",
    "OPUSGPU_REF_OFF_GAINS_INDICES": "SideInfoIndices.GainsIndices",
    "OPUSGPU_REF_OFF_LAG_INDEX": "SideInfoIndices.lagIndex",
    "OPUSGPU_REF_OFF_CONTOUR_INDEX": "SideInfoIndices.contourIndex",
    "OPUSGPU_REF_OFF_FIX_SSHAPE": "silk_encoder_state_FIX.sShape",
    "OPUSGPU_REF_OFF_FIX_SPREFILT": "silk_encoder_state_FIX.sPrefilt",
    "OPUSGPU_REF_OFF_FIX_LTPCORR_Q15": "silk_encoder_state_FIX.LTPCorr_Q15",
    "OPUSGPU_REF_OFF_SHAPE_LAST_GAIN_INDEX": "silk_shape_state_FIX.LastGainIndex",
    "OPUSGPU_REF_OFF_SHAPE_HARM_BOOST_SMTH_Q16": "silk_shape_state_FIX.HarmBoost_smth_Q16",
    "OPUSGPU_REF_OFF_SHAPE_HARM_SHAPE_GAIN_SMTH_Q16": "silk_shape_state_FIX.HarmShapeGain_smth_Q16",
    "OPUSGPU_REF_OFF_SHAPE_TILT_SMTH_Q16": "silk_shape_state_FIX.Tilt_smth_Q16",
    "OPUSGPU_REF_OFF_CTRL_AR1_Q13": "silk_encoder_control_FIX.AR1_Q13",
    "OPUSGPU_REF_OFF_CTRL_AR2_Q13": "silk_encoder_control_FIX.AR2_Q13",
    "OPUSGPU_REF_OFF_CTRL_LF_SHP_Q14": "silk_encoder_control_FIX.LF_shp_Q14",
    "OPUSGPU_REF_OFF_CTRL_GAINS_PRE_Q14": "silk_encoder_control_FIX.GainsPre_Q14",
    "OPUSGPU_REF_OFF_CTRL_HARM_BOOST_Q14": "silk_encoder_control_FIX.HarmBoost_Q14",
    "OPUSGPU_REF_OFF_CTRL_TILT_Q14": "silk_encoder_control_FIX.Tilt_Q14",
    "OPUSGPU_REF_OFF_CTRL_HARM_SHAPE_GAIN_Q14": "silk_encoder_control_FIX.HarmShapeGain_Q14",
    "OPUSGPU_REF_OFF_CTRL_LAMBDA_Q10": "silk_encoder_control_FIX.Lambda_Q10",
    "OPUSGPU_REF_OFF_CTRL_INPUT_QUALITY_Q14": "silk_encoder_control_FIX.input_quality_Q14",
    "OPUSGPU_REF_OFF_CTRL_SPARSENESS_Q8": "silk_encoder_control_FIX.sparseness_Q8",
    "OPUSGPU_REF_OFF_CTRL_PRED_GAIN_Q16": "silk_encoder_control_FIX.predGain_Q16",
    "OPUSGPU_REF_OFF_CTRL_GAINS_UNQ_Q16": "silk_encoder_control_FIX.GainsUnq_Q16",
    "OPUSGPU_REF_OFF_CTRL_LAST_GAIN_INDEX_PREV": "silk_encoder_control_FIX.lastGainIndexPrev",
    "OPUSGPU_REF_SIZEOF_SILK_PREFILTER_STATE_FIX": "sizeof.silk_prefilter_state_FIX",
    "OPUSGPU_REF_SIZEOF_SILK_ENCODER_STATE_FIX": "sizeof.silk_encoder_state_FIX",
    "OPUSGPU_REF_OFF_SVAD": "silk_encoder_state.sVAD",
    "OPUSGPU_REF_SIZEOF_SILK_VAD_STATE": "sizeof.silk_VAD_state",
}


def header_defines():
    src = open(os.path.join(ROOT, "include", "opusgpu_hooks.h")).read()
    return {k: int(v) for k, v in re.findall(r"#define\s+(OPUSGPU_REF_\w+)\s+(\d+)", src)}


@pytest.mark.ref
def test_hook_struct_offsets_match_the_reference_headers():
    if not os.path.exists(LAYOUT):
        pytest.skip("oracle/_ref/layout.json not built")
    ref = json.load(open(LAYOUT))
    have = header_defines()
    assert set(have) == set(PAIRS)
    for macro, key in PAIRS.items():
        assert have[macro] == ref[key], (macro, have[macro], ref[key])
    assert ref["sizeof.silk_nsq_state"] == 4380          # == sizeof(opusgpu_nsq_state), include/opusgpu_silk.h
