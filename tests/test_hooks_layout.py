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
