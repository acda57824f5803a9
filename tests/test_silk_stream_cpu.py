"""CPU tier: DATAFLOW CLOSURE OF THE SILK CHAIN ACROSS FRAMES ("end to end from PCM instead of from captured
records"). tests/test_silk_chain_cpu.py pins where every record field comes from INSIDE a frame; this test pins what
silk_encode_frame_FIX carries from one frame of a stream to the next -- the specification of the device-side carry
(opusgpu_silk_stream_carry_in / _out, csrc/silk_stream.hip, concentus_amd/silk_stream.py). On consecutive frames of one run of the
unmodified reference encoder (the chain corpus keeps a segment's frames in order) every "carried" input field of frame t + 1 must
equal the output / state frame t left behind:

    x_buf                         <- x_buf shifted by one frame (encode_frame_FIX.c:427) + the new input (:145)
    prevLag, prevSignalType       <- pitchL[nb_subfr - 1], indices.signalType                    (:437-438)
    first_frame_after_reset       <- 0                                                            (:441)
    LTPCorr_Q15                   <- silk_find_pitch_lags_FIX's                                   (find_pitch_lags_FIX.c:128-140)
    sShape smoothers              <- silk_noise_shape_analysis_FIX's                              (noise_shape_analysis_FIX.c:431-449)
    prev_NLSFq_Q15, sum_log_gain  <- silk_find_pred_coefs_FIX's                                   (find_pred_coefs_FIX.c:139-147, :82)
    sShape.LastGainIndex          <- the bitrate loop's final one                                 (:263-423)
    sPrefilt, sNSQ                <- the states the calls update in place
    Seed                          <- frameCounter++ & 3                                           (:128)
    ec_prevSignalType / LagIndex  <- silk_encode_indices'                                         (encode_indices.c:180-181)
Everything else in the records is configuration, the frame's own input (inputBuf, VAD results, SNR_dB_Q7, maxBits, condCoding --
computed outside the frame function) or filled inside the frame (CHAIN_FED_FIELDS)."""
import numpy as np
import pytest

import silk_corpus
from concentus_amd import silk as S


def view(a, cls):
    return np.ascontiguousarray(a).view(np.dtype(cls))[:, 0]


@pytest.mark.ref
@pytest.mark.parametrize("kind,variant", [("chain_dd", "wb20cbr"), ("chain_nsq", "nb20cbr"), ("chain_dd", "wb20")])
def test_what_a_frame_carries_to_the_next(kind, variant):
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    n = 384                                                    # one segment: consecutive frames of one encoder
    rec = {k: np.array(v) for k, v in silk_corpus.corpus(n, kind, variant=variant).items()}
    P, PO = view(rec["c_pitch_in"], S.FindPitchLagsIn), view(rec["c_pitch_out"], S.FindPitchLagsOut)
    SI, SO = view(rec["c_shape_in"], S.NoiseShapeIn), view(rec["c_shape_out"], S.NoiseShapeOut)
    FI, FO = view(rec["c_fpc_in"], S.FindPredCoefsIn), view(rec["c_fpc_out"], S.FindPredCoefsOut)
    GI = view(rec["c_gains_in"], S.ProcessGainsIn)
    XI = view(rec["c_prefilter_in"], S.PrefilterIn)
    dd = kind == "chain_dd"
    Q = view(rec["c_q_in"], S.NsqDdIn)["base"] if dd else view(rec["c_q_in"], S.NsqIn)
    BI, BO = view(rec["c_bits_in"], S.SilkBitsIn), view(rec["c_bits_out"], S.SilkBitsOut)
    misc = rec["c_frame_misc"]
    last_gain = misc[:, 324:328].copy().view(np.int32)[:, 0]
    eq = np.array_equal
    checked = voiced = 0
    for t in range(n - 1):
        a, b = t, t + 1
        fs, nb = int(P["fs_kHz"][a]), int(P["nb_subfr"][a])
        fl, ltp, la_s, la_p = int(P["frame_length"][a]), int(P["ltp_mem_length"][a]), int(SI["la_shape"][a]), int(P["la_pitch"][a])
        assert (fl, ltp, la_s) == (5 * fs * nb, 20 * fs, 5 * fs)
        # the whole x_buf of frame t: [0, ltp + fl + la_p) from the pitch record, the rest (up to ltp + la_s + fl) from the shaping record
        xb = np.zeros(ltp + la_s + fl, np.int16)
        xb[:ltp + fl + la_p] = P["x_buf"][a][:ltp + fl + la_p]
        xb[ltp - la_s:] = SI["x"][a][:fl + 2 * la_s]
        assert eq(xb[ltp - la_s:ltp + fl + la_p], P["x_buf"][a][ltp - la_s:ltp + fl + la_p]), "the two records overlap consistently"
        # ---- x_buf: shifted by one frame, the new input behind it
        assert eq(P["x_buf"][b][:ltp + la_s], xb[fl:fl + ltp + la_s]), t
        assert eq(FI["x"][b][:ltp + la_s], xb[fl:fl + ltp + la_s]) and eq(XI["x"][b][:la_s], xb[fl + ltp:fl + ltp + la_s])
        # ---- scalars of the pitch analysis
        final_type = int(BI["signalType"][a])                  # indices.signalType as coded = after find_pitch_lags
        assert int(P["prevSignalType"][b]) == final_type and int(P["first_frame_after_reset"][b]) == 0
        assert int(P["prevLag"][b]) == int(PO["pitchL"][a][nb - 1]), (t, int(P["prevLag"][b]), PO["pitchL"][a])
        assert int(P["LTPCorr_Q15"][b]) == int(PO["LTPCorr_Q15"][a]) and int(SI["LTPCorr_Q15"][b]) == int(PO["LTPCorr_Q15"][b])
        # ---- noise shaping smoothers
        for f in ("HarmBoost_smth_Q16", "HarmShapeGain_smth_Q16", "Tilt_smth_Q16"):
            assert int(SI[f][b]) == int(SO[f][a]), f
        # ---- prediction: previous quantised NLSFs, the LTP gain limiter's running sum
        D = int(FI["predictLPCOrder"][a])
        assert eq(FI["prev_NLSFq_Q15"][b][:D], FO["NLSF_Q15"][a][:D]) and int(FI["first_frame_after_reset"][b]) == 0
        assert int(FI["sum_log_gain_Q7"][b]) == int(FO["sum_log_gain_Q7"][a])
        # ---- gains: LastGainIndex after the bitrate loop
        assert int(GI["LastGainIndex"][b]) == int(last_gain[a])
        # ---- states updated in place
        assert eq(rec["c_prefilter_state_in"][b], rec["c_prefilter_state_out"][a])
        assert eq(rec["c_q_state_in"][b], rec["c_frame_nsq"][a]), "sNSQ after the frame (loop included) is the next frame's"
        # ---- Seed = frameCounter++ & 3; the entropy coder's conditional-coding memory
        assert int(Q["Seed"][b]) == (int(Q["Seed"][a]) + 1) & 3
        assert int(BI["ec_prevSignalType"][b]) == int(BO["ec_prevSignalType"][a]) and int(BI["ec_prevLagIndex"][b]) == int(BO["ec_prevLagIndex"][a])
        checked += 1
        voiced += final_type == 2
    assert checked == n - 1 and 20 < voiced < checked
