"""CPU tier: DATAFLOW CLOSURE of the SILK analysis chain. The batched entry points are driven from one record
per call; a batched SILK front end has to fill each record from the outputs of the calls before it. This test captures ALL the
analysis functions and both quantisers AT ONCE from one run of the unmodified reference encoder (oracle/ref_silk_capture.c,
refcap_start_chain: every record carries the number of the frame it belongs to) and checks, frame by frame, that every
signal / parameter field of a record equals the output field of the earlier call that INTEGRATION.md names as its source:

    find_pitch_lags -> noise_shape_analysis -> find_pred_coefs -> process_gains -> prefilter -> NSQ / NSQ_del_dec

Together with the per-function bit-exactness tests (tests/test_silk_pred_cpu.py, tests/test_silk_gpu.py) this shows that the
seven kernels compose to what silk_encode_frame_FIX computes for the first pass of a frame; what is NOT produced inside the
chain (the input buffer, VAD outputs, the configuration, and the states carried from the previous frame) is listed at the end."""
import ctypes as C

import numpy as np
import pytest

import silk_corpus
from concentus_amd import silk as S

FID = {"pitch": 0, "shape": 1, "fpc": 2, "gains": 3, "prefilter": 4, "nsq": 5, "dd": 6}


def _capture(complexity, nframes, seed, variant="wb20"):
    lib = C.CDLL(silk_corpus.CAPLIB)
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    cap = 4 * nframes + 16
    lib.refcap_start_chain(cap)
    err = C.c_int()
    fs, frame = silk_corpus.VARIANTS[variant]
    enc = C.c_void_p(lib.opus_encoder_create(fs, 1, 2048, C.byref(err)))
    assert enc and err.value == 0
    for req, v in ((4002, 32000), (4006, 1), (4020, 0), (4010, complexity), (4012, 0), (4016, 0), (4014, 0), (4036, 16)):
        lib.opus_encoder_ctl(enc, req, v)
    pcm = silk_corpus.synth_voice(nframes * frame * (16000 // fs), seed)[::16000 // fs]
    out = (C.c_ubyte * 1500)()
    for f in range(nframes):
        fr = np.ascontiguousarray(pcm[f * frame:(f + 1) * frame])
        assert lib.opus_encode(enc, fr.ctypes.data_as(C.c_void_p), frame, out, 1500) > 0
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def grab(count, getter, classes):
        bufs = [np.zeros(count, np.dtype(c)) for c in classes]
        getter(*[p(b) for b in bufs])
        return bufs

    def fids(kind, n):
        a = np.zeros(max(n, 1), np.int32)
        lib.refcap_get_frame_ids(FID[kind], p(a), n)
        return a[:n]
    r = {}
    n = lib.refcap_count_pitch(); r["pitch"] = grab(n, lib.refcap_get_pitch, (S.FindPitchLagsIn, S.FindPitchLagsOut)) + [fids("pitch", n)]
    n = lib.refcap_count_shape(); r["shape"] = grab(n, lib.refcap_get_shape, (S.NoiseShapeIn, S.NoiseShapeOut)) + [fids("shape", n)]
    n = lib.refcap_count_fpc(); r["fpc"] = grab(n, lib.refcap_get_fpc, (S.FindPredCoefsIn, S.FindPredCoefsOut)) + [fids("fpc", n)]
    n = lib.refcap_count_gains(); r["gains"] = grab(n, lib.refcap_get_gains, (S.ProcessGainsIn, S.ProcessGainsOut)) + [fids("gains", n)]
    n = lib.refcap_count_prefilter()
    r["prefilter"] = grab(n, lib.refcap_get_prefilter, (S.PrefilterIn, S.PrefilterState, S.PrefilterState, S.PrefilterOut)) + [fids("prefilter", n)]
    nb, nn = lib.refcap_count_burg(), lib.refcap_count_nsq()
    if nn:
        b = [np.zeros((nb, 784), np.uint8), np.zeros((nb, 72), np.uint8)]
        q = [np.zeros(nn, np.dtype(S.NsqIn)), np.zeros(nn, np.dtype(S.NsqState)), np.zeros(nn, np.dtype(S.NsqState)), np.zeros((nn, 320), np.int8)]
        lib.refcap_get(*[p(a) for a in b + q])
        r["q"] = (q[0], fids("nsq", nn))
        r["q_full"] = (q[0], q[1], q[2], q[3].view(np.uint8), False)
    else:
        nd = lib.refcap_count_dd()
        q = [np.zeros(nd, np.dtype(S.NsqDdIn)), np.zeros(nd, np.dtype(S.NsqState)), np.zeros(nd, np.dtype(S.NsqState)), np.zeros((nd, 324), np.uint8)]
        lib.refcap_get_dd(*[p(a) for a in q])
        r["q"] = (q[0]["base"], fids("dd", nd))
        r["q_full"] = (q[0], q[1], q[2], q[3], True)
    return r


def _first_of_frame(fid, frame):
    k = np.nonzero(fid == frame)[0]
    return int(k[0]) if k.size else None


@pytest.mark.ref
@pytest.mark.parametrize("complexity,variant", [(3, "wb20"), (8, "wb20"), (7, "nb20"), (5, "wb10")])
def test_every_record_field_is_an_output_of_an_earlier_call(complexity, variant):
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    nframes = 160
    r = _capture(complexity, nframes, 424242 + complexity, variant)
    (pin, pout, pf), (sin_, sout, sf), (fin, fout, ff), (gin, gout, gf), (xin, xst0, xst1, xout, xf), (qin, qf) = (
        r["pitch"], r["shape"], r["fpc"], r["gains"], r["prefilter"], r["q"])
    eq = np.array_equal
    seen_voiced = checked = 0
    prev_pf_state = None
    for frame in range(9, nframes + 1):                       # frame ids count from 1; skip the encoder's start-up frames
        kp, ks, kf, kg, kx, kq = (_first_of_frame(a, frame) for a in (pf, sf, ff, gf, xf, qf))
        assert None not in (kp, ks, kf, kg, kx, kq), (frame, kp, ks, kf, kg, kx, kq)
        P, PO, SI, SO, FI, FO, GI, GO, XI, XO, Q = pin[kp], pout[kp], sin_[ks], sout[ks], fin[kf], fout[kf], gin[kg], gout[kg], xin[kx], xout[kx], qin[kq]
        ltp, fl, la_s, nb = int(P["ltp_mem_length"]), int(P["frame_length"]), int(SI["la_shape"]), int(P["nb_subfr"])
        voiced = int(PO["signalType"]) == 2
        seen_voiced += voiced
        # ---- noise_shape_analysis <- find_pitch_lags (+ the input buffer)
        assert eq(SI["pitch_res"][:fl], PO["res"][ltp:ltp + fl])
        lap = int(P["la_pitch"])
        assert eq(SI["x"][:la_s + fl + lap], P["x_buf"][ltp - la_s:ltp + fl + lap])        # the part of x_buf both records hold
        assert int(SI["signalType"]) == int(PO["signalType"]) and int(SI["LTPCorr_Q15"]) == int(PO["LTPCorr_Q15"])
        assert int(SI["predGain_Q16"]) == int(PO["predGain_Q16"]) and eq(SI["pitchL"][:nb], PO["pitchL"][:nb])
        # ---- find_pred_coefs <- find_pitch_lags, noise_shape_analysis
        assert eq(FI["res_pitch"][:ltp + fl], PO["res"][:ltp + fl]) and eq(FI["x"][:ltp + fl], P["x_buf"][:ltp + fl])
        assert eq(FI["Gains_Q16"][:nb], SO["Gains_Q16"][:nb]) and eq(FI["pitchL"][:nb], PO["pitchL"][:nb])
        assert int(FI["signalType"]) == int(PO["signalType"]) and int(FI["coding_quality_Q14"]) == int(SO["coding_quality_Q14"])
        # ---- process_gains <- noise_shape_analysis, find_pred_coefs
        assert eq(GI["Gains_Q16"][:nb], SO["Gains_Q16"][:nb]) and eq(GI["ResNrg"][:nb], FO["ResNrg"][:nb]) and eq(GI["ResNrgQ"][:nb], FO["ResNrgQ"][:nb])
        assert int(GI["LTPredCodGain_Q7"]) == int(FO["LTPredCodGain_Q7"]) and int(GI["quantOffsetType"]) == int(SO["quantOffsetType"])
        assert int(GI["input_quality_Q14"]) == int(SO["input_quality_Q14"]) and int(GI["coding_quality_Q14"]) == int(SO["coding_quality_Q14"])
        assert int(GI["signalType"]) == int(PO["signalType"]) and int(GI["condCoding"]) == int(FI["condCoding"])
        # ---- prefilter <- noise_shape_analysis, find_pitch_lags (+ the frame itself)
        assert eq(XI["x"][:fl], P["x_buf"][ltp:ltp + fl]) and eq(XI["pitchL"][:nb], PO["pitchL"][:nb])
        order = int(SI["shapingLPCOrder"])
        for k in range(nb):
            assert eq(XI["AR1_Q13"][16 * k:16 * k + order], SO["AR1_Q13"][16 * k:16 * k + order])
        for name in ("HarmShapeGain_Q14", "HarmBoost_Q14", "Tilt_Q14", "GainsPre_Q14", "LF_shp_Q14"):
            assert eq(XI[name][:nb], SO[name][:nb]), name
        assert int(XI["coding_quality_Q14"]) == int(SO["coding_quality_Q14"]) and int(XI["signalType"]) == int(PO["signalType"])
        if prev_pf_state is not None:                         # the prefilter state is carried from frame to frame, nothing else writes it
            assert eq(np.frombuffer(xst0[kx].tobytes(), np.uint8), prev_pf_state)
        prev_pf_state = np.frombuffer(xst1[kx].tobytes(), np.uint8)
        # ---- NSQ / NSQ_del_dec (first pass of the frame) <- everything before
        assert eq(Q["x_Q3"][:fl], XO["xw_Q3"][:fl])
        D = int(FI["predictLPCOrder"])
        assert eq(Q["PredCoef_Q12"][:D], FO["PredCoef_Q12"][:D]) and eq(Q["PredCoef_Q12"][16:16 + D], FO["PredCoef_Q12"][16:16 + D])
        assert eq(Q["LTPCoef_Q14"][:5 * nb], FO["LTPCoef_Q14"][:5 * nb])
        for k in range(nb):
            assert eq(Q["AR2_Q13"][16 * k:16 * k + order], SO["AR2_Q13"][16 * k:16 * k + order])
        for name in ("HarmShapeGain_Q14", "Tilt_Q14", "LF_shp_Q14"):
            assert eq(Q[name][:nb], SO[name][:nb]), name
        assert eq(Q["Gains_Q16"][:nb], GO["Gains_Q16"][:nb]) and eq(Q["pitchL"][:nb], PO["pitchL"][:nb])
        assert int(Q["Lambda_Q10"]) == int(GO["Lambda_Q10"]) and int(Q["quantOffsetType"]) == int(GO["quantOffsetType"])
        assert int(Q["NLSFInterpCoef_Q2"]) == int(FO["NLSFInterpCoef_Q2"]) and int(Q["signalType"]) == int(PO["signalType"])
        if voiced:
            assert int(Q["LTP_scale_Q14"]) == int(FO["LTP_scale_Q14"])
        checked += 1
    assert checked > 100 and seen_voiced > 20 and seen_voiced < checked
    # Not produced inside the chain (inputs of a batched front end): the input buffer x_buf (after the variable low-pass filter),
    # signalType / speech_activity_Q8 / input_quality_bands_Q15 / input_tilt_Q15 from the VAD, the configuration derived from the
    # complexity and the sampling rate (silk/control_codec.c), and what the previous frame left behind: prevLag, prevSignalType,
    # LTPCorr_Q15, prev_NLSFq_Q15, sum_log_gain_Q7, sShape (LastGainIndex, three smoothers), sPrefilt, sNSQ, first_frame_after_reset.
