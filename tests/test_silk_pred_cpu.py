"""CPU tier for silk_find_pred_coefs_FIX, whole (third slice): the device sources
(concentus_amd/csrc/silk_pred_dev.h: silk_ltp_dev.h + silk_nlsf_dev.h + silk_lpc_dev.h + silk_burg_dev.h) compiled for the host
(tests/emu) against records freshly captured from the UNMODIFIED reference encoder (oracle/_ref/libopus_ref_silkcap.so wraps
silk_find_pred_coefs_FIX, oracle/ref_silk_capture.c): every field the call writes -- PredCoef_Q12, LTPCoef_Q14, the quantised
NLSFs, ResNrg / ResNrgQ, LTPredCodGain_Q7, LTP_scale_Q14, sum_log_gain_Q7, NLSFIndices, NLSFInterpCoef_Q2, LTPIndex, PERIndex,
LTP_scaleIndex -- on voiced and unvoiced frames at complexities 3 / 5 / 8 / 10."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus


@pytest.mark.ref
def test_find_pred_coefs_sources_match_the_reference_on_fresh_records():
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "fpc", cache=tmp, workers=4)
        fin = np.ascontiguousarray(c["fpc_in"])
        want = np.asarray(c["fpc_out"])
        n = fin.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["fpc_out"]), np.uint8)
        emu.emu_silk_find_pred_coefs(fin.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = fin[:, 2624:2688].view(np.int32)     # nb_subfr, subfr_length, order, ltp_mem, signalType, condCoding, ...
        assert (hdr[:, 4] == 2).sum() > 1000 and (hdr[:, 4] != 2).sum() > 200, "voiced and unvoiced frames"
        bad = np.nonzero((got[:, :204] != want[:, :204]).any(1))[0]
        if bad.size:
            k = bad[0]
            cols = np.nonzero(got[k, :204] != want[k, :204])[0]
            raise AssertionError((bad.size, bad[:8], "signalType", hdr[k, 4], "first differing bytes", cols[:12]))
        del c


@pytest.mark.ref
def test_process_gains_sources_match_the_reference_on_fresh_records():
    """silk_process_gains_FIX (fourth slice; concentus_amd/csrc/silk_gains_dev.h): quantised and unquantised gains,
    GainsIndices, LastGainIndex, lastGainIndexPrev, quantOffsetType, Lambda_Q10 of every captured call. Several frames of one
    packet are coded conditionally (the reference's bitrate loop calls the function again after changing the gains): both kinds
    must be in the corpus."""
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "gains", cache=tmp, workers=4)
        gin = np.ascontiguousarray(c["gains_in"])
        want = np.asarray(c["gains_out"])
        n = gin.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["gains_out"]), np.uint8)
        emu.emu_silk_process_gains(gin.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = gin[:, 48:112].view(np.int32)
        assert (hdr[:, 1] == 2).sum() > 1000 and (hdr[:, 1] != 2).sum() > 200, "voiced and unvoiced frames"
        bad = np.nonzero((got[:, :52] != want[:, :52]).any(1))[0]
        assert bad.size == 0, (bad.size, bad[:8], got[bad[:1], :52].view(np.int32), want[bad[:1], :52].view(np.int32))
        del c


@pytest.mark.ref
def test_noise_shape_analysis_sources_match_the_reference_on_fresh_records():
    """silk_noise_shape_analysis_FIX (fifth slice; concentus_amd/csrc/silk_shape_dev.h): every field the call
    writes. Complexity 3 records take the plain autocorrelation (_celt_autocorr), 5 / 8 / 10 the warped one with shaping orders
    12 / 16 / 16."""
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "shape", cache=tmp, workers=4)
        sin_ = np.ascontiguousarray(c["shape_in"])
        want = np.asarray(c["shape_out"])
        n = sin_.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["shape_out"]), np.uint8)
        emu.emu_silk_noise_shape_analysis(sin_.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = sin_[:, 1600:1696].view(np.int32)
        assert (hdr[:, 6] == 0).sum() > 500 and (hdr[:, 6] > 0).sum() > 500, "plain and warped autocorrelation"
        assert (hdr[:, 10] == 2).sum() > 1000 and (hdr[:, 10] != 2).sum() > 200, "voiced and unvoiced frames"
        bad = np.nonzero((got[:, :380] != want[:, :380]).any(1))[0]
        if bad.size:
            k = bad[0]
            cols = np.nonzero(got[k, :380] != want[k, :380])[0]
            raise AssertionError((bad.size, bad[:8], "warping", hdr[k, 6], "signalType", hdr[k, 10], "first differing bytes", cols[:12]))
        del c


@pytest.mark.ref
def test_prefilter_sources_match_the_reference_on_fresh_records():
    """silk_prefilter_FIX (sixth slice; concentus_amd/csrc/silk_prefilter_dev.h): xw_Q3 (the quantiser's input) and
    every byte of silk_prefilter_state_FIX after the call, from the state captured before it."""
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "prefilter", cache=tmp, workers=4)
        xin = np.ascontiguousarray(c["prefilter_in"])
        st = np.array(c["prefilter_state_in"])
        want, want_st = np.asarray(c["prefilter_out"]), np.asarray(c["prefilter_state_out"])
        n = xin.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["prefilter_out"]), np.uint8)
        emu.emu_silk_prefilter(xin.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = xin[:, 864:896].view(np.int32)
        assert (hdr[:, 3] == 2).sum() > 1000 and (hdr[:, 3] != 2).sum() > 200 and (hdr[:, 4] == 0).sum() > 500 and (hdr[:, 4] > 0).sum() > 500
        bad = np.nonzero((got[:, :1280] != want[:, :1280]).any(1))[0]
        assert bad.size == 0, (bad.size, bad[:8], got[bad[:1], :32].view(np.int32), want[bad[:1], :32].view(np.int32))
        bad = np.nonzero((st != want_st).any(1))[0]
        assert bad.size == 0, (bad.size, bad[:8], np.nonzero(st[bad[0]] != want_st[bad[0]])[0][:12])
        del c


@pytest.mark.ref
def test_find_pitch_lags_sources_match_the_reference_on_fresh_records():
    """silk_find_pitch_lags_FIX with silk_pitch_analysis_core (seventh slice; concentus_amd/csrc/silk_pitch_dev.h):
    the whitened buffer res[], pitchL, lagIndex, contourIndex, LTPCorr_Q15, the voicing decision and predGain_Q16 of every
    captured call, at pitch-estimation complexities 1 and 2 (encoder complexity 3 / 5 / 8 / 10)."""
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "pitch", cache=tmp, workers=4)
        tin = np.ascontiguousarray(c["pitch_in"])
        want = np.asarray(c["pitch_out"])
        n = tin.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["pitch_out"]), np.uint8)
        emu.emu_silk_find_pitch_lags(tin.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        st = want[:, 1344 + 28:1344 + 32].view(np.int32)[:, 0]
        assert (st == 2).sum() > 1000 and (st == 1).sum() > 200, "voiced and unvoiced decisions"
        bad = np.nonzero((got[:, :1380] != want[:, :1380]).any(1))[0]
        if bad.size:
            k = bad[0]
            cols = np.nonzero(got[k, :1380] != want[k, :1380])[0]
            raise AssertionError((bad.size, bad[:8], "first differing bytes", cols[:12], got[k, 1344:1380].view(np.int32), want[k, 1344:1380].view(np.int32)))
        del c
