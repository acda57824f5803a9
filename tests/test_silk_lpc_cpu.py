"""CPU tier for silk_find_LPC_FIX (first slice): the device sources (concentus_amd/csrc/silk_lpc_dev.h,
silk_burg_dev.h) compiled for the host (tests/emu) against records freshly captured from the UNMODIFIED reference encoder
(oracle/_ref/libopus_ref_silkcap.so wraps silk_find_LPC_FIX, oracle/ref_silk_capture.c): NLSF_Q15 and NLSFInterpCoef_Q2 of
every record, at complexity 3 (Burg + silk_A2NLSF) and 5 / 8 / 10 (second Burg analysis + the four-factor interpolation
search through silk_NLSF2A / silk_LPC_analysis_filter / silk_sum_sqr_shift)."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus


@pytest.mark.ref
def test_find_lpc_sources_match_the_reference_on_fresh_records():
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "lpc", cache=tmp, workers=4)
        lin = np.ascontiguousarray(c["lpc_in"])
        want = np.asarray(c["lpc_out"])
        n = lin.shape[0]
        got = np.zeros((n, 40), np.uint8)
        emu.emu_silk_find_lpc(lin.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = lin[:, 768:792].view(np.int32)            # minInvGain, subfr_length, nb_subfr, order, useInterp, first_frame
        interp = want[:, 32:36].view(np.int32)[:, 0]
        assert set(np.unique(hdr[:, 4])) == {0, 1}, "both the plain and the interpolating path must be in the corpus"
        assert (interp < 4).sum() > 50, "the interpolation search must win somewhere"
        bad = np.nonzero((got[:, :36] != want[:, :36]).any(1))[0]
        assert bad.size == 0, (bad[:8], got[bad[:2], :36].view(np.int16), want[bad[:2], :36].view(np.int16))
        del c
