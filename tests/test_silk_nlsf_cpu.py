"""CPU tier for silk_process_NLSFs and silk_residual_energy_FIX (second slice: the tail of
silk_find_pred_coefs_FIX): the device sources (concentus_amd/csrc/silk_nlsf_dev.h) compiled for the host (tests/emu) against
records freshly captured from the UNMODIFIED reference encoder (oracle/_ref/libopus_ref_silkcap.so wraps both functions,
oracle/ref_silk_capture.c): NLSFIndices, the quantised NLSFs, both PredCoef_Q12 rows, and the four residual energies with
their Q values, at complexities 3 / 5 / 8 / 10 (different survivor counts, with and without NLSF interpolation)."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus


@pytest.mark.ref
def test_nlsf_and_residual_energy_sources_match_the_reference_on_fresh_records():
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(4 * silk_corpus.SEG_FRAMES, "pred", cache=tmp, workers=4)
        nin = np.ascontiguousarray(c["nlsf_in"])
        want = np.asarray(c["nlsf_out"])
        n = nin.shape[0]
        got = np.zeros((n, silk_corpus.SIZES["nlsf_out"]), np.uint8)
        emu.emu_silk_process_nlsfs(nin.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), C.c_long(n))
        hdr = nin[:, 64:96].view(np.int32)     # speech_activity, nb_subfr, order, useInterp, interpCoef, survivors, signalType
        assert len(np.unique(hdr[:, 5])) >= 3, "several survivor counts must be in the corpus"
        assert (hdr[:, 4] < 4).sum() > 50, "interpolated frames must be in the corpus"
        assert set(np.unique(hdr[:, 6])) >= {1, 2}, "voiced and unvoiced frames"
        bad = np.nonzero((got[:, :116] != want[:, :116]).any(1))[0]
        assert bad.size == 0, (bad.size, bad[:8], got[bad[:1], 96:113].view(np.int8), want[bad[:1], 96:113].view(np.int8))

        ein = np.ascontiguousarray(c["resnrg_in"])
        ewant = np.asarray(c["resnrg_out"])
        egot = np.zeros((ein.shape[0], silk_corpus.SIZES["resnrg_out"]), np.uint8)
        emu.emu_silk_residual_energy(ein.ctypes.data_as(C.c_void_p), egot.ctypes.data_as(C.c_void_p), C.c_long(ein.shape[0]))
        bad = np.nonzero((egot[:, :32] != ewant[:, :32]).any(1))[0]
        assert bad.size == 0, (bad.size, bad[:8], egot[bad[:1], :32].view(np.int32), ewant[bad[:1], :32].view(np.int32))
        del c
