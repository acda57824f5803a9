"""CPU tier: the SILK analysis chain at the OTHER operating points of the reference -- 8 kHz narrowband input (fs_kHz 8, LPC order 10,
the NB/MB NLSF codebook, shaping / pitch windows of the 8 kHz configuration, the stage-2-only pitch search) and 10 ms frames
(nb_subfr 2: the 10 ms lag code books, 1.5 x NLSF_mu, two-subframe LTP / gain paths). Records captured from the unmodified
reference encoder as in tests/test_silk_pred_cpu.py (tests/silk_corpus.py variants "nb20", "wb10"); the device sources compiled
for the host must reproduce every field the reference wrote."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus

CASES = {      # kind -> (input key(s), output key, bytes compared, emu entry, optional in/out state keys)
    "lpc": ("lpc_in", "lpc_out", 36, "emu_silk_find_lpc", None),
    "fpc": ("fpc_in", "fpc_out", 204, "emu_silk_find_pred_coefs", None),
    "gains": ("gains_in", "gains_out", 52, "emu_silk_process_gains", None),
    "shape": ("shape_in", "shape_out", 380, "emu_silk_noise_shape_analysis", None),
    "prefilter": ("prefilter_in", "prefilter_out", 1280, "emu_silk_prefilter", ("prefilter_state_in", "prefilter_state_out")),
    "pitch": ("pitch_in", "pitch_out", 1380, "emu_silk_find_pitch_lags", None),
    "vad": ("vad_in", "vad_out", 24, "emu_silk_vad", ("vad_state_in", "vad_state_out")),
}


@pytest.mark.ref
@pytest.mark.parametrize("variant", ["wb20", "nb20", "wb10", "wb40"])
@pytest.mark.parametrize("kind", sorted(CASES))
def test_analysis_sources_match_the_reference_at_other_rates_and_frame_sizes(kind, variant):
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    ik, ok, nb, entry, stkeys = CASES[kind]
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(2 * silk_corpus.SEG_FRAMES, kind, cache=tmp, workers=2, variant=variant, complexities=(4, 9))
        rin = np.ascontiguousarray(c[ik])
        want = np.asarray(c[ok])
        n = rin.shape[0]
        got = np.zeros((n, want.shape[1]), np.uint8)
        args = [rin.ctypes.data_as(C.c_void_p)]
        if stkeys:
            st = np.array(c[stkeys[0]])
            args.append(st.ctypes.data_as(C.c_void_p))
        getattr(emu, entry)(*args, got.ctypes.data_as(C.c_void_p), C.c_long(n))
        bad = np.nonzero((got[:, :nb] != want[:, :nb]).any(1))[0]
        assert bad.size == 0, (kind, variant, bad.size, bad[:6], np.nonzero(got[bad[0], :nb] != want[bad[0], :nb])[0][:12])
        if stkeys:
            assert np.array_equal(st, np.asarray(c[stkeys[1]]))
        # the records really are from the other operating point
        if kind == "pitch":
            hdr = rin[:, 1344:1408].view(np.int32)
            assert (hdr[:, 0] == (8 if variant == "nb20" else 16)).all() and (hdr[:, 1] == (2 if variant == "wb10" else 4)).all()
        if kind == "gains" and variant == "wb40":
            assert (rin[:, 48:112].view(np.int32)[:, 6] == 2).sum() > 500, "conditionally coded frames (delta gain indices)"
        if kind == "fpc":
            hdr = rin[:, 2624:2688].view(np.int32)
            assert (hdr[:, 2] == (10 if variant == "nb20" else 16)).all() and (hdr[:, 0] == (2 if variant == "wb10" else 4)).all()
        del c
