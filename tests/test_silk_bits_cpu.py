"""CPU tier for silk_encode_indices and silk_encode_pulses (eighth slice): the device sources
(concentus_amd/csrc/silk_bits_dev.h + rangecoder.h) compiled for the host against the range coder of the UNMODIFIED reference encoder
captured before and after each call (oracle/ref_silk_capture.c): every ec_ctx field (offs, rng, val, ext, rem, nbits_total, ...) and
every byte the coder has written, plus ec_prevSignalType / ec_prevLagIndex. Wideband 20 ms, narrowband, 10 ms frames, and 40 ms packets (two frames on one coder, the second coded conditionally: delta gain and delta pitch-lag models)."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus


@pytest.mark.ref
@pytest.mark.parametrize("variant", ["wb20", "nb20", "wb10", "wb40"])
def test_side_information_and_excitation_coding_match_the_reference(variant):
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        c = silk_corpus.corpus(2 * silk_corpus.SEG_FRAMES, "bits", cache=tmp, workers=2, variant=variant, complexities=(3, 8))
        for tag in ("idx", "pls"):
            rin = np.ascontiguousarray(c["bits_%s_in" % tag])
            ec = np.array(c["bits_%s_ec_in" % tag])
            want_ec, want_out = np.asarray(c["bits_%s_ec_out" % tag]), np.asarray(c["bits_%s_out" % tag])
            n = rin.shape[0]
            out = np.zeros((n, 16), np.uint8)
            emu.emu_silk_encode_bits(rin.ctypes.data_as(C.c_void_p), ec.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(n))
            bad = np.nonzero((ec != want_ec).any(1))[0]
            assert bad.size == 0, (tag, variant, bad.size, bad[:6], np.nonzero(ec[bad[0]] != want_ec[bad[0]])[0][:12])
            if tag == "idx":
                assert np.array_equal(out[:, :8], want_out[:, :8])
                hdr = rin[:, 348:412].view(np.int32)
                assert (hdr[:, 2] == 2).sum() > 500, "voiced frames"
                if variant == "wb40":
                    assert (hdr[:, 12] == 2).sum() > 1000, "the second frame of every 40 ms packet is coded conditionally"
            grew = want_ec[:, 20:24].view(np.uint32)[:, 0] - np.asarray(c["bits_%s_ec_in" % tag])[:, 20:24].view(np.uint32)[:, 0]
            assert grew.max() > 4, "the calls must really write bytes"
        # the two calls of one frame are consecutive on the same coder: indices-out is pulses-in
        assert np.array_equal(np.asarray(c["bits_idx_ec_out"]), np.asarray(c["bits_pls_ec_in"]))
        del c
