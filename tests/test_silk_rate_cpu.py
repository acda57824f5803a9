"""CPU tier for the bitrate-control loop of silk_encode_frame_FIX (tenth slice;
opus-fix/silk/fixed/encode_frame_FIX.c:263-423): concentus_amd/csrc/silk_rate_dev.h compiled for the host and driven with what the
UNMODIFIED reference encoder did, pass by pass (oracle/ref_silk_capture.c notes every quantiser / entropy-coder call of a frame):
the step is given ec_tell() of the reference's pass k and must ask for a pass k + 1 with exactly the gain indices, quantised gains
and Lambda_Q10 the reference used next -- or finish where the reference finished, with its LastGainIndex and gain indices.
Constant-bitrate encoders (the loop runs on every frame, 1-7 passes) and a VBR encoder squeezed by a small max_data_bytes."""
import ctypes as C
import tempfile

import numpy as np
import pytest

import emulib
import silk_corpus
from concentus_amd import silk as S


def fresh_ctl(c, n):
    """opusgpu_silk_rate_ctl of each frame before the first step: the arguments of silk_encode_frame_FIX and what silk_process_gains_FIX
    left in sEncCtrl / psEnc->sShape / indices."""
    ctl = np.zeros(n, dtype=np.dtype(S.RateCtl))
    args = np.asarray(c["c_frame_args"]).view(np.int32)
    gout = np.ascontiguousarray(c["c_gains_out"]).view(np.dtype(S.ProcessGainsOut))[:, 0]
    gin = np.ascontiguousarray(c["c_gains_in"]).view(np.dtype(S.ProcessGainsIn))[:, 0]
    ctl["condCoding"], ctl["maxBits"], ctl["useCBR"] = args[:, 0], args[:, 1], args[:, 2]
    ctl["nb_subfr"] = gin["nb_subfr"]
    ctl["frame_length"] = gin["nb_subfr"] * gin["subfr_length"]
    for k in ("GainsUnq_Q16", "Gains_Q16", "lastGainIndexPrev", "LastGainIndex", "Lambda_Q10", "GainsIndices"):
        ctl[k] = gout[k]
    return ctl


@pytest.mark.ref
@pytest.mark.parametrize("kind,variant", [("chain_dd", "wb20cbr"), ("chain_nsq", "nb20cbr"), ("chain_dd", "wb20lo"), ("chain_dd", "wb20")])
def test_rate_loop_follows_the_reference_pass_by_pass(kind, variant):
    if not silk_corpus.available():
        pytest.skip("oracle/_ref/libopus_ref_silkcap.so not built")
    emu = emulib.lib()
    with tempfile.TemporaryDirectory() as tmp:
        n = 2 * silk_corpus.SEG_FRAMES
        c = silk_corpus.corpus(n, kind, cache=tmp, workers=2, variant=variant)
        ctl = fresh_ctl(c, n)
        passes = np.asarray(c["c_frame_passes"])
        pv = passes.view(np.int32)
        npass = pv[:, 0].copy()
        misc = np.asarray(c["c_frame_misc"])
        alive = np.ones(n, bool)
        for k in range(silk_corpus.MAX_PASSES):
            base = 1 + 7 * k
            nbits = np.ascontiguousarray(pv[:, base + 1])
            emu.emu_silk_rate_control(ctl.ctypes.data_as(C.c_void_p), nbits.ctypes.data_as(C.c_void_p), C.c_long(n))
            assert (ctl["status"] == 0).all()
            # frames the reference coded again: same request here, with the next pass's gain indices / gains / Lambda
            more = alive & (npass > k + 1)
            assert np.array_equal(ctl["recode"][alive] == 1, more[alive]), (variant, k)
            assert np.array_equal(ctl["done"][alive] == 0, more[alive]), (variant, k)
            if more.any() and k + 1 < silk_corpus.MAX_PASSES:
                nb = 1 + 7 * (k + 1)
                assert np.array_equal(ctl["GainsIndices"][more].view(np.uint8), passes[more, 4 * nb:4 * nb + 4]), (variant, k)
                nsub = ctl["nb_subfr"][more]
                got_g, want_g = ctl["Gains_Q16"][more], pv[more, nb + 2:nb + 6]
                for j in range(4):
                    rows = nsub > j
                    assert np.array_equal(got_g[rows, j], want_g[rows, j]), (variant, k, j)
                assert np.array_equal(ctl["Lambda_Q10"][more], pv[more, nb + 6]), (variant, k)
            alive = more
        assert not alive.any()
        assert np.array_equal(ctl["passes"], npass)
        # where the loop ends: the gain state the next frame starts from
        assert np.array_equal(ctl["LastGainIndex"], misc[:, 324:328].copy().view(np.int32)[:, 0]), variant
        nsub = ctl["nb_subfr"]
        for j in range(4):
            rows = nsub > j
            assert np.array_equal(ctl["GainsIndices"][rows, j].view(np.uint8), misc[rows, 320 + j]), (variant, j)
        hist = np.bincount(npass, minlength=8)
        if variant.endswith("cbr"):
            assert hist[2:].sum() > n // 2 and hist[5:].sum() > 50, hist                 # the loop really iterates
            assert (ctl["restore2"] == 1).sum() > 0 or (ctl["found_lower"] & ctl["found_upper"]).sum() > 100, "bracketing paths"
        elif variant == "wb20lo":
            assert hist[2:].sum() > 20, hist
        else:
            assert hist[1] == n
