"""CPU tier for the decoder: the decoder SOURCES (concentus_amd/csrc/celt_dec.h, rangedec.h), compiled for the
host with CA_HOST_EMU, against committed opus_decode() outputs of the compiled reference (tests/golden/
decode_golden.npz, made by tests/golden/make_golden.py): PCM and final range, bit-exact, for every golden
encode case (each stream through its own fresh decoder) and three low-rate streams that exercise band
folding, noise fill, intensity stereo and anti-collapse."""
import ctypes as C
import os

import numpy as np
import pytest

import emulib
import encode_cases as ec

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "decode_golden.npz")


def decode_cases():
    gm = ec.golden_module()
    return [(c[0], c[3], True) for c in gm.ENCODE_CASES] + [(c[0], c[3], False) for c in gm.DECODE_EXTRA_CASES]


def load_decode_case(name, from_encode):
    g = np.load(GOLD)
    if from_encode:
        _pcm, pk, ln, rg = ec.load_case(name)
    else:
        pk, ln, rg = g[name + "_packets"], g[name + "_len"], g[name + "_rng"]
    pk, ln = np.ascontiguousarray(pk), np.ascontiguousarray(ln.astype(np.int32))
    if name + "_dpcm" in g.files:
        return pk, ln, rg, g[name + "_dpcm"]
    # the real-audio cases carry no committed PCM (3.3 MB each): their expected output comes from the live reference
    gm = ec.golden_module()
    if not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "librefdrv.so")):
        pytest.skip("no committed PCM for this case and oracle/_ref is absent")
    fps = [c[3] for c in gm.ENCODE_CASES if c[0] == name][0]
    want, wrng, wret = gm.ref_decode(pk, ln, fps)
    assert (wret == 960).all() and np.array_equal(wrng, rg)
    return pk, ln, rg, want


@pytest.mark.parametrize("case", decode_cases(), ids=lambda c: c[0])
def test_emulated_decoder_matches_reference_pcm(case):
    name, fps, from_encode = case
    pk, ln, rg, want = load_decode_case(name, from_encode)
    n = pk.shape[0]
    emu = emulib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pcm = np.zeros((n, 960, 2), np.int16)
    rng = np.zeros(n, np.uint32)
    ret = np.zeros(n, np.int32)
    emu.emu_celt_decode_frames(p(pk), pk.shape[1], p(ln), n, fps, p(pcm), p(rng), p(ret))
    assert (ret == 960).all()
    assert np.array_equal(rng, rg), "final range differs (it must equal the encoder's, tests/test_opus_encode.c:305)"
    assert np.array_equal(pcm, want), "PCM differs at frame %d" % int(np.nonzero((pcm != want).reshape(n, -1).any(1))[0][0])


def test_decoder_rejects_what_it_does_not_implement():
    emu = emulib.lib()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pk = np.zeros((3, 8), np.uint8)
    pk[0, 0] = 0x78          # SILK-only TOC
    pk[1, 0] = 0xFC          # CELT FB 20 ms stereo, but a 1-byte packet: DTX/PLC
    pk[2, 0] = 0xFD          # code 1 (two frames)
    ln = np.array([8, 1, 8], np.int32)
    pcm = np.zeros((3, 960, 2), np.int16)
    rng = np.zeros(3, np.uint32)
    ret = np.zeros(3, np.int32)
    emu.emu_celt_decode_frames(p(pk), 8, p(ln), 3, 1, p(pcm), p(rng), p(ret))
    assert ret.tolist() == [-5, -5, -5]
