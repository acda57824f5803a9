"""Randomised parity sweep on the CPU tier: the kernel sources under host emulation (split encoder pipeline and the
decoder) against the compiled reference, live, over random settings -- bitrate 32-510 kb/s, VBR / constrained VBR /
CBR, complexity 0-10, lsb_depth, expected loss, max_data_bytes, noise / music / edge input at several levels,
independent frames and streams. Needs oracle/_ref (this container); the committed fixtures cover the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

import emulib
import encode_cases as ec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")),
                                reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("seed,lane", [(101, False), (202, False), (303, False), (404, True), (505, True), (606, True)])
def test_random_settings_match_reference(seed, lane):
    """lane = True: the lane-per-frame variant of the sources (tests/emu/celt_lane_emu.cpp: what celt_back_lane_kernel /
    celt_decode_lane_kernel run), in a random column of the LDS image"""
    gm = ec.golden_module()
    if lane and not os.path.exists(emulib.HOST_CLANG):
        pytest.skip("host clang of the ROCm image not present")
    emu = emulib.lane_lib() if lane else emulib.lib()
    enc_fn = emu.emu_lane_celt_encode_frames if lane else emu.emu_celt_encode_frames_split
    dec_fn = emu.emu_lane_celt_decode_frames if lane else emu.emu_celt_decode_frames
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(seed)
    for _t in range(20):
        br = int(rng.choice([32000, 33000, 34500, 36000, 38000, 38400, 40000, 48000, 64000, 96000, 128000, 192000, 320000, 510000]))
        vbr, cvbr, cx = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 11))
        lsb, loss = int(rng.choice([8, 12, 16, 24])), int(rng.choice([0, 0, 3, 5, 10, 25]))
        maxb = int(rng.choice([1500, 1276, 800, 400, 250, 120]))
        kind, fps, n = str(rng.choice(["noise", "music", "edge"])), int(rng.choice([1, 4, 16])), 32
        pcm = gm.synth_pcm(kind, n, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.3:
            pcm = (pcm.astype(np.int32) * int(rng.choice([0, 1, 3])) // int(rng.choice([1, 4, 64]))).clip(-32768, 32767).astype(np.int16)
        what = dict(br=br, vbr=vbr, cvbr=cvbr, cx=cx, lsb=lsb, loss=loss, maxb=maxb, kind=kind, fps=fps)
        pk, ln, rg = gm.ref_encode(gm._Cfg(2, br, vbr, cvbr, cx, lsb, loss, maxb), pcm, fps, threads=4)
        cfg = emulib.Config(2, br, vbr, cvbr, cx, lsb, loss, maxb)
        out = np.zeros((n, 1280), np.uint8)
        lens = np.zeros(n, np.int32)
        r2 = np.zeros(n, np.uint32)
        st = emulib.fresh_states(n // fps) if fps > 1 else None
        pcmc = np.ascontiguousarray(pcm)
        if lane:
            emu.emu_lane_set_slot(int(rng.integers(0, 64)))
        assert enc_fn(C.byref(cfg), p(st) if st is not None else None, p(pcmc), n, fps, p(out), 1280, p(lens), p(r2)) == 0
        ec.assert_packets_equal(out, lens, r2, pk, ln, rg, str(what))
        if (ln > 1).all():
            pkc, lnc = np.ascontiguousarray(pk), np.ascontiguousarray(ln.astype(np.int32))
            want, wr, wret = gm.ref_decode(pkc, lnc, fps)
            got = np.zeros((n, 960, 2), np.int16)
            gr = np.zeros(n, np.uint32)
            gret = np.zeros(n, np.int32)
            assert dec_fn(p(pkc), pkc.shape[1], p(lnc), n, fps, p(got), p(gr), p(gret)) == 0
            assert np.array_equal(gret, wret) and np.array_equal(gr, wr) and np.array_equal(got, want), what
