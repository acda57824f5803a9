"""CPU tier: the LANE-PER-FRAME variant of the kernel sources (what celt_back_lane_kernel and celt_decode_lane_kernel run:
per-lane LDS columns [slot][64], typed address spaces, the per-lane working set, the leaf quantiser written for one lane) compiled
for the host (tests/emu/celt_lane_emu.cpp, CA_LANE_FRAME + CA_HOST_EMU) against the golden packets / PCM of the compiled
reference, in several columns of the LDS image; the library also checks that a lane never writes outside its own column."""
import ctypes as C
import os

import numpy as np
import pytest

import emulib
import encode_cases as ec
from test_decode_emu_cpu import decode_cases, load_decode_case

pytestmark = pytest.mark.skipif(not os.path.exists(emulib.HOST_CLANG), reason="host clang of the ROCm image not present")
p = lambda a: a.ctypes.data_as(C.c_void_p)


def run_lane_emu(pcm, fps, cfgvals, slot=0, max_data_bytes=1500, lsb_depth=16, loss=0):
    emu = emulib.lane_lib()
    br, vbr, cvbr, cx = cfgvals
    cfg = emulib.Config(2, br, vbr, cvbr, cx, lsb_depth, loss, max_data_bytes)
    n = pcm.shape[0]
    out = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rng = np.zeros(n, np.uint32)
    pcm = np.ascontiguousarray(pcm)
    emu.emu_lane_set_slot(slot)
    st = emulib.fresh_states(n // fps) if fps > 1 else None
    rc = emu.emu_lane_celt_encode_frames(C.byref(cfg), p(st) if st is not None else None, p(pcm), n, fps, p(out), 1280, p(lens), p(rng))
    assert rc == 0, "the lane wrote outside its own LDS column"
    return out, lens, rng


@pytest.mark.parametrize("case", ec.cases(), ids=lambda c: c[0])
def test_lane_build_matches_golden_packets(case):
    name, _kind, _n, fps, _seed, cfgvals = case
    pcm, pk, ln, rg = ec.load_case(name)
    out, lens, rng = run_lane_emu(pcm, fps, cfgvals, slot=(len(name) * 7) & 63)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, name)


@pytest.mark.parametrize("case", decode_cases(), ids=lambda c: c[0])
def test_lane_build_decoder_matches_reference_pcm(case):
    name, fps, from_encode = case
    pk, ln, rg, want = load_decode_case(name, from_encode)
    n = pk.shape[0]
    emu = emulib.lane_lib()
    pcm = np.zeros((n, 960, 2), np.int16)
    rng = np.zeros(n, np.uint32)
    ret = np.zeros(n, np.int32)
    emu.emu_lane_set_slot(len(name) & 63)
    assert emu.emu_lane_celt_decode_frames(p(pk), pk.shape[1], p(ln), n, fps, p(pcm), p(rng), p(ret)) == 0
    assert (ret == 960).all() and np.array_equal(rng, rg)
    assert np.array_equal(pcm, want), "PCM differs at frame %d" % int(np.nonzero((pcm != want).reshape(n, -1).any(1))[0][0])


def test_rare_lane_paths_are_reached_and_match_the_reference():
    """The paths that only the lane build has and that ordinary frames seldom take -- a leaf too wide for the column (search state in
    the frame's already-coded bins of X), the extra TF level of transient frames (band fetched twice), the intra pass of coarse
    energy winning (bytes restored from the column) or being forced -- must occur in this corpus, and every packet must equal the
    live reference's."""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref not built")
    gm = ec.golden_module()
    emu = emulib.lane_lib()
    emu.emu_lane_count.restype = C.c_long
    emu.emu_lane_count.argtypes = [C.c_char_p]
    emu.emu_lane_counts_reset()
    rng = np.random.default_rng(77)
    for kind, br, vbr, cvbr, cx, fps, n, scale in (("music", 32000, 1, 0, 10, 16, 32, 1), ("edge", 64000, 1, 0, 10, 1, 16, 1), ("noise", 40000, 0, 0, 5, 8, 32, 1),
                                                   ("music", 36000, 1, 1, 10, 16, 32, 1), ("noise", 48000, 1, 0, 10, 1, 32, 1), ("music", 96000, 1, 0, 3, 4, 16, 1),
                                                   ("noise", 33000, 0, 0, 10, 1, 32, 4), ("music", 128000, 1, 0, 10, 16, 32, 1), ("edge", 32000, 0, 0, 8, 4, 16, 1)):
        pcm = gm.synth_pcm(kind, n, int(rng.integers(1, 1 << 30)))
        if scale != 1:
            pcm = (pcm.astype(np.int32) // scale).astype(np.int16)
        # bursts make transient frames (short blocks + the extra TF level)
        pcm = pcm.copy()
        pcm[1::3, 400:520] = (pcm[1::3, 400:520].astype(np.int32) * 6).clip(-32768, 32767).astype(np.int16)
        pk, ln, rg = gm.ref_encode(gm._Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500), pcm, fps, threads=4)
        out, lens, r2 = run_lane_emu(pcm, fps, (br, vbr, cvbr, cx), slot=int(rng.integers(0, 64)))
        ec.assert_packets_equal(out, lens, r2, pk, ln, rg, "%s %d" % (kind, br))
    seen = {k: emu.emu_lane_count(k.encode()) for k in ("lane.wide_leaf", "lane.tf_extra_level", "lane.coarse_intra_restored")}
    assert all(v > 0 for v in seen.values()), seen


def test_lane_build_pulse_cache_lookups_equal_the_reference_bisection():
    """bits2pulses / the split threshold of the lane build (two rounds of independent probes, derived per-(LM, band) tables:
    celt_enc_back.h, celt_lane_tables.h) against rate.h:51-77's bisection and cache[cache[0]] on the same tables: every band,
    every LM the walk can reach (0..3; -1 for the bands wider than one bin), every budget the 16 383-clamped b can take that
    changes the answer (entries are bytes: -3..300) plus the extremes."""
    emu = emulib.lane_lib()
    for LM in (-1, 0, 1, 2, 3):
        for band in range(21):
            if LM == -1 and band < 8:
                continue                       # one-bin bands at LM -1: no cache row (index -1), never looked up (N > 2 guards)
            assert emu.emu_lane_pulse_cache_max(band, LM) == emu.emu_lane_pulse_cache_max_ref(band, LM), (band, LM)
            for bits in list(range(-3, 301)) + [1000, 16383, 16384]:
                assert emu.emu_lane_bits2pulses(band, LM, bits) == emu.emu_lane_bits2pulses_bisect(band, LM, bits), (band, LM, bits)
