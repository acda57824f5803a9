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
