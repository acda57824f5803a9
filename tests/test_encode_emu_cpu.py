"""CPU tier: the frame-kernel SOURCES (concentus_amd/csrc/celt_enc*.h), compiled for the host with
CA_HOST_EMU (one lane), against the committed golden packets made from the compiled reference.
This checks the kernel's logic bit-for-bit without a GPU; the wave-parallel execution itself is
covered by the -m gpu tier."""
import ctypes as C

import numpy as np
import pytest

import emulib
import encode_cases as ec


def _run_emu(pcm, fps, cfgvals, split=False):
    emu = emulib.lib()
    br, vbr, cvbr, cx = cfgvals
    cfg = emulib.Config(2, br, vbr, cvbr, cx, 16, 0, 1500)
    n = pcm.shape[0]
    out = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rng = np.zeros(n, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pcm = np.ascontiguousarray(pcm)
    fn = emu.emu_celt_encode_frames_split if split else emu.emu_celt_encode_frames
    if fps == 1:
        fn(C.byref(cfg), None, p(pcm), n, 1, p(out), 1280, p(lens), p(rng))
    else:
        st = emulib.fresh_states(n // fps)
        fn(C.byref(cfg), p(st), p(pcm), n, fps, p(out), 1280, p(lens), p(rng))
    return out, lens, rng


@pytest.mark.parametrize("case", ec.cases(), ids=lambda c: c[0])
def test_emulated_kernel_matches_golden_packets(case):
    name, _kind, _n, fps, _seed, cfgvals = case
    pcm, pk, ln, rg = ec.load_case(name)
    out, lens, rng = _run_emu(pcm, fps, cfgvals)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, name)


@pytest.mark.parametrize("case", ec.cases(), ids=lambda c: c[0])
def test_emulated_split_pipeline_matches_golden_packets(case):
    """dc_reject stage -> front phase 1 -> transient stage -> front phase 2 -> back phase, as the GPU library runs them"""
    name, _kind, _n, fps, _seed, cfgvals = case
    pcm, pk, ln, rg = ec.load_case(name)
    out, lens, rng = _run_emu(pcm, fps, cfgvals, split=True)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, name)


@pytest.mark.ref
def test_emulated_kernel_matches_live_reference_on_fresh_noise():
    gm = ec.golden_module()
    cfgvals = (96000, 1, 0, 10)
    pcm = gm.synth_pcm("noise", 48, 1234)
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, 96000, 1, 0, 10, 16, 0, 1500), pcm, 1)
    out, lens, rng = _run_emu(pcm, 1, cfgvals)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, "fresh noise")
