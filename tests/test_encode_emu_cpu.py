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


@pytest.mark.ref
@pytest.mark.parametrize("mdb,vbr,cvbr", [(1500, 1, 0), (400, 1, 0), (400, 1, 1), (400, 0, 0), (1276, 0, 0)])
def test_bitrate_max_resolves_per_call_like_the_reference(mdb, vbr, cvbr):
    """OPUS_SET_BITRATE(OPUS_BITRATE_MAX): the reference resolves it at every opus_encode() call as
    IMIN(1276, max_data_bytes) * 8 * Fs / frame_size (user_bitrate_to_bitrate, src/opus_encoder.c:512-521, :1040-1050) --
    510 400 b/s for a 1276-byte buffer, 160 000 for 400 bytes. The host side (OpusEncoderBatch / opusgpu_encode) hands the
    kernels that number; here the kernel sources run with it against the reference run with -1."""
    gm = ec.golden_module()
    pcm = gm.synth_pcm("music", 32, 77)
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, -1, vbr, cvbr, 10, 16, 0, mdb), pcm, 16)
    emu = emulib.lib()
    cfg = emulib.Config(2, min(1276, mdb) * 400, vbr, cvbr, 10, 16, 0, mdb)
    out = np.zeros((32, 1280), np.uint8)
    lens = np.zeros(32, np.int32)
    rng = np.zeros(32, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    st = emulib.fresh_states(2)
    emu.emu_celt_encode_frames(C.byref(cfg), p(st), p(np.ascontiguousarray(pcm)), 32, 16, p(out), 1280, p(lens), p(rng))
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, "OPUS_BITRATE_MAX, max_data_bytes %d" % mdb)
