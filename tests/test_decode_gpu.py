"""GPU parity tests for the batched CELT decoder, through the C-ABI (opusgpu_decode_batch): PCM + final range
bit-exact against (a) committed opus_decode() outputs of the compiled reference (tests/golden/decode_golden.npz),
(b) the compiled reference itself, live, on fresh packets at several rates when oracle/_ref travelled."""
import ctypes as C
import os

import numpy as np
import pytest

import encode_cases as ec
from test_decode_emu_cpu import decode_cases, load_decode_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ca():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import concentus_amd
    concentus_amd.lib.load()
    return concentus_amd


def _gpu_decode(ca, pk, ln, fps):
    """Streams of fps packets, every stream through its own decoder; returns pcm [n][960][2], rng [n], ret [n]."""
    import torch
    n = pk.shape[0]
    ns = n // fps
    dec = ca.OpusDecoderBatch(ns)
    pcm = np.zeros((n, 960, 2), np.int16)
    rng = np.zeros(n, np.uint32)
    ret = np.zeros(n, np.int32)
    p3 = pk.reshape(ns, fps, pk.shape[1])
    l2 = ln.reshape(ns, fps)
    for f in range(fps):
        o, r = dec.decode(torch.from_numpy(np.ascontiguousarray(p3[:, f])).cuda(), torch.from_numpy(np.ascontiguousarray(l2[:, f])).cuda())
        torch.cuda.synchronize()
        idx = np.arange(ns) * fps + f
        pcm[idx] = o.cpu().numpy()
        ret[idx] = r.cpu().numpy()
        rng[idx] = dec.ctl(4031).cpu().numpy().view(np.uint32)
    return pcm, rng, ret


@pytest.mark.parametrize("case", decode_cases(), ids=lambda c: c[0])
def test_gpu_decoder_matches_golden_pcm(ca, case):
    name, fps, from_encode = case
    pk, ln, rg, want = load_decode_case(name, from_encode)
    pcm, rng, ret = _gpu_decode(ca, pk, ln, fps)
    assert (ret == 960).all()
    assert np.array_equal(rng, rg)
    assert np.array_equal(pcm, want), "PCM differs at frame %d" % int(np.nonzero((pcm != want).reshape(len(ln), -1).any(1))[0][0])


@pytest.mark.parametrize("kind,n,fps,cfgvals", [
    ("music", 2048, 16, (32000, 1, 0, 10)),
    ("noise", 1024, 8, (36000, 1, 1, 10)),
    ("music", 1024, 1, (96000, 1, 0, 10)),
    ("noise", 512, 16, (256000, 0, 0, 10)),
    ("edge", 256, 8, (40000, 1, 0, 5)),
])
def test_gpu_decoder_matches_live_reference(ca, kind, n, fps, cfgvals):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref did not travel")
    gm = ec.golden_module()
    br, vbr, cvbr, cx = cfgvals
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500), gm.synth_pcm(kind, n, 555 + n + fps), fps, threads=8)
    pk = np.ascontiguousarray(pk[:, :int(ln.max()) + 3 & ~3])
    want, wrng, wret = gm.ref_decode(pk, ln, fps, threads=8)
    assert (wret == 960).all()
    pcm, rng, ret = _gpu_decode(ca, pk, ln.astype(np.int32), fps)
    assert (ret == 960).all()
    assert np.array_equal(rng, wrng)
    assert np.array_equal(pcm, want)


def test_gpu_encode_then_decode_round_trip(ca):
    """GPU encoder -> GPU decoder on 4096 independent frames: the decoder's final range equals the encoder's for
    every packet (the reference's own consistency check, tests/test_opus_encode.c:305-306)."""
    import torch
    rng = np.random.default_rng(11)
    pcm = torch.from_numpy(rng.integers(-8192, 8192, size=(4096, 960, 2), dtype=np.int16)).cuda()
    pk, ln, erng = ca.encode_independent(pcm)
    out, ret, drng = ca.decode_independent(pk, ln)
    torch.cuda.synchronize()
    assert (ret.cpu().numpy() == 960).all()
    assert np.array_equal(erng.cpu().numpy(), drng.cpu().numpy())


def test_gpu_decoder_rejects_what_it_does_not_implement(ca):
    import torch
    pk = np.zeros((3, 8), np.uint8)
    pk[0, 0] = 0x78
    pk[1, 0] = 0xFC
    pk[2, 0] = 0xFD
    ln = np.array([8, 1, 8], np.int32)
    _pcm, ret, _r = ca.decode_independent(torch.from_numpy(pk).cuda(), torch.from_numpy(ln).cuda())
    assert ret.cpu().numpy().tolist() == [-5, -5, -5]


def test_single_stream_decoder_shim_config1_plumbing(ca):
    """BASELINE config #1 through the libopus-shaped single-stream entry points: opusgpu_encoder_* encodes a stream
    frame by frame, opusgpu_decoder_* decodes it; packets, PCM and both final ranges equal the golden stream."""
    name = "music_vbr_stream"
    pcm_in, pk, ln, rg = ec.load_case(name)
    want = np.load(os.path.join(ROOT, "tests", "golden", "decode_golden.npz"))[name + "_dpcm"]
    L = ca.lib.load()
    err = C.c_int(0)
    enc = L.opusgpu_encoder_create(48000, 2, 2051, C.byref(err))
    dec = L.opusgpu_decoder_create(48000, 2, C.byref(err))
    assert enc and dec and err.value == 0
    for req, v in ((4002, 96000), (4006, 1), (4020, 0), (4010, 10), (4036, 16)):
        assert L.opusgpu_encoder_ctl(C.c_void_p(enc), req, C.c_int32(v)) == 0
    data = (C.c_ubyte * 1500)()
    out = np.zeros((960, 2), np.int16)
    for f in range(16):
        frame = np.ascontiguousarray(pcm_in[f])
        n = L.opusgpu_encode(C.c_void_p(enc), frame.ctypes.data_as(C.c_void_p), 960, data, 1500)
        assert n == ln[f] and bytes(data[:n]) == pk[f, :n].tobytes()
        got = L.opusgpu_decode(C.c_void_p(dec), data, n, out.ctypes.data_as(C.c_void_p), 960, 0)
        assert got == 960 and np.array_equal(out, want[f])
        er, dr = C.c_uint32(0), C.c_uint32(1)
        L.opusgpu_encoder_ctl(C.c_void_p(enc), 4031, C.byref(er))
        L.opusgpu_decoder_ctl(C.c_void_p(dec), 4031, C.byref(dr))
        assert er.value == dr.value == rg[f]
    assert L.opusgpu_decode(C.c_void_p(dec), None, 0, out.ctypes.data_as(C.c_void_p), 960, 0) == -5       # PLC not implemented
    L.opusgpu_encoder_destroy(C.c_void_p(enc))
    L.opusgpu_decoder_destroy(C.c_void_p(dec))


def test_gpu_decodes_bit_file_written_by_reference_cli(ca, tmp_path):
    """A .bit file produced by the reference's own opus_demo -e (oracle/_ref/opus_demo) goes through
    concentus_amd.bitstream and the GPU decoder as one stream; PCM equals what the reference CLI decodes from the same
    file (including the flush packet the CLI appends), final ranges equal the ones embedded in the file."""
    import subprocess
    import torch
    demo = os.path.join(ROOT, "oracle", "_ref", "opus_demo")
    if not os.path.exists(demo):
        pytest.skip("oracle/_ref/opus_demo did not travel")
    from concentus_amd import bitstream
    gm = ec.golden_module()
    pcm = gm.synth_pcm("music", 50, 4242)
    raw, bit, out = tmp_path / "in.pcm", tmp_path / "x.bit", tmp_path / "ref.pcm"
    pcm.astype("<i2").tofile(str(raw))
    for args in (["-e", "restricted-lowdelay", "48000", "2", "64000", str(raw), str(bit)], ["-d", "48000", "2", str(bit), str(out)]):
        r = subprocess.run([demo] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    pk, ln, rg = bitstream.read_opus_demo_bit(str(bit), stride=1500)
    want = np.fromfile(str(out), np.int16).reshape(-1, 960, 2)
    assert want.shape[0] == len(ln)
    dec = ca.OpusDecoderBatch(1)
    for k in range(len(ln)):
        o, r = dec.decode(torch.from_numpy(pk[k:k + 1]).cuda(), torch.from_numpy(ln[k:k + 1]).cuda())
        torch.cuda.synchronize()
        assert int(r.item()) == 960
        assert np.uint32(dec.ctl(4031).cpu().numpy().view(np.uint32)[0]) == rg[k]
        assert np.array_equal(o.cpu().numpy()[0], want[k]), k


def test_packet_length_past_the_row_is_rejected_per_stream(ca):
    """lens[k] > packet_stride would make stream k read the next stream's packet (the last stream: past the slab). The
    stream gets OPUS_BAD_ARG and is left alone; the others decode as usual."""
    import torch
    pcm, pk, ln, rg = ec.load_case("noise_vbr_indep")
    want = np.load(os.path.join(ROOT, "tests", "golden", "decode_golden.npz"))["noise_vbr_indep_dpcm"]
    w = int(ln.max())
    d_pk = torch.from_numpy(np.ascontiguousarray(pk[:, :w])).cuda()
    bad_ln = ln.astype(np.int32).copy()
    bad_ln[5] = w + 1
    bad_ln[len(ln) - 1] = 1276
    dpcm, ret, drng = ca.decode_independent(d_pk, torch.from_numpy(bad_ln).cuda())
    torch.cuda.synchronize()
    ret = ret.cpu().numpy()
    good = np.setdiff1d(np.arange(len(ln)), [5, len(ln) - 1])
    assert (ret[[5, len(ln) - 1]] == -1).all() and (ret[good] == 960).all()
    assert np.array_equal(dpcm.cpu().numpy()[good], want[good])
