"""GPU parity tests for the batched CELT frame encoder, through the C-ABI (opusgpu_encode_batch):
packets + final range byte-exact against (a) committed golden vectors from the compiled reference,
(b) the compiled reference itself, live, when oracle/_ref travelled with the snapshot,
(c) at BASELINE config #3's full size (65 536 frames) through batch-size independence and a sampled
exact check."""
import ctypes as C
import os

import numpy as np
import pytest

import encode_cases as ec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
THREADS = min(16, len(os.sched_getaffinity(0)))


@pytest.fixture(scope="module")
def ca():
    import torch
    assert torch.cuda.is_available()
    import concentus_amd
    concentus_amd.lib.load()
    return concentus_amd


def _cfg(ca, vals):
    br, vbr, cvbr, cx = vals
    return ca.CeltConfig(2, br, vbr, cvbr, cx, 16, 0, 1500)


def _gpu_encode(ca, pcm, fps, cfgvals):
    import torch
    cfg = _cfg(ca, cfgvals)
    n = pcm.shape[0]
    if fps == 1:
        out, lens, rng = ca.encode_independent(torch.from_numpy(np.ascontiguousarray(pcm)).cuda(), cfg)
        torch.cuda.synchronize()
        return out.cpu().numpy(), lens.cpu().numpy(), rng.cpu().numpy().view(np.uint32)
    ns = n // fps
    enc = ca.OpusEncoderBatch(ns).apply_opus_demo_ctls(*cfgvals)
    outs = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rngs = np.zeros(n, np.uint32)
    p4 = pcm.reshape(ns, fps, 960, 2)
    for f in range(fps):
        o, l = enc.encode(torch.from_numpy(np.ascontiguousarray(p4[:, f])).cuda())
        torch.cuda.synchronize()
        idx = np.arange(ns) * fps + f
        o = o.cpu().numpy()
        outs[idx, :o.shape[1]] = o
        lens[idx] = l.cpu().numpy()
        rngs[idx] = enc.ctl(4031).cpu().numpy().view(np.uint32)
    return outs, lens, rngs


@pytest.mark.parametrize("case", ec.cases(), ids=lambda c: c[0])
def test_gpu_matches_golden_packets(ca, case):
    name, _kind, _n, fps, _seed, cfgvals = case
    pcm, pk, ln, rg = ec.load_case(name)
    out, lens, rng = _gpu_encode(ca, pcm, fps, cfgvals)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, name)


@pytest.mark.parametrize("kind,n,fps,cfgvals", [
    ("noise", 512, 1, (96000, 1, 0, 10)),
    ("music", 512, 1, (96000, 1, 0, 10)),
    ("music", 384, 24, (96000, 1, 0, 10)),
    ("noise", 256, 16, (96000, 0, 0, 10)),
    ("edge", 64, 1, (96000, 1, 0, 10)),
    ("music", 256, 16, (40000, 1, 1, 7)),
])
def test_gpu_matches_live_reference(ca, kind, n, fps, cfgvals):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref did not travel")
    gm = ec.golden_module()
    pcm = gm.synth_pcm(kind, n, 4321 + n + fps)
    br, vbr, cvbr, cx = cfgvals
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500), pcm, fps, threads=8)
    out, lens, rng = _gpu_encode(ca, pcm, fps, cfgvals)
    ec.assert_packets_equal(out, lens, rng, pk, ln, rg, "%s n=%d fps=%d %r" % (kind, n, fps, cfgvals))


def test_config3_full_size_65536_frames(ca):
    """65 536 independent frames, 96 kb/s, complexity 10 (BASELINE config #3): EVERY packet and final range against
    the compiled reference (oracle/_ref, ~1 s on the box's host threads) -- small fixtures do not see inter-wave bugs --
    and the result of a frame must not depend on the batch it was encoded in."""
    import torch
    gm = ec.golden_module()
    n = 65536
    rng = np.random.default_rng(3)
    pcm = rng.integers(-8192, 8192, size=(n, 960, 2), dtype=np.int16)
    d = torch.from_numpy(pcm).cuda()
    cfg = ca.default_config()
    out, lens, fr = ca.encode_independent(d, cfg)
    torch.cuda.synchronize()
    lens_h = lens.cpu().numpy()
    assert (lens_h > 2).all() and (lens_h <= 1276).all()
    idx = np.arange(0, n, 128)
    sub = torch.from_numpy(np.ascontiguousarray(pcm[idx])).cuda()
    o2, l2, r2 = ca.encode_independent(sub, cfg)
    torch.cuda.synchronize()
    out_h = out.cpu().numpy()
    fr_h = fr.cpu().numpy().view(np.uint32)
    assert np.array_equal(l2.cpu().numpy(), lens_h[idx])
    assert np.array_equal(r2.cpu().numpy().view(np.uint32), fr_h[idx])
    assert np.array_equal(o2.cpu().numpy(), out_h[idx])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref did not travel: batch-size independence only")
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, 96000, 1, 0, 10, 16, 0, 1500), pcm, 1, threads=THREADS)
    ec.assert_packets_equal_fast(out_h, lens_h, fr_h, pk, ln, rg, "config3, all 65 536 frames")


def test_config5_mixed_shard_on_one_gpu(ca):
    """One rank's shard of BASELINE configs[4] (1 M mixed units over 8 GPUs -> 131 072 per GPU = 114 688 CELT frames +
    16 384 SILK records, sharding.mixed_counts), run the way bench.py --workload mixed runs it: the CELT pipeline on the
    current stream, silk_burg_modified + silk_NSQ on a side stream. CELT: every 8th frame against the compiled reference;
    SILK: every record against the outputs the reference produced when the records were captured."""
    import torch
    from concentus_amd.sharding import mixed_counts
    import silk_corpus
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")) or not silk_corpus.available():
        pytest.skip("oracle/_ref did not travel")
    gm = ec.golden_module()
    F, NS = mixed_counts(131072)
    assert (F, NS) == (114688, 16384)
    rec = silk_corpus.corpus(NS, "nsq")
    rng = np.random.default_rng(5)
    pcm = rng.integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)
    d = torch.from_numpy(pcm).cuda()
    bi, ni, st = (torch.from_numpy(np.array(rec[k])).cuda() for k in ("burg_in", "nsq_in", "nsq_state_in"))
    side = torch.cuda.Stream()
    for _ in range(2):                      # twice: the second pass runs with both workspaces already allocated
        st.copy_(torch.from_numpy(np.array(rec["nsq_state_in"])).cuda())
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            bo = ca.silk_burg_modified(bi)
            pulses = ca.silk_NSQ(ni, st)
        out, lens, fr = ca.encode_independent(d, ca.default_config())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    assert np.array_equal(bo.cpu().numpy(), rec["burg_out"])
    assert np.array_equal(pulses.cpu().numpy().view(np.uint8), rec["nsq_out"])
    assert np.array_equal(st.cpu().numpy(), rec["nsq_state_out"])
    idx = np.arange(0, F, 8)
    t_idx = torch.from_numpy(idx).cuda()
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, 96000, 1, 0, 10, 16, 0, 1500), pcm[idx], 1, threads=THREADS)
    ec.assert_packets_equal_fast(out[t_idx].cpu().numpy(), lens[t_idx].cpu().numpy(), fr[t_idx].cpu().numpy().view(np.uint32),
                                 pk, ln, rg, "config5 shard, CELT sample")


@pytest.mark.parametrize("mdb", [1500, 400])
def test_bitrate_max_follows_the_buffer_of_each_call(ca, mdb):
    """OPUS_SET_BITRATE(OPUS_BITRATE_MAX) through both host mirrors (OpusEncoderBatch.ctl and the C verbs
    opusgpu_encoder_ctl / opusgpu_encode) against the live reference driven with -1: user_bitrate_to_bitrate
    (src/opus_encoder.c:512-521) resolves it per call from max_data_bytes."""
    import torch
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        pytest.skip("oracle/_ref did not travel")
    gm = ec.golden_module()
    ns, fps = 4, 6
    pcm = gm.synth_pcm("music", ns * fps, 91)
    pk, ln, rg = gm.ref_encode(gm._Cfg(2, -1, 1, 0, 10, 16, 0, mdb), pcm, fps, threads=2)
    enc = ca.OpusEncoderBatch(ns).apply_opus_demo_ctls(-1, 1, 0, 10)
    p4 = pcm.reshape(ns, fps, 960, 2)
    for f in range(fps):
        o, l = enc.encode(torch.from_numpy(np.ascontiguousarray(p4[:, f])).cuda(), max_data_bytes=mdb)
        torch.cuda.synchronize()
        idx = np.arange(ns) * fps + f
        ec.assert_packets_equal(o.cpu().numpy(), l.cpu().numpy(), enc.ctl(4031).cpu().numpy().view(np.uint32),
                                pk[idx], ln[idx], rg[idx], "OpusEncoderBatch, BITRATE_MAX, frame %d" % f)
    L = ca.lib.load()
    err = C.c_int(0)
    st = L.opusgpu_encoder_create(48000, 2, 2051, C.byref(err))
    for req, v in ((4002, -1), (4006, 1), (4020, 0), (4010, 10), (4036, 16)):
        assert L.opusgpu_encoder_ctl(C.c_void_p(st), req, C.c_int32(v)) == 0
    data = (C.c_ubyte * 1500)()
    for f in range(fps):
        frame = np.ascontiguousarray(pcm[f])
        got = L.opusgpu_encode(C.c_void_p(st), frame.ctypes.data_as(C.c_void_p), 960, data, mdb)
        assert got == ln[f] and bytes(data[:got]) == pk[f, :got].tobytes(), f
    L.opusgpu_encoder_destroy(C.c_void_p(st))


def test_bad_configs_rejected(ca):
    import torch
    pcm = torch.zeros((2, 960, 2), dtype=torch.int16, device="cuda")
    with pytest.raises(ca.lib.OpusGpuError) as e:
        ca.encode_independent(pcm, ca.CeltConfig(2, 16000, 1, 0, 10, 16, 0, 1500))
    assert e.value.code == -5
    with pytest.raises(ca.lib.OpusGpuError) as e:
        ca.encode_independent(pcm, ca.CeltConfig(2, 96000, 1, 0, 11, 16, 0, 1500))
    assert e.value.code == -1
    with pytest.raises(ca.lib.OpusGpuError) as e:          # CBR into 60-byte packets = 24 kb/s: the Opus layer would go mono
        ca.encode_independent(pcm, ca.CeltConfig(2, 96000, 0, 0, 10, 16, 0, 60))
    assert e.value.code == -5
    ca.encode_independent(pcm, ca.CeltConfig(2, 96000, 1, 0, 10, 16, 0, 60))      # VBR with the same cap is inside the region
    with pytest.raises(ValueError):
        ca.encode_independent(torch.zeros((2, 480, 2), dtype=torch.int16, device="cuda"))
    out, lens, _ = ca.encode_independent(torch.zeros((0, 960, 2), dtype=torch.int16, device="cuda"))
    assert out.shape[0] == 0


def test_wave_per_frame_back_kernel_agrees_with_lane_per_frame(ca, monkeypatch):
    """The library ships two mappings of the back phase (one lane per frame: default; one wave per frame:
    OPUSGPU_BACK_WAVE=1, also the build the stage-stamp diagnostics use). Same sources, same packets."""
    gm = ec.golden_module()
    pcm = gm.synth_pcm("music", 1024, 99)
    a = _gpu_encode(ca, pcm, 1, (96000, 1, 0, 10))
    monkeypatch.setenv("OPUSGPU_BACK_WAVE", "1")
    b = _gpu_encode(ca, pcm, 1, (96000, 1, 0, 10))
    monkeypatch.delenv("OPUSGPU_BACK_WAVE")
    ec.assert_packets_equal(a[0], a[1], a[2], b[0], b[1], b[2], "lane vs wave back kernel")
    pk, ln, rg = ec.load_case("music_vbr_indep")[1:]
    monkeypatch.setenv("OPUSGPU_BACK_WAVE", "1")
    w = _gpu_encode(ca, ec.load_case("music_vbr_indep")[0], 1, (96000, 1, 0, 10))
    ec.assert_packets_equal(w[0], w[1], w[2], pk, ln, rg, "wave back kernel vs golden")


def test_frames_per_wavefront_of_the_lane_kernels(ca, monkeypatch):
    """The lane-per-frame kernels (encoder back phase, decoder front) run 64 or 32 frames per wavefront
    (OPUSGPU_LANE_FRAMES; at 32 two half-filled waves share the 64-frame workgroup). Same packets, same PCM; the batch
    size 1000 leaves a ragged last workgroup."""
    import torch
    gm = ec.golden_module()
    pcm = gm.synth_pcm("music", 1000, 5)
    res = {}
    for a in (64, 32):
        monkeypatch.setenv("OPUSGPU_LANE_FRAMES", str(a))
        pk, ln, rg = _gpu_encode(ca, pcm, 1, (64000, 1, 1, 10))
        dpcm, ret, drng = ca.decode_independent(torch.from_numpy(pk).cuda(), torch.from_numpy(ln).cuda())
        res[a] = (pk, ln, rg, dpcm.cpu().numpy(), ret.cpu().numpy(), drng.cpu().numpy().view(np.uint32))
    monkeypatch.delenv("OPUSGPU_LANE_FRAMES")
    for a in (64,):
        ec.assert_packets_equal(*res[a][:3], *res[32][:3], "lane frames %d vs 32" % a)
        assert np.array_equal(res[a][3], res[32][3]) and np.array_equal(res[a][5], res[32][5])
    assert (res[32][4] == 960).all() and np.array_equal(res[32][5], res[32][2])
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")):
        want = gm.ref_encode(gm._Cfg(2, 64000, 1, 1, 10, 16, 0, 1500), pcm, 1, threads=8)
        ec.assert_packets_equal(*res[32][:3], *want, "lane frames 32 vs reference")


def test_transient_kernels_agree(ca, monkeypatch):
    """The transient metric runs tiled through LDS (default) or as the streaming lane kernel
    (OPUSGPU_TRANSIENT_LANE=1): same arithmetic, same packets -- noise (every frame transient), music (few) and edge
    frames, with a ragged last tile of rows (1 001 frames = 2 002 rows)."""
    gm = ec.golden_module()
    for kind, n, seed in (("noise", 1001, 31), ("music", 515, 32), ("edge", 64, 33)):
        pcm = gm.synth_pcm(kind, n, seed)
        a = _gpu_encode(ca, pcm, 1, (96000, 1, 0, 10))
        monkeypatch.setenv("OPUSGPU_TRANSIENT_LANE", "1")
        b = _gpu_encode(ca, pcm, 1, (96000, 1, 0, 10))
        monkeypatch.delenv("OPUSGPU_TRANSIENT_LANE")
        ec.assert_packets_equal(a[0], a[1], a[2], b[0], b[1], b[2], "transient kernels, %s" % kind)


def test_small_workspace_chunks_the_batch(ca, monkeypatch):
    """A workspace smaller than the batch is legal (include/opusgpu.h): the library then runs the pipeline over
    chunks of workspace_bytes / opusgpu_encode_workspace_bytes(1) frames. Same packets, and the fused
    single-kernel front phase (OPUSGPU_FRONT_FUSED=1) agrees with the split pipeline."""
    import torch
    gm = ec.golden_module()
    pcm = gm.synth_pcm("music", 700, 7)
    ref = _gpu_encode(ca, pcm, 1, (96000, 1, 0, 10))
    cfg = _cfg(ca, (96000, 1, 0, 10))
    L = ca.lib.load()
    d = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    stride = ca.encoder.out_stride_for(cfg)
    for fused in (False, True):
        if fused:
            monkeypatch.setenv("OPUSGPU_FRONT_FUSED", "1")
        out = torch.zeros((700, stride), dtype=torch.uint8, device="cuda")
        lens = torch.empty((700,), dtype=torch.int32, device="cuda")
        rng = torch.empty((700,), dtype=torch.int32, device="cuda")
        ws = torch.empty((L.opusgpu_encode_workspace_bytes(256) + 100,), dtype=torch.uint8, device="cuda")   # 256-frame chunks: 256+256+188
        rc = L.opusgpu_encode_batch(C.byref(cfg), None, d.data_ptr(), out.data_ptr(), stride, lens.data_ptr(), rng.data_ptr(),
                                    700, ws.data_ptr(), ws.numel(), None)
        torch.cuda.synchronize()
        assert rc == 0
        ec.assert_packets_equal(out.cpu().numpy(), lens.cpu().numpy(), rng.cpu().numpy().view(np.uint32),
                                ref[0], ref[1], ref[2], "chunked, fused=%s" % fused)
    monkeypatch.delenv("OPUSGPU_FRONT_FUSED")


@pytest.mark.parametrize("name", ["music_vbr_stream", "noise_cvbr_stream", "music_cbr_stream_cx5"])
def test_single_stream_api_shim_matches_golden_stream(ca, name):
    """opusgpu_encoder_create / _ctl / opusgpu_encode / _destroy driven exactly as src/opus_demo.c:519-543,
    :740-760 drives libopus (host pointers, one frame per call): the packets and OPUS_GET_FINAL_RANGE of each
    stream of the golden case, frame by frame."""
    case = [c for c in ec.cases() if c[0] == name][0]
    _n, _kind, n, fps, _seed, (br, vbr, cvbr, cx) = case
    pcm, pk, ln, rg = ec.load_case(name)
    L = ca.lib.load()
    for s in range(n // fps):
        err = C.c_int(12345)
        st = L.opusgpu_encoder_create(48000, 2, 2051, C.byref(err))
        assert st and err.value == 0
        i32 = C.c_int32
        for req, v in ((4002, br), (4008, -1000), (4006, vbr), (4020, cvbr), (4010, cx), (4012, 0), (4022, -1000),
                       (4016, 0), (4014, 0), (4036, 16), (4040, 5000)):
            assert L.opusgpu_encoder_ctl(C.c_void_p(st), req, i32(v)) == 0, req
        data = (C.c_ubyte * 1500)()
        for f in range(fps):
            k = s * fps + f
            frame = np.ascontiguousarray(pcm[k])
            got = L.opusgpu_encode(C.c_void_p(st), frame.ctypes.data_as(C.c_void_p), 960, data, 1500)
            assert got == ln[k], (name, s, f, got, ln[k])
            assert bytes(data[:got]) == pk[k, :got].tobytes(), (name, s, f)
            fr = C.c_uint32(0)
            assert L.opusgpu_encoder_ctl(C.c_void_p(st), 4031, C.byref(fr)) == 0
            assert fr.value == rg[k]
        assert L.opusgpu_encode(C.c_void_p(st), frame.ctypes.data_as(C.c_void_p), 480, data, 1500) == -5     # legal in libopus, not here
        assert L.opusgpu_encode(C.c_void_p(st), frame.ctypes.data_as(C.c_void_p), 961, data, 1500) == -1     # OPUS_BAD_ARG
        L.opusgpu_encoder_destroy(C.c_void_p(st))
    err = C.c_int(0)
    assert not L.opusgpu_encoder_create(44100, 2, 2051, C.byref(err)) and err.value == -1
    assert not L.opusgpu_encoder_create(48000, 2, 2049, C.byref(err)) and err.value == -5
