#!/usr/bin/env python3
"""Generate golden vectors from the compiled reference (oracle/_ref/libopus_ref.so, i.e. the
unmodified opus-fix tree built by oracle/Makefile). Runs only where /root/reference exists; the
resulting .npz files (plain arrays, no pickles) are committed so the GPU box -- which has no
reference -- can check against them.

  mdct_golden.npz   clt_mdct_forward_c / clt_mdct_backward_c (celt/mdct.c:121,263), shifts 0 and 3,
                    in the frame layout of compute_mdcts / celt_synthesis.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import reflib  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def mdct_vectors(nframes=6, seed=2):
    ref = reflib.lib()
    m = reflib.mode()
    rng = np.random.default_rng(seed)
    out = {}
    # celt_sig-range noise (celt/tests/test_unit_mdct.c:146-154) plus one full-scale frame to exercise wrap-around
    sig = (rng.integers(-16384, 16384, size=(nframes, 2, 1080), dtype=np.int64) * 4096).astype(np.int32)
    sig[-1] = rng.integers(-(1 << 30), 1 << 30, size=(2, 1080), dtype=np.int64).astype(np.int32)
    prev = (rng.integers(-16384, 16384, size=(nframes, 2, 1080), dtype=np.int64) * 4096).astype(np.int32)
    out["sig"] = sig
    out["prev"] = prev
    for shift in (0, 3):
        B, n2 = 1 << shift, 960 >> shift
        freq = np.zeros((nframes, 2, 960), np.int32)
        rec = prev.copy()
        for f in range(nframes):
            for c in range(2):
                for b in range(B):
                    xin = np.ascontiguousarray(sig[f, c, b * n2:b * n2 + n2 + 120]).copy()
                    o = np.zeros(960, np.int32)
                    ref.clt_mdct_forward_c(C.byref(m.mdct), _p(xin), _p(o), m.window, 120, shift, B, 0)
                    freq[f, c, b::B] = o[0:960:B][:n2]
                buf = np.ascontiguousarray(rec[f, c])
                for b in range(B):
                    src = np.ascontiguousarray(freq[f, c, b:])
                    ref.clt_mdct_backward_c(C.byref(m.mdct), _p(src), C.c_void_p(buf.ctypes.data + 4 * n2 * b),
                                            m.window, 120, shift, B, 0)
                rec[f, c] = buf
        out["freq_shift%d" % shift] = freq
        out["rec_shift%d" % shift] = rec
    return out


if __name__ == "__main__":
    if not reflib.available():
        sys.exit("oracle/_ref/libopus_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    np.savez_compressed(os.path.join(HERE, "mdct_golden.npz"), **mdct_vectors())
    print("wrote mdct_golden.npz")


# ---- opus_encode() packets (BASELINE configs #1/#3) ------------------------------------------------
class _Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]


def synth_pcm(kind, nframes, seed):
    """Synthetic stereo int16 frames [n][960][2]: 'noise' = uniform in [-8192, 8191],
    'music' = sine mix + low-level noise (band-limited, exercises the pitch pre-filter and long blocks),
    'edge' = silence, full-scale square, single impulse, DC."""
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(-8192, 8192, size=(nframes, 960, 2), dtype=np.int16)
    if kind == "gmusic":
        # the reference's own encoder-test signal (tests/test_opus_encode.c:59-88 generate_music), restated in
        # oracle/oracle_testsignals.c and pinned against the reference's function by tests/test_oracle_signals.py;
        # seed 13371337 + one banner draw is what the reference's test uses
        import oraclelib
        buf = np.zeros((nframes * 960, 2), np.int16)
        oraclelib.lib().orc_generate_music(oraclelib.ptr(buf), nframes * 960, C.c_uint32(seed), 1)
        return buf.reshape(nframes, 960, 2)
    if kind == "real48":
        # real music shipped with the reference's Java test console; whole 20 ms frames only
        raw = np.fromfile(REAL48, dtype="<i2")
        n = min(nframes, raw.size // 1920)
        return np.ascontiguousarray(raw[:n * 1920].reshape(n, 960, 2))
    if kind == "music":
        t = np.arange(nframes * 960)
        f0 = 220.0 * (1 + (seed % 7))
        x = 7000 * np.sin(2 * np.pi * f0 * t / 48000) + 2500 * np.sin(2 * np.pi * 3.01 * f0 * t / 48000)
        x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t / 48000))
        pcm = np.stack([x, 0.7 * x], -1).reshape(nframes, 960, 2)
        return (pcm + rng.integers(-40, 40, size=(nframes, 960, 2))).astype(np.int16)
    pcm = np.zeros((nframes, 960, 2), np.int16)
    for n in range(nframes):
        m = n % 4
        if m == 1:
            pcm[n, :, :] = np.where((np.arange(960) // 24) % 2 == 0, 32767, -32768)[:, None]
        elif m == 2:
            pcm[n, 480 + n % 100, 0] = 30000
        elif m == 3:
            pcm[n, :, :] = 12345
    return pcm


REAL48 = "/root/reference/Java/ConcentusTestConsole/src/main/resources/AudioData/48Khz Stereo.raw"


def ref_encode(cfg, pcm, frames_per_stream, threads=4):
    drv = C.CDLL(os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "librefdrv.so"))
    n = pcm.shape[0]
    out = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rng = np.zeros(n, np.uint32)
    pcm = np.ascontiguousarray(pcm)
    drv.refdrv_encode_frames(C.byref(cfg), _p(pcm), C.c_long(n), frames_per_stream, _p(out), 1280, _p(lens), _p(rng), threads)
    return out, lens, rng


ENCODE_CASES = [
    # name, kind, nframes, frames_per_stream, seed, (bitrate, vbr, cvbr, complexity)
    ("noise_vbr_indep", "noise", 24, 1, 3, (96000, 1, 0, 10)),
    ("music_vbr_indep", "music", 24, 1, 4, (96000, 1, 0, 10)),
    ("edge_vbr_indep", "edge", 8, 1, 5, (96000, 1, 0, 10)),
    ("noise_cbr_indep", "noise", 16, 1, 6, (96000, 0, 0, 10)),
    ("music_vbr_stream", "music", 32, 16, 7, (96000, 1, 0, 10)),
    ("noise_cvbr_stream", "noise", 32, 16, 8, (64000, 1, 1, 10)),
    ("music_cbr_stream_cx5", "music", 32, 16, 9, (128000, 0, 0, 5)),
    ("noise_vbr_stream_cx0", "noise", 16, 8, 10, (48000, 1, 0, 0)),
    # 30-38.2 kb/s: the Opus layer narrows the stereo image (stereo_fade, src/opus_encoder.c:1790-1809)
    ("music_34k_cbr_stream", "music", 16, 8, 11, (34000, 0, 0, 10)),
    ("noise_33k_vbr_indep", "noise", 8, 1, 12, (33000, 1, 0, 7)),
    # the reference's own test signal (generate_music, seed 13371337) and its own audio file, as streams and as
    # independent frames; the real-audio PCM is stored once ("real48_pcm") for both cases
    ("gmusic_vbr_indep", "gmusic", 48, 1, 13371337, (96000, 1, 0, 10)),
    ("gmusic_cvbr_stream", "gmusic", 96, 48, 13371337, (64000, 1, 1, 10)),
    ("real48_vbr_stream", "real48", 873, 873, 0, (96000, 1, 0, 10)),
    ("real48_vbr_indep", "real48", 873, 1, 0, (96000, 1, 0, 10)),
]


def encode_vectors():
    out = {}
    for name, kind, n, fps, seed, (br, vbr, cvbr, cx) in ENCODE_CASES:
        cfg = _Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500)
        pcm = synth_pcm(kind, n, seed)
        pk, ln, rg = ref_encode(cfg, pcm, fps)
        assert (ln > 0).all(), name
        out[(kind if kind == "real48" else name) + "_pcm"] = pcm
        out[name + "_packets"] = pk[:, :int(ln.max())]
        out[name + "_len"] = ln
        out[name + "_rng"] = rg
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "encode_golden.npz"), **encode_vectors())
    print("wrote encode_golden.npz")


# ---- opus_decode() of packets (decoder batch) ----------------------------------------------------------
DECODE_EXTRA_CASES = [
    # low rates: band folding, noise fill, intensity stereo and anti-collapse are all exercised here
    ("dec_music_32k_stream", "music", 32, 16, 21, (32000, 1, 0, 10)),
    ("dec_noise_36k_cvbr_stream", "noise", 32, 16, 22, (36000, 1, 1, 10)),
    ("dec_edge_40k_cbr_stream", "edge", 16, 8, 23, (40000, 0, 0, 5)),
]


def ref_decode(packets, lens, frames_per_stream, threads=4):
    drv = C.CDLL(os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "librefdrv.so"))
    n = packets.shape[0]
    packets = np.ascontiguousarray(packets)
    lens = np.ascontiguousarray(lens.astype(np.int32))
    pcm = np.zeros((n, 960, 2), np.int16)
    rng = np.zeros(n, np.uint32)
    ret = np.zeros(n, np.int32)
    drv.refdrv_decode_frames(_p(packets), packets.shape[1], _p(lens), C.c_long(n), frames_per_stream, _p(pcm), _p(rng), _p(ret), threads)
    return pcm, rng, ret


def decode_vectors():
    """Expected opus_decode() output (PCM, final range) for the packets of the encode cases, each stream decoded by
    its own fresh decoder, plus a few low-rate streams encoded by the reference for the purpose."""
    out = {}
    enc = np.load(os.path.join(HERE, "encode_golden.npz"))
    for name, kind, n, fps, seed, _cfg in ENCODE_CASES:
        if kind == "real48":
            continue               # 3.3 MB of PCM per case: decoded against the live reference instead (tests/test_decode_gpu.py)
        pcm, rng, ret = ref_decode(enc[name + "_packets"], enc[name + "_len"], fps)
        assert (ret == 960).all() and (rng == enc[name + "_rng"]).all(), name
        out[name + "_dpcm"] = pcm
    for name, kind, n, fps, seed, (br, vbr, cvbr, cx) in DECODE_EXTRA_CASES:
        pk, ln, rg = ref_encode(_Cfg(2, br, vbr, cvbr, cx, 16, 0, 1500), synth_pcm(kind, n, seed), fps)
        pk = pk[:, :int(ln.max())]
        pcm, rng, ret = ref_decode(pk, ln, fps)
        assert (ret == 960).all() and (rng == rg).all(), name
        out[name + "_packets"] = pk
        out[name + "_len"] = ln
        out[name + "_rng"] = rg
        out[name + "_dpcm"] = pcm
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "decode_golden.npz"), **decode_vectors())
    print("wrote decode_golden.npz")


# ---- SILK function-boundary records (BASELINE config #4) ---------------------------------------------
def synth_voice(nsamples, seed, fs=16000):
    """Synthetic voiced/unvoiced speech-like mono int16 signal: glottal pulse train with slowly varying pitch
    through three formant resonators, alternating with noise bursts and short pauses."""
    rng = np.random.default_rng(seed)
    t = np.arange(nsamples)
    f0 = 110 + 40 * np.sin(2 * np.pi * t / (fs * 1.7)) + 15 * np.sin(2 * np.pi * t / (fs * 0.31))
    phase = np.cumsum(f0 / fs)
    pulses = (np.diff(np.floor(phase), prepend=0) > 0).astype(np.float64)
    voiced_env = (np.sin(2 * np.pi * t / (fs * 0.9)) > -0.3).astype(np.float64)
    exc = pulses * voiced_env * 8000 + rng.normal(0, 300, nsamples) * (1 - voiced_env) * 3 + rng.normal(0, 20, nsamples)
    y = exc
    for fc, bw in ((700, 130), (1220, 70), (2600, 160)):
        r = np.exp(-np.pi * bw / fs)
        a1, a2 = -2 * r * np.cos(2 * np.pi * fc / fs), r * r
        out = np.zeros(nsamples)
        for n in range(nsamples):
            out[n] = y[n] - a1 * (out[n - 1] if n > 0 else 0) - a2 * (out[n - 2] if n > 1 else 0)
        y = out * (1 - r)
    pause = ((t // (fs // 2)) % 7 == 6)
    y = np.where(pause, 0, y)
    y = y / (np.abs(y).max() + 1e-9) * 20000
    return y.astype(np.int16)


def silk_capture(pcm16k, max_records=200000, complexity=3, bitrate=32000):
    """Run the reference SILK encoder (opus_encode, OPUS_APPLICATION_VOIP, 16 kHz mono, 20 ms) over pcm16k with the
    capture variant of the library and return the recorded silk_burg_modified / silk_NSQ calls as raw byte arrays."""
    lib = C.CDLL(os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "libopus_ref_silkcap.so"))
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.refcap_start(max_records)
    err = C.c_int()
    enc = C.c_void_p(lib.opus_encoder_create(16000, 1, 2048, C.byref(err)))
    for req, v in ((4002, bitrate), (4006, 1), (4020, 0), (4010, complexity), (4012, 0), (4016, 0), (4014, 0), (4036, 16)):
        lib.opus_encoder_ctl(enc, req, v)
    out = (C.c_ubyte * 1500)()
    nfr = len(pcm16k) // 320
    for f in range(nfr):
        fr = np.ascontiguousarray(pcm16k[f * 320:(f + 1) * 320])
        r = lib.opus_encode(enc, fr.ctypes.data_as(C.c_void_p), 320, out, 1500)
        assert r > 0
    nb, nn = lib.refcap_count_burg(), lib.refcap_count_nsq()
    sz = [lib.refcap_sizes(i) for i in range(5)]
    bufs = [np.zeros((nb, sz[0]), np.uint8), np.zeros((nb, sz[1]), np.uint8), np.zeros((nn, sz[2]), np.uint8),
            np.zeros((nn, sz[3]), np.uint8), np.zeros((nn, sz[3]), np.uint8), np.zeros((nn, sz[4]), np.uint8)]
    lib.refcap_get(*[_p(b) for b in bufs])
    return dict(burg_in=bufs[0], burg_out=bufs[1], nsq_in=bufs[2], nsq_state_in=bufs[3], nsq_state_out=bufs[4], nsq_out=bufs[5])


def silk_dd_capture(pcm16k, complexity, max_records=200000, bitrate=32000):
    """As silk_capture, for silk_NSQ_del_dec (the quantizer of complexity >= 4): records of include/opusgpu_silk.h
    (opusgpu_nsq_dd_in, state in/out, opusgpu_nsq_dd_out)."""
    lib = C.CDLL(os.path.join(os.path.dirname(HERE), "..", "oracle", "_ref", "libopus_ref_silkcap.so"))
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.refcap_start_dd(max_records)
    err = C.c_int()
    enc = C.c_void_p(lib.opus_encoder_create(16000, 1, 2048, C.byref(err)))
    for req, v in ((4002, bitrate), (4006, 1), (4020, 0), (4010, complexity), (4012, 0), (4016, 0), (4014, 0), (4036, 16)):
        lib.opus_encoder_ctl(enc, req, v)
    out = (C.c_ubyte * 1500)()
    for f in range(len(pcm16k) // 320):
        fr = np.ascontiguousarray(pcm16k[f * 320:(f + 1) * 320])
        assert lib.opus_encode(enc, fr.ctypes.data_as(C.c_void_p), 320, out, 1500) > 0
    n = lib.refcap_count_dd()
    sz = [lib.refcap_sizes_dd(i) for i in range(3)]
    bufs = [np.zeros((n, sz[0]), np.uint8), np.zeros((n, sz[1]), np.uint8), np.zeros((n, sz[1]), np.uint8), np.zeros((n, sz[2]), np.uint8)]
    lib.refcap_get_dd(*[_p(b) for b in bufs])
    return dict(dd_in=bufs[0], dd_state_in=bufs[1], dd_state_out=bufs[2], dd_out=bufs[3])


def silk_dd_vectors(per_source=14):
    srcs = [synth_voice(16000 * 6, 41)]
    raw = "/root/reference/Java/ConcentusTestConsole/src/main/resources/AudioData/16Khz Mono.raw"
    if os.path.exists(raw):
        srcs.append(np.fromfile(raw, dtype="<i2")[:16000 * 6])
    parts = [silk_dd_capture(s_, cx) for cx in (5, 7, 10) for s_ in srcs]      # 2, 3 and 4 delayed-decision states
    out = {}
    for k in parts[0]:
        sel = []
        for p in parts:
            n = p[k].shape[0]
            sel.append(p[k][np.linspace(0, n - 1, per_source).astype(int)])
        out["silk_" + k] = np.concatenate(sel)
    return out


def silk_vectors():
    parts = []
    parts.append(silk_capture(synth_voice(16000 * 6, 41)))
    raw = "/root/reference/Java/ConcentusTestConsole/src/main/resources/AudioData/16Khz Mono.raw"
    if os.path.exists(raw):   # real speech shipped with the reference's Java test console (first 6 s)
        parts.append(silk_capture(np.fromfile(raw, dtype="<i2")[:16000 * 6]))
    out = {}
    for k in parts[0]:
        sel = []
        for p in parts:
            n = p[k].shape[0]
            idx = np.linspace(0, n - 1, 40).astype(int)          # 40 records per source, spread over the signal
            sel.append(p[k][idx])
        out["silk_" + k] = np.concatenate(sel)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "silk_golden.npz"), **silk_vectors())
    print("wrote silk_golden.npz")
    np.savez_compressed(os.path.join(HERE, "silk_dd_golden.npz"), **silk_dd_vectors())
    print("wrote silk_dd_golden.npz")
