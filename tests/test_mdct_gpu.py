"""GPU parity tests for the MDCT path, through the C-ABI (libopusgpu.so), bit-exact against
 (a) the committed golden vectors made from the compiled reference,
 (b) the CPU oracle on seeded inputs (edge values included),
 (c) the compiled reference itself when oracle/_ref/libopus_ref.so travelled with the snapshot,
and, at BASELINE config #2's full size (4096 stereo frames), through size-independent properties."""
import ctypes as C
import os

import numpy as np
import pytest

import oraclelib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ca():
    import torch
    assert torch.cuda.is_available()
    import concentus_amd
    concentus_amd.lib.load()
    return concentus_amd


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _noise(rng, shape, full=False):
    if full:
        return rng.integers(-(1 << 31), 1 << 31, size=shape, dtype=np.int64).astype(np.int32)
    return (rng.integers(-16384, 16384, size=shape, dtype=np.int64) * 4096).astype(np.int32)


@pytest.mark.parametrize("shift", [0, 3])
def test_batch_matches_golden(ca, shift):
    import torch
    g = np.load(os.path.join(GOLD, "mdct_golden.npz"))
    freq = ca.mdct_forward_batch(_dev(g["sig"]), shift=shift)
    rec = _dev(g["prev"])
    ca.mdct_backward_batch(freq, rec, shift=shift)
    torch.cuda.synchronize()
    assert np.array_equal(freq.cpu().numpy(), g["freq_shift%d" % shift])
    assert np.array_equal(rec.cpu().numpy(), g["rec_shift%d" % shift])


@pytest.mark.parametrize("shift", [0, 3])
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("nframes,full", [(1, False), (3, True), (257, False), (1500, True)])
def test_batch_matches_oracle(ca, shift, channels, nframes, full):
    import torch
    orc = oraclelib.lib()
    rng = np.random.default_rng(1000 * shift + 10 * nframes + channels)
    sig = _noise(rng, (nframes, channels, 1080), full)
    if nframes == 3:   # edge values: zeros, extremes
        sig[0] = 0
        sig[1, :, ::2] = np.int32(2**31 - 1)
        sig[1, :, 1::2] = np.int32(-2**31)
    exp = np.zeros((nframes, channels, 960), np.int32)
    orc.orc_mdct_forward_batch(oraclelib.ptr(sig), oraclelib.ptr(exp), nframes, channels, shift)
    d_sig = _dev(sig)
    got = ca.mdct_forward_batch(d_sig, shift=shift)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), exp)
    assert np.array_equal(d_sig.cpu().numpy(), sig)        # forward does not trash its input
    prev = _noise(rng, (nframes, channels, 1080), full)
    rec_exp = prev.copy()
    orc.orc_mdct_backward_batch(oraclelib.ptr(exp), oraclelib.ptr(rec_exp), nframes, channels, shift)
    d_rec = _dev(prev)
    ca.mdct_backward_batch(got, d_rec, shift=shift)
    torch.cuda.synchronize()
    r = d_rec.cpu().numpy()
    assert np.array_equal(r, rec_exp)
    assert np.array_equal(r[..., 1020:], prev[..., 1020:])  # tail left untouched, as the reference


def test_empty_batch_and_bad_args(ca):
    import torch
    z = torch.zeros((0, 2, 1080), dtype=torch.int32, device="cuda")
    out = ca.mdct_forward_batch(z)
    assert out.shape == (0, 2, 960)
    with pytest.raises(ValueError):
        ca.mdct_forward_batch(torch.zeros((2, 3, 1080), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        ca.mdct_forward_batch(torch.zeros((2, 2, 1080), dtype=torch.int16, device="cuda"))
    with pytest.raises(ca.lib.OpusGpuError) as e:
        ca.mdct_forward_batch(torch.zeros((2, 2, 1080), dtype=torch.int32, device="cuda"), shift=1)
    assert e.value.code == -5


@pytest.mark.parametrize("shift,stride", [(0, 1), (1, 2), (2, 4), (3, 8), (3, 1)])
def test_per_call_hooks_match_oracle(ca, shift, stride):
    """The RTCD-shaped single-call entry points (host pointers, reference signature)."""
    orc = oraclelib.lib()
    rng = np.random.default_rng(50 + shift)
    n2 = 960 >> shift
    x = _noise(rng, n2 + 120)
    exp = np.zeros((n2 - 1) * stride + 1, np.int32)
    orc.orc_mdct_forward(oraclelib.ptr(x), oraclelib.ptr(exp), shift, stride)
    got = ca.clt_mdct_forward(x, shift=shift, stride=stride)
    assert np.array_equal(got, exp)
    prev = _noise(rng, n2 + 120)
    rec_exp = prev.copy()
    orc.orc_mdct_backward(oraclelib.ptr(exp), oraclelib.ptr(rec_exp), shift, stride)
    rec = prev.copy()
    ca.clt_mdct_backward(exp, rec, shift=shift, stride=stride)
    assert np.array_equal(rec, rec_exp)


def test_per_call_hook_rejects_foreign_mode(ca):
    L = ca.lib.load()

    class Head(C.Structure):
        _fields_ = [("n", C.c_int), ("maxshift", C.c_int)]
    bad = Head(2048, 3)
    buf = np.zeros(1080, np.int32)
    w = (C.c_int16 * 120)()
    L.opusgpu_clt_mdct_forward(C.byref(bad), buf.ctypes.data, buf.ctypes.data, C.addressof(w), 120, 0, 1, 0)
    assert L.opusgpu_get_last_error() == -1


def test_against_compiled_reference_if_present(ca):
    import reflib
    if not reflib.available():
        pytest.skip("oracle/_ref/libopus_ref.so did not travel")
    import torch
    ref = reflib.lib()
    m = reflib.mode()
    rng = np.random.default_rng(77)
    sig = _noise(rng, (5, 2, 1080))
    got = ca.mdct_forward_batch(_dev(sig), shift=0)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for f in range(5):
        for c in range(2):
            xin = sig[f, c].copy()
            o = np.zeros(960, np.int32)
            ref.clt_mdct_forward_c(C.byref(m.mdct), oraclelib.ptr(xin), oraclelib.ptr(o), m.window, 120, 0, 1, 0)
            assert np.array_equal(got[f, c], o)


def test_config2_full_size_properties(ca):
    """4096 stereo frames (BASELINE config #2): linearity-free integer properties that do not need the
    oracle at full size: (1) every frame of a batch made of one repeated frame gives the same result
    as that frame alone; (2) TDAC reconstruction SNR >= 60 dB (celt/tests/test_unit_mdct.c:69-129);
    (3) a checksum over the whole batch equals the oracle's on a 64-frame stride sample."""
    import torch
    orc = oraclelib.lib()
    rng = np.random.default_rng(2)
    n = 4096
    sig = _noise(rng, (n, 2, 1080))
    d = _dev(sig)
    freq = ca.mdct_forward_batch(d, shift=0)
    torch.cuda.synchronize()
    fh = freq.cpu().numpy()
    # (3) sampled exact check
    idx = np.arange(0, n, 64)
    sub = np.ascontiguousarray(sig[idx])
    exp = np.zeros((len(idx), 2, 960), np.int32)
    orc.orc_mdct_forward_batch(oraclelib.ptr(sub), oraclelib.ptr(exp), len(idx), 2, 0)
    assert np.array_equal(fh[idx], exp)
    # (1) repetition
    rep = np.broadcast_to(sig[7], (n, 2, 1080)).copy()
    fr = ca.mdct_forward_batch(_dev(rep), shift=0).cpu().numpy()
    assert np.array_equal(fr, np.broadcast_to(fh[7], (n, 2, 960)))
    # (2) TDAC: treat consecutive batch rows as consecutive frames of one long signal
    x = (rng.integers(-16384, 16384, size=(2, 120 + 960 * 64), dtype=np.int64) * 4096).astype(np.int32)
    frames = np.stack([x[:, f * 960:f * 960 + 1080] for f in range(64)])       # [64][2][1080]
    F = ca.mdct_forward_batch(_dev(frames), shift=0)
    out = np.zeros((64, 2, 960), np.int32)
    prev = torch.zeros((1, 2, 1080), dtype=torch.int32, device="cuda")
    for f in range(64):
        ca.mdct_backward_batch(F[f:f + 1].contiguous(), prev, shift=0)
        p = prev.cpu().numpy()
        out[f] = p[0, :, :960]
        prev[:, :, :120] = prev[:, :, 960:1080].clone()
    y = np.concatenate([out[f] for f in range(64)], axis=1).astype(np.float64)[:, 960:]
    r = x[:, 960:960 * 64].astype(np.float64)
    gain = float(np.sum(r * y) / np.sum(r * r))
    snr = 10 * np.log10(np.sum((gain * r) ** 2) / np.sum((y - gain * r) ** 2))
    assert snr > 60.0, snr


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_fft_batch_and_per_call_hook_match_oracle(ca, shift):
    """opus_fft_c (celt/kiss_fft.c:580): the batched kernel over 1 000 transforms (noise in celt_sig range and
    full-scale wrap-around inputs) and the per-call hook with the reference's argument list, against the oracle."""
    import torch
    orc = oraclelib.lib()
    nfft = 480 >> shift
    rng = np.random.default_rng(300 + shift)
    x = _noise(rng, (1000, nfft, 2))
    x[500:] = rng.integers(-(1 << 28), 1 << 28, size=(500, nfft, 2), dtype=np.int64).astype(np.int32)
    got = ca.fft_batch(_dev(x), shift=shift)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for t in list(range(0, 1000, 37)) + [999]:
        exp = np.zeros((nfft, 2), np.int32)
        orc.orc_fft(oraclelib.ptr(np.ascontiguousarray(x[t])), oraclelib.ptr(exp), shift)
        assert np.array_equal(got[t], exp), t
    one = ca.opus_fft(x[3], shift=shift)
    assert np.array_equal(one, got[3])


def test_fft_hook_takes_the_reference_state_objects_if_present(ca):
    """The per-call hook reads the head of the reference's own kiss_fft_state (mode->mdct.kfft[shift]) and must agree
    with opus_fft_c run on it; a foreign state is rejected."""
    import reflib
    if not reflib.available():
        pytest.skip("oracle/_ref/libopus_ref.so did not travel")
    ref = reflib.lib()
    m = reflib.mode()
    rng = np.random.default_rng(301)
    for shift in range(4):
        nfft = 480 >> shift
        x = _noise(rng, 2 * nfft)
        want = np.zeros(2 * nfft, np.int32)
        ref.opus_fft_c(m.mdct.kfft[shift], oraclelib.ptr(x), oraclelib.ptr(want))
        got = ca.opus_fft(x, shift=shift, cfg=m.mdct.kfft[shift])
        assert np.array_equal(got.ravel(), want)
    L = ca.lib.load()
    bad = ca.mdct._KissFftStateHead(512, 17476, 8, -1)
    buf = np.zeros(1024, np.int32)
    out = np.zeros(1024, np.int32)
    L.opusgpu_opus_fft(C.byref(bad), buf.ctypes.data, out.ctypes.data)
    assert L.opusgpu_get_last_error() == -1


@pytest.mark.parametrize("len_,max_pitch", [(240, 128), (480, 257), (1024, 979), (17, 3)])
def test_celt_pitch_xcorr_hook(ca, len_, max_pitch):
    """opusgpu_celt_pitch_xcorr(x, y, xcorr, len, max_pitch, arch) -- celt/pitch.c:214: wrapping 32-bit sums and
    max(1, max), against a numpy restatement and, when it travelled, the compiled reference's celt_pitch_xcorr."""
    rng = np.random.default_rng(len_ + max_pitch)
    x = rng.integers(-32768, 32768, size=len_, dtype=np.int16)
    y = rng.integers(-32768, 32768, size=len_ + max_pitch - 1, dtype=np.int16)
    L = ca.lib.load()
    got = np.zeros(max_pitch, np.int32)
    mx = L.opusgpu_celt_pitch_xcorr(x.ctypes.data, y.ctypes.data, got.ctypes.data, len_, max_pitch, 0)
    assert L.opusgpu_get_last_error() == 0
    win = np.lib.stride_tricks.sliding_window_view(y.astype(np.int64), len_)[:max_pitch]
    exp = ((win * x.astype(np.int64)).sum(1) & 0xffffffff).astype(np.uint32).view(np.int32)
    assert np.array_equal(got, exp)
    assert mx == max(1, int(exp.max()))
    import reflib
    if reflib.available():
        ref = reflib.lib()
        want = np.zeros(max_pitch, np.int32)
        ref.celt_pitch_xcorr.restype = C.c_int32
        rmx = ref.celt_pitch_xcorr(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(want.ctypes.data), len_, max_pitch, 0)
        assert np.array_equal(got, want) and mx == rmx
