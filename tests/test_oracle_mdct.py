"""Pin the CPU restatement (oracle/oracle_mdct.c) against the compiled reference:
clt_mdct_forward_c / clt_mdct_backward_c (opus-fix/celt/mdct.c:121,263) and opus_fft_c
(celt/kiss_fft.c:580), bit-exact, all four shifts, both strides the encoder uses."""
import ctypes as C

import numpy as np
import pytest

import oraclelib
import reflib

pytestmark = pytest.mark.ref


def _sig(rng, n, full_scale=False):
    if full_scale:
        return rng.integers(-(1 << 28), 1 << 28, size=n, dtype=np.int64).astype(np.int32)
    # celt_sig range as in celt/tests/test_unit_mdct.c:146-154 (int16 noise in Q12)
    return (rng.integers(-16384, 16384, size=n, dtype=np.int64) * 4096).astype(np.int32)


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
@pytest.mark.parametrize("full", [False, True])
def test_fft_matches_reference(shift, full):
    ref, orc = reflib.lib(), oraclelib.lib()
    m = reflib.mode()
    nfft = 480 >> shift
    rng = np.random.default_rng(100 + shift)
    for _ in range(20):
        x = _sig(rng, 2 * nfft, full)
        a = np.zeros(2 * nfft, np.int32)
        b = np.zeros(2 * nfft, np.int32)
        ref.opus_fft_c(m.mdct.kfft[shift], oraclelib.ptr(x), oraclelib.ptr(a))
        orc.orc_fft(oraclelib.ptr(x), oraclelib.ptr(b), shift)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shift,stride", [(0, 1), (1, 2), (2, 4), (3, 8), (3, 1)])
@pytest.mark.parametrize("full", [False, True])
def test_mdct_forward_matches_reference(shift, stride, full):
    ref, orc = reflib.lib(), oraclelib.lib()
    m = reflib.mode()
    n2 = 960 >> shift
    rng = np.random.default_rng(200 + shift)
    for _ in range(20):
        x = _sig(rng, n2 + 120, full)
        a = np.zeros(n2 * stride, np.int32)
        b = np.zeros(n2 * stride, np.int32)
        xin = x.copy()  # reference trashes its input (mdct.h:64)
        ref.clt_mdct_forward_c(C.byref(m.mdct), oraclelib.ptr(xin), oraclelib.ptr(a), m.window, 120, shift, stride, 0)
        orc.orc_mdct_forward(oraclelib.ptr(x), oraclelib.ptr(b), shift, stride)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shift,stride", [(0, 1), (1, 2), (2, 4), (3, 8), (3, 1)])
@pytest.mark.parametrize("full", [False, True])
def test_mdct_backward_matches_reference(shift, stride, full):
    ref, orc = reflib.lib(), oraclelib.lib()
    m = reflib.mode()
    n2 = 960 >> shift
    rng = np.random.default_rng(300 + shift)
    for _ in range(20):
        x = _sig(rng, n2 * stride, full)
        prev = _sig(rng, n2 + 120, full)
        a, b = prev.copy(), prev.copy()
        ref.clt_mdct_backward_c(C.byref(m.mdct), oraclelib.ptr(x), oraclelib.ptr(a), m.window, 120, shift, stride, 0)
        orc.orc_mdct_backward(oraclelib.ptr(x), oraclelib.ptr(b), shift, stride)
        assert np.array_equal(a, b)


def test_batch_short_blocks_match_reference_compute_mdcts_layout():
    """8 short blocks, hop 120, stride 8 -- the layout of compute_mdcts (celt_encoder.c:418-461)."""
    ref, orc = reflib.lib(), oraclelib.lib()
    m = reflib.mode()
    rng = np.random.default_rng(7)
    sig = _sig(rng, 3 * 2 * 1080).reshape(3, 2, 1080)
    got = np.zeros((3, 2, 960), np.int32)
    orc.orc_mdct_forward_batch(oraclelib.ptr(sig), oraclelib.ptr(got), 3, 2, 3)
    exp = np.zeros((3, 2, 960), np.int32)
    for f in range(3):
        for c in range(2):
            for b in range(8):
                xin = np.ascontiguousarray(sig[f, c, b * 120:b * 120 + 240]).copy()
                out = np.zeros(960, np.int32)
                ref.clt_mdct_forward_c(C.byref(m.mdct), oraclelib.ptr(xin), oraclelib.ptr(out), m.window, 120, 3, 8, 0)
                exp[f, c, b::8] = out[0:960:8][:120]
    assert np.array_equal(got, exp)
    # and back
    rec = _sig(rng, 3 * 2 * 1080).reshape(3, 2, 1080)
    rec_ref = rec.copy()
    orc.orc_mdct_backward_batch(oraclelib.ptr(got), oraclelib.ptr(rec), 3, 2, 3)
    for f in range(3):
        for c in range(2):
            buf = np.ascontiguousarray(rec_ref[f, c])
            for b in range(8):
                src = np.ascontiguousarray(exp[f, c, b:])
                ref.clt_mdct_backward_c(C.byref(m.mdct), oraclelib.ptr(src),
                                        C.c_void_p(buf.ctypes.data + 4 * 120 * b), m.window, 120, 3, 8, 0)
            rec_ref[f, c] = buf
    assert np.array_equal(rec, rec_ref)
