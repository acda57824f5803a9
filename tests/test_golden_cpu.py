"""CPU tier: the oracle restatement against the committed golden vectors (made from the compiled
reference by tests/golden/make_golden.py), so parity is pinned even where the reference is absent."""
import os

import numpy as np
import pytest

import oraclelib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("shift", [0, 3])
def test_oracle_mdct_matches_golden(shift):
    g = np.load(os.path.join(GOLD, "mdct_golden.npz"))
    orc = oraclelib.lib()
    sig = np.ascontiguousarray(g["sig"])
    nf = sig.shape[0]
    freq = np.zeros((nf, 2, 960), np.int32)
    orc.orc_mdct_forward_batch(oraclelib.ptr(sig), oraclelib.ptr(freq), nf, 2, shift)
    assert np.array_equal(freq, g["freq_shift%d" % shift])
    rec = np.ascontiguousarray(g["prev"]).copy()
    orc.orc_mdct_backward_batch(oraclelib.ptr(freq), oraclelib.ptr(rec), nf, 2, shift)
    assert np.array_equal(rec, g["rec_shift%d" % shift])


def test_oracle_mdct_tdac_reconstruction():
    """Size-independent property (celt/tests/test_unit_mdct.c:69-129 checks SNR >= 60 dB): forward
    then backward over overlapping frames reconstructs the input up to the window/scale."""
    orc = oraclelib.lib()
    rng = np.random.default_rng(11)
    nfr = 6
    x = (rng.integers(-16384, 16384, size=120 + 960 * nfr, dtype=np.int64) * 4096).astype(np.int32)
    w = np.array([int(v) for v in _window()], np.float64) / 32768.0
    xw = x.astype(np.float64)
    prev = np.zeros(1080, np.int32)
    outs = []
    for f in range(nfr):
        sig = np.ascontiguousarray(x[f * 960:f * 960 + 1080])
        freq = np.zeros(960, np.int32)
        orc.orc_mdct_forward(oraclelib.ptr(sig), oraclelib.ptr(freq), 0, 1)
        orc.orc_mdct_backward(oraclelib.ptr(freq), oraclelib.ptr(prev), 0, 1)
        outs.append(prev[:960].copy())
        prev[:120] = prev[960:1080]
    y = np.concatenate(outs).astype(np.float64)
    # frame f output sample n corresponds to input sample f*960 + n (after the first frame's fade-in);
    # the fixed-point forward transform scales by scale*2^-(scale_shift): find the gain by least squares
    ref = xw[960:960 * nfr]
    got = y[960:960 * nfr]
    gain = float(np.dot(ref, got) / np.dot(ref, ref))
    err = got - gain * ref
    snr = 10 * np.log10(np.sum((gain * ref) ** 2) / np.sum(err ** 2))
    assert snr > 60.0, snr
    assert w.size == 120


def _window():
    import gen_tables
    return gen_tables.window(120)
