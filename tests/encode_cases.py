"""Shared helpers for the encoder parity tests (CPU emulation tier and GPU tier)."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "encode_golden.npz")

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))


def golden_module():
    m = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(m)
    return m


def cases():
    return golden_module().ENCODE_CASES


def load_case(name):
    g = np.load(GOLD)
    pcm_key = name + "_pcm"
    if pcm_key not in g.files:          # cases that share one input (the real-audio file) store it once under their kind
        pcm_key = [c[1] for c in cases() if c[0] == name][0] + "_pcm"
    return g[pcm_key], g[name + "_packets"], g[name + "_len"], g[name + "_rng"]


def assert_packets_equal(got_pk, got_len, got_rng, exp_pk, exp_len, exp_rng, what=""):
    """Byte-for-byte packet equality + final range, the reference's own parity criterion
    (CSharp/ParityTest/TestDriver.cs:227-250, tests/test_opus_encode.c:305-306)."""
    got_len = np.asarray(got_len)
    assert np.array_equal(got_len, exp_len), "%s: packet lengths differ at %s" % (what, np.nonzero(got_len != exp_len)[0][:8])
    assert np.array_equal(np.asarray(got_rng).astype(np.uint32), exp_rng), "%s: final range differs" % what
    for n in range(len(exp_len)):
        L = int(exp_len[n])
        if not np.array_equal(got_pk[n, :L], exp_pk[n, :L]):
            d = np.nonzero(got_pk[n, :L] != exp_pk[n, :L])[0]
            raise AssertionError("%s: frame %d differs first at byte %d of %d" % (what, n, d[0], L))


def assert_packets_equal_fast(got_pk, got_len, got_rng, exp_pk, exp_len, exp_rng, what=""):
    """The same criterion for large batches: vectorised, the per-frame loop only runs to report a mismatch."""
    got_len = np.asarray(got_len)
    assert np.array_equal(got_len, exp_len), "%s: packet lengths differ at %s" % (what, np.nonzero(got_len != exp_len)[0][:8])
    assert np.array_equal(np.asarray(got_rng).astype(np.uint32), exp_rng), "%s: final range differs at %s" % (
        what, np.nonzero(np.asarray(got_rng).astype(np.uint32) != exp_rng)[0][:8])
    w = int(exp_len.max())
    mask = np.arange(w)[None, :] < np.asarray(exp_len)[:, None]
    bad = np.nonzero(((got_pk[:, :w] != exp_pk[:, :w]) & mask).any(1))[0]
    assert bad.size == 0, "%s: %d of %d packets differ, first %s" % (what, bad.size, len(exp_len), bad[:8])
