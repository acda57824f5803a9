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
