#!/usr/bin/env python3
"""Debug aid: diff the range-coder symbol stream of the kernel sources (host emulation build,
tests/emu) against the reference's own EC_DIFF trace (oracle/_ref/libopus_ref_trace.so).

usage: trace_diff.py <nframes> <frames_per_stream> <seed> [noise|tone] [vbr]
Prints the first differing coder operation and the emulation's stage markers around it."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITRATE = int(os.environ.get("TD_BITRATE", 96000))      # optional overrides of the encoder settings
COMPLEXITY = int(os.environ.get("TD_COMPLEXITY", 10))
CODES = {"1f", "1g", "1h", "1j", "1k", "1l", "1m", "1n", "1p", "1q", "1r", "1s", "1t"}


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]


def make_pcm(nframes, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(-8192, 8192, size=(nframes, 960, 2), dtype=np.int16)
    t = np.arange(nframes * 960)
    x = (8000 * np.sin(2 * np.pi * 440 * t / 48000) + 2000 * np.sin(2 * np.pi * 3000 * t / 48000)).astype(np.int16)
    pcm = np.stack([x, (x * 0.7).astype(np.int16)], -1).reshape(nframes, 960, 2)
    return (pcm + rng.integers(-50, 50, size=(nframes, 960, 2), dtype=np.int16)).astype(np.int16)


def child_ref(nframes, fps, seed, kind, vbr):
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libopus_ref_trace.so"))
    lib.opus_encoder_create.restype = C.c_void_p
    lib.opus_encoder_ctl.argtypes = [C.c_void_p, C.c_int, C.c_int]
    pcm = make_pcm(nframes, seed, kind)
    out = (C.c_ubyte * 1500)()
    for s in range(nframes // fps):
        err = C.c_int()
        enc = C.c_void_p(lib.opus_encoder_create(48000, 2, 2051, C.byref(err)))
        for req, v in ((4002, BITRATE), (4008, -1000), (4006, vbr), (4020, 0), (4010, COMPLEXITY), (4012, 0), (4022, -1000), (4016, 0), (4014, 0), (4036, 16), (4040, 5000)):
            lib.opus_encoder_ctl(enc, req, v)
        for f in range(fps):
            n = s * fps + f
            sys.stdout.flush()
            os.write(1, ("FRAME %d\n" % n).encode())
            fr = np.ascontiguousarray(pcm[n])
            lib.opus_encode(enc, fr.ctypes.data_as(C.c_void_p), 960, out, 1500)
            lib.fflush(None)


def child_emu(nframes, fps, seed, kind, vbr):
    emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libcelt_emu_trace.so"))
    pcm = make_pcm(nframes, seed, kind)
    cfg = Cfg(2, BITRATE, vbr, 0, COMPLEXITY, 16, 0, 1500)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    st = emulib.fresh_states(nframes // fps) if fps > 1 else None
    o = np.zeros((nframes, 1280), np.uint8)
    l = np.zeros(nframes, np.int32)
    r = np.zeros(nframes, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    for n in range(nframes):
        os.write(2, ("FRAME %d\n" % n).encode())
        sp = C.c_void_p(st.ctypes.data + (n // fps) * st.shape[1]) if st is not None else None
        emu.emu_celt_encode_frames(C.byref(cfg), sp, p(pcm[n:n + 1].copy()), 1, 1, p(o[n:n + 1]), 1280, p(l[n:n + 1]), p(r[n:n + 1]))


def parse(text):
    ops, ctx = [], []
    for line in text.splitlines():
        m = re.match(r"^([0-9][a-z]+) 0x([0-9a-f]+)$", line.strip())
        if m and m.group(1) in CODES:
            ops.append((m.group(1), int(m.group(2), 16), len(ctx)))
        elif line.startswith("FRAME") or (line and not re.match(r"^[0-9][a-z]+ ", line) and "ec_ctx" not in line):
            ctx.append(line.strip())
    return ops, ctx


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        which, a = sys.argv[2], sys.argv[3:]
        args = (int(a[0]), int(a[1]), int(a[2]), a[3], int(a[4]))
        (child_ref if which == "ref" else child_emu)(*args)
        sys.exit(0)
    a = sys.argv[1:] + ["noise", "1"][len(sys.argv) - 4:]
    base = [sys.executable, __file__, "--child"]
    ref = subprocess.run(base + ["ref"] + a, capture_output=True, text=True, timeout=120)
    emu = subprocess.run(base + ["emu"] + a, capture_output=True, text=True, timeout=120)
    if emu.returncode:
        print(emu.stderr[-2000:])
    ro, _ = parse(ref.stdout)
    eo, ectx = parse(emu.stderr)
    print("ref ops", len(ro), "emu ops", len(eo))
    for i, (x, y) in enumerate(zip(ro, eo)):
        if x[:2] != y[:2]:
            print("first difference at op", i, "ref", x[:2], "emu", y[:2])
            print("emu context:", ectx[max(0, y[2] - 4):y[2] + 1])
            print("ref ops around:", [(c, hex(v)) for c, v, _ in ro[max(0, i - 6):i + 6]])
            print("emu ops around:", [(c, hex(v)) for c, v, _ in eo[max(0, i - 6):i + 6]])
            break
    else:
        print("all common ops equal")
