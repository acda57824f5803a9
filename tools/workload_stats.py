#!/usr/bin/env python3
"""Per-frame work counts of the PVQ back-end on bench.py's config #3 input (host-emulation build,
CA_COUNT sites): leaves, vector sizes, pulses searched. Debug/analysis aid, not a product path."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import emulib  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = emulib.lib()
cfg = emulib.Config(2, 96000, 1, 0, 10, 16, 0, 1500)
pcm = np.random.default_rng(3).integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)
out = np.zeros((F, 1280), np.uint8)
lens = np.zeros(F, np.int32)
rng = np.zeros(F, np.uint32)
lib.emu_counts_reset()
p = lambda a: a.ctypes.data_as(C.c_void_p)
lib.emu_celt_encode_frames(C.byref(cfg), None, p(pcm), F, 1, p(out), 1280, p(lens), p(rng))
buf = C.create_string_buffer(1 << 16)
lib.emu_counts_dump(buf, len(buf))
print("frames %d, mean packet %.1f B" % (F, lens.mean()))
print("%-24s %12s %12s %10s" % ("counter", "events/frame", "sum/frame", "mean"))
for line in buf.value.decode().splitlines():
    n, e, s = line.split()
    e, s = int(e), int(s)
    print("%-24s %12.2f %12.2f %10.2f" % (n, e / F, s / F, s / max(e, 1)))
