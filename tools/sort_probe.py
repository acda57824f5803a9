#!/usr/bin/env python3
"""Does the back lane kernel gain from homogeneous wavefronts? Encodes bench.py's synthetic batch as it comes, then the same frames
permuted so that frames with similar packet lengths (a proxy of the per-frame bit budget / pulse counts) share a wavefront, and
with the first payload bits as key (silence / post-filter / transient flags). GPU only; prints the back kernel's time per case."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import concentus_amd as ca

F = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
pcm_h = np.random.default_rng(3).integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)      # bench.py's config-#3 batch
dev = torch.device("cuda:0")
cfg = ca.default_config(2, 96000)
L = ca.lib.load()


def run(pcm, label):
    for _ in range(2):
        out, lens, rng_ = ca.encode_independent(pcm, cfg)
    torch.cuda.synchronize()
    L.opusgpu_kernel_timing_read((C.c_double * 8)(), (C.c_int * 8)(), 8)
    L.opusgpu_kernel_timing_enable(1)
    t0 = time.perf_counter()
    for _ in range(5):
        out, lens, rng_ = ca.encode_independent(pcm, cfg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ms = (C.c_double * 8)(); cnt = (C.c_int * 8)()
    L.opusgpu_kernel_timing_read(ms, cnt, 8)
    L.opusgpu_kernel_timing_enable(0)
    per = [round(ms[i] / max(cnt[i], 1), 3) for i in range(8)]
    print("%-28s step %.3f ms  kernels %s" % (label, dt * 1e3, per), flush=True)
    return out, lens


pcm = torch.from_numpy(pcm_h).to(dev)
out, lens = run(pcm, "as generated")
lens_h = lens.cpu().numpy()
first = out[:, 0].cpu().numpy()
print("packet bytes: min %d mean %.1f max %d; distinct first bytes %d" % (lens_h.min(), lens_h.mean(), lens_h.max(), len(np.unique(first))))
order = np.argsort(lens_h, kind="stable")
run(pcm[torch.from_numpy(order).to(dev)].contiguous(), "sorted by packet length")
order2 = np.lexsort((lens_h, first >> 4))
run(pcm[torch.from_numpy(order2).to(dev)].contiguous(), "by first bits, then length")
rngp = np.random.default_rng(1).permutation(F)
run(pcm[torch.from_numpy(rngp).to(dev)].contiguous(), "random permutation")
