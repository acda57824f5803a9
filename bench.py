#!/usr/bin/env python3
"""bench.py -- throughput of the batched Opus frame path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mdct] [--frames F]

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM.
Workload `mdct` is BASELINE.json configs[1]: 4 096 independent 48 kHz stereo 20 ms frames,
clt_mdct_forward + clt_mdct_backward (33 600 algorithmic bytes per stereo frame).
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank processes its own shard of
F frames -- the frame corpus partitions with no data-path collective -- so scaling is "weak" and
`value` is the whole-job frames/s.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
timed with events on the launch stream) and `cpu_baseline` (the reference's own C code, or the
oracle port, timed on this box's host cores over a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_FWD = 2 * 1080 * 4 + 2 * 960 * 4            # 16 320 B / stereo frame  
BYTES_BWD = 2 * 960 * 4 + 2 * 1080 * 4 + 2 * 120 * 4   # 17 280 B / stereo frame
BYTES_FRAME = BYTES_FWD + BYTES_BWD               # 33 600 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="mdct", choices=["mdct"])
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline_mdct(seed_sig):
    """Time the reference's clt_mdct_forward_c + clt_mdct_backward_c (oracle/_ref, kind "reference")
    or, if that library did not travel, our C restatement (kind "port") on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)                    # one GPU's share of the host (the box allots 16 per GPU)
    n = 2048                                  # stereo frames per pass
    sig = np.ascontiguousarray(np.tile(seed_sig, (n // seed_sig.shape[0] + 1, 1, 1))[:n])
    freq = np.zeros((n, 2, 960), np.int32)
    rec = sig.copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    refdrv = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    if os.path.exists(refdrv):
        drv = C.CDLL(refdrv)
        drv.refdrv_mdct_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]

        def run(threads):
            drv.refdrv_mdct_batch(p(sig), p(freq), p(rec), n * 2, 0, 3, threads)
        kind = "reference"
    else:
        import oraclelib
        orc = oraclelib.lib()

        def run(threads):
            orc.orc_mdct_forward_batch(p(sig), p(freq), n, 2, 0)
            orc.orc_mdct_backward_batch(p(freq), p(rec), n, 2, 0)
        kind = "port"
        cores = 1
    run(1)
    t0 = time.perf_counter()
    reps1 = 0
    while time.perf_counter() - t0 < 4.0:
        run(1)
        reps1 += 1
    one = reps1 * n / (time.perf_counter() - t0)
    multi = one
    if cores > 1:
        run(cores)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            run(cores)
            reps += 1
        multi = reps * n / (time.perf_counter() - t0)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": "%d stereo frames x clt_mdct_forward_c+clt_mdct_backward_c (shift 0) per pass, "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import concentus_amd as ca
    ca.lib.load()

    F = a.frames
    rng = np.random.default_rng(2 + rank)      # seed 2, int16 uniform x 4096 (Q12)
    host = (rng.integers(-16384, 16384, size=(F, 2, 1080), dtype=np.int64) * 4096).astype(np.int32)
    sig = torch.from_numpy(host).to(dev)
    freq = torch.empty((F, 2, 960), dtype=torch.int32, device=dev)
    rec = sig.clone()

    def step():
        ca.mdct_forward_batch(sig, freq, shift=0)
        ca.mdct_backward_batch(freq, rec, shift=0)

    for _ in range(a.warmup):
        step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()
        ca.mdct_forward_batch(sig, freq, shift=0)
        ev[k][1].record()
        ca.mdct_backward_batch(freq, rec, shift=0)
        ev[k][2].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    bwd_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    if fwd_ms >= bwd_ms:
        kname, kbytes, kms = "mdct_forward_kernel<0>", BYTES_FWD * F, fwd_ms
    else:
        kname, kbytes, kms = "mdct_backward_kernel<0>", BYTES_BWD * F, bwd_ms
    achieved = kbytes / (kms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(kname)
            except Exception:
                traffic = None
        out = {
            "metric": "48kHz stereo 20ms CELT frames/sec (clt_mdct_forward+backward, config #2)",
            "value": round(F * world * a.steps / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d independent 48 kHz stereo 20 ms frames per GPU, "
                                   "clt_mdct_forward+backward only (shift 0), bit-exact vs FIXED_POINT" % F,
                       "frames_per_gpu": F, "channels": 2, "sharding": "frames block-partitioned, no collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": round(kms, 5),
                         "other_kernel_ms": round(bwd_ms if kname.startswith("mdct_forward") else fwd_ms, 5),
                         "whole_step_GBps": round(BYTES_FRAME * F / ((fwd_ms + bwd_ms) * 1e-3) / 1e9, 1)},
        }
        if not a.no_cpu_baseline and world >= 1:
            try:
                out["cpu_baseline"] = cpu_baseline_mdct(host[:256])
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
