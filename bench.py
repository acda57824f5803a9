#!/usr/bin/env python3
"""bench.py -- throughput of the batched Opus frame path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload celt|mdct|silk|decode|mixed] [--frames F]

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM.

  celt (default)  BASELINE.json configs[2]: 65 536 independent 48 kHz stereo 20 ms frames per GPU,
                  full CELT encode (opus_encode(): dc_reject, pre-emphasis, pitch pre-filter, MDCT,
                  PVQ quant_all_bands, range coder) at 96 kb/s VBR complexity 10 -- the configuration the
                  metric "frames encoded/sec" is quoted on. Algorithmic bytes: 3 840 B PCM in + packet
                  (~255 B) + 8 B (len, rng) per frame (~4 090 B/frame).
  silk            BASELINE.json configs[3]: 65 536 function-boundary records, silk_burg_modified + silk_NSQ
                  (16 kHz mono, order 16, 4 x 80-sample subframes; records captured from the reference encoder).
  mixed           BASELINE.json configs[4]: per GPU 131 072 units = 7/8 CELT frames (as celt) + 1/8 SILK records (as silk);
                  with --gpus 8 that is the 1 M-unit corpus sharded over the node.
  decode          the packets of configs[2] through opusgpu_decode_batch (fresh decoder each).
  mdct            BASELINE.json configs[1]: 4 096 frames, clt_mdct_forward + clt_mdct_backward only
                  (33 600 algorithmic bytes per stereo frame) -- the HBM-bound slice.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank encodes its own shard of F
frames -- the frame corpus partitions with no data-path collective -- then the packets are gathered to
rank 0 over RCCL (inside the timed region, once per step; rows trimmed to the longest packet, the exchange of step k
overlapping the encode of step k+1), so scaling is "weak" and `value` is the whole-job frames/s.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
timed with events on the launch stream) and `cpu_baseline` (the reference's own C code timed on this
box's host cores over a bounded sample).
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
BYTES_FWD = 2 * 1080 * 4 + 2 * 960 * 4                   # 16 320 B / stereo frame  
BYTES_BWD = 2 * 960 * 4 + 2 * 1080 * 4 + 2 * 120 * 4     # 17 280 B / stereo frame
PCM_BYTES = 960 * 2 * 2                                   # 3 840 B / stereo frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="celt", choices=["celt", "mdct", "silk", "silk_deldec", "decode", "mixed"])
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("CONCENTUS_BENCH_STREAMS", "1")),
                    help="celt: HIP streams the consecutive batches (steps) alternate over (each with its own workspace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL packet gather (N > 1)")
    return ap.parse_args()


def host_threads():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, 16)          # one GPU's share of the host (the box allots 16 per GPU)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cpu_baseline_mdct(seed_sig):
    """Reference clt_mdct_forward_c + clt_mdct_backward_c (oracle/_ref, kind "reference"), or our C
    restatement (kind "port") if that library did not travel, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = host_threads()
    n = 2048
    sig = np.ascontiguousarray(np.tile(seed_sig, (n // seed_sig.shape[0] + 1, 1, 1))[:n])
    freq = np.zeros((n, 2, 960), np.int32)
    rec = sig.copy()
    refdrv = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    if os.path.exists(refdrv):
        drv = C.CDLL(refdrv)
        drv.refdrv_mdct_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]

        def run(threads):
            drv.refdrv_mdct_batch(_p(sig), _p(freq), _p(rec), n * 2, 0, 3, threads)
        kind = "reference"
    else:
        import oraclelib
        orc = oraclelib.lib()

        def run(threads):
            orc.orc_mdct_forward_batch(_p(sig), _p(freq), n, 2, 0)
            orc.orc_mdct_backward_batch(_p(freq), _p(rec), n, 2, 0)
        kind = "port"
        cores = 1
    one, multi = _time_cpu(run, n, cores)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": kind,
            "sample": "%d stereo frames x clt_mdct_forward_c+clt_mdct_backward_c (shift 0) per pass, "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def _time_cpu(run, n, cores):
    run(1)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 4.0:
        run(1)
        reps += 1
    one = reps * n / (time.perf_counter() - t0)
    multi = one
    if cores > 1:
        run(cores)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 8.0:
            run(cores)
            reps += 1
        multi = reps * n / (time.perf_counter() - t0)
    return one, multi


def cpu_baseline_celt(pcm_sample, cfgvals):
    """The reference's own opus_encode() (unmodified opus-fix, FIXED_POINT, -O2; oracle/_ref) over a
    bounded sample of the same workload: independent frames = fresh opus_encoder_create + opus_demo
    ctl sequence + one opus_encode per frame."""
    refdrv = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    if not os.path.exists(refdrv):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/librefdrv.so did not travel; no CPU baseline"}
    drv = C.CDLL(refdrv)

    class Cfg(C.Structure):
        _fields_ = [(k, C.c_int32) for k in "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]
    cfg = Cfg(*cfgvals)
    cores = host_threads()
    n = pcm_sample.shape[0]
    pcm = np.ascontiguousarray(pcm_sample)
    out = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rng = np.zeros(n, np.uint32)

    def run(threads):
        drv.refdrv_encode_frames(C.byref(cfg), _p(pcm), C.c_long(n), 1, _p(out), 1280, _p(lens), _p(rng), threads)
    one, multi = _time_cpu(run, n, cores)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "reference",
            "sample": "%d independent frames per pass through opus-fix opus_encode() (create+ctl+encode per frame), "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def cpu_baseline_decode(pk, ln):
    """The reference's own opus_decode() over a bounded sample: fresh opus_decoder_create + one opus_decode per packet."""
    refdrv = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    if not os.path.exists(refdrv):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/librefdrv.so did not travel; no CPU baseline"}
    drv = C.CDLL(refdrv)
    cores = host_threads()
    n = pk.shape[0]
    pk = np.ascontiguousarray(pk)
    ln = np.ascontiguousarray(ln.astype(np.int32))
    pcm = np.zeros((n, 960, 2), np.int16)
    rng = np.zeros(n, np.uint32)
    ret = np.zeros(n, np.int32)

    def run(threads):
        drv.refdrv_decode_frames(_p(pk), pk.shape[1], _p(ln), C.c_long(n), 1, _p(pcm), _p(rng), _p(ret), threads)
    one, multi = _time_cpu(run, n, cores)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "reference",
            "sample": "%d independent packets per pass through opus-fix opus_decode() (create+decode per packet), "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def cpu_baseline_silk(rec, n):
    """CPU baseline for the SILK records: our C restatement (oracle/oracle_silk.c, kind "port"; the reference's
    silk_NSQ_c needs its whole encoder state struct, so it is not driven directly), chunks on a thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    from concurrent.futures import ThreadPoolExecutor
    orc = oraclelib.lib()
    cores = host_threads()
    bi = np.ascontiguousarray(np.tile(rec["burg_in"], (n // 80 + 1, 1))[:n])
    ni = np.ascontiguousarray(np.tile(rec["nsq_in"], (n // 80 + 1, 1))[:n])
    st0 = np.ascontiguousarray(np.tile(rec["nsq_state_in"], (n // 80 + 1, 1))[:n])
    bo = np.zeros((n, 72), np.uint8)
    no = np.zeros((n, 320), np.uint8)

    def work(lo, hi):
        st = st0[lo:hi].copy()
        orc.orc_silk_burg_batch(C.c_void_p(bi.ctypes.data + lo * 784), C.c_void_p(bo.ctypes.data + lo * 72), hi - lo)
        orc.orc_silk_nsq_batch(C.c_void_p(ni.ctypes.data + lo * 1640), _p(st), C.c_void_p(no.ctypes.data + lo * 320), hi - lo)

    def run(threads):
        if threads == 1:
            work(0, n)
            return
        per = (n + threads - 1) // threads
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda t: work(t * per, min(n, (t + 1) * per)), range(threads)))
    one, multi = _time_cpu(run, n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port",
            "sample": "%d records (silk_burg_modified + silk_NSQ each) per pass through oracle/oracle_silk.c, repeated ~8 s on "
                      "%d thread(s); 1 thread: %.0f records/s" % (n, cores, one)}


def cpu_baseline_silk_dd(rec, n):
    """CPU baseline for the silk_NSQ_del_dec records: the C restatement (oracle/oracle_silk.c, kind "port"), chunks on a
    thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    from concurrent.futures import ThreadPoolExecutor
    orc = oraclelib.lib()
    cores = host_threads()
    m = rec["dd_in"].shape[0]
    di = np.ascontiguousarray(np.tile(rec["dd_in"], (n // m + 1, 1))[:n])
    st0 = np.ascontiguousarray(np.tile(rec["dd_state_in"], (n // m + 1, 1))[:n])
    do = np.zeros((n, 324), np.uint8)

    def work(lo, hi):
        st = st0[lo:hi].copy()
        orc.orc_silk_nsq_del_dec_batch(C.c_void_p(di.ctypes.data + lo * 1648), _p(st), C.c_void_p(do.ctypes.data + lo * 324), hi - lo)

    def run(threads):
        if threads == 1:
            work(0, n)
            return
        per = (n + threads - 1) // threads
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda t: work(t * per, min(n, (t + 1) * per)), range(threads)))
    one, multi = _time_cpu(run, n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port",
            "sample": "%d silk_NSQ_del_dec records per pass through oracle/oracle_silk.c, repeated ~8 s on %d thread(s); "
                      "1 thread: %.0f records/s" % (n, cores, one)}


def silk_records(F, rank, dev):
    """F function-boundary records on the device: the 80 captured from the reference encoder (tests/golden), tiled, the
    Burg inputs dithered per record."""
    import torch
    g = np.load(os.path.join(ROOT, "tests", "golden", "silk_golden.npz"))
    rec = {k[5:]: g[k] for k in g.files}
    reps = F // 80 + 1
    rng = np.random.default_rng(4 + rank)
    bi_h = np.tile(rec["burg_in"], (reps, 1))[:F].copy()
    bi_h[:, :768].view(np.int16)[...] += rng.integers(-3, 4, size=(F, 384), dtype=np.int16)
    ni_h = np.tile(rec["nsq_in"], (reps, 1))[:F].copy()
    st_h = np.tile(rec["nsq_state_in"], (reps, 1))[:F].copy()
    bi, ni, st0 = (torch.from_numpy(x).to(dev) for x in (bi_h, ni_h, st_h))
    return rec, bi, ni, st0


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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    traffic_db = {}
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            traffic_db = json.load(open(tp))
        except Exception:
            traffic_db = {}

    if a.workload == "mdct":
        F = a.frames or 4096
        steps = a.steps or 200
        warm = a.warmup if a.warmup is not None else 20
        rng = np.random.default_rng(2 + rank)      # seed 2, int16 uniform x 4096 (Q12)
        host = (rng.integers(-16384, 16384, size=(F, 2, 1080), dtype=np.int64) * 4096).astype(np.int32)
        sig = torch.from_numpy(host).to(dev)
        freq = torch.empty((F, 2, 960), dtype=torch.int32, device=dev)
        rec = sig.clone()
        for _ in range(warm):
            ca.mdct_forward_batch(sig, freq, shift=0)
            ca.mdct_backward_batch(freq, rec, shift=0)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.mdct_forward_batch(sig, freq, shift=0)
            ev[k][1].record()
            ca.mdct_backward_batch(freq, rec, shift=0)
            ev[k][2].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        fwd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        bwd_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        if fwd_ms >= bwd_ms:
            kname, kbytes, kms = "mdct_forward_kernel<0>", BYTES_FWD * F, fwd_ms
        else:
            kname, kbytes, kms = "mdct_backward_kernel<0>", BYTES_BWD * F, bwd_ms
        metric = "48kHz stereo 20ms CELT frames/sec (clt_mdct_forward+backward only, config #2)"
        workload = ("configs[1]: %d independent 48 kHz stereo 20 ms frames per GPU, clt_mdct_forward+backward "
                    "only (shift 0), bit-exact vs FIXED_POINT" % F)
        dtype = "int32"
        extra = {"other_kernel_ms": round(bwd_ms if kname.startswith("mdct_forward") else fwd_ms, 5)}
        cpu = (lambda: cpu_baseline_mdct(host[:256]))
    elif a.workload == "decode":
        # packets of config #3 (GPU-encoded, bit-exact with the reference) decoded by fresh decoders
        F = a.frames or 65536
        steps = a.steps or 5
        warm = a.warmup if a.warmup is not None else 1
        rng = np.random.default_rng(3 + rank)
        host = rng.integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)
        pk, ln, _r = ca.encode_independent(torch.from_numpy(host).to(dev), ca.default_config(2, 96000))
        torch.cuda.synchronize()
        dec = ca.OpusDecoderBatch(F, device=dev)
        for _ in range(warm):
            dec.reset()
            pcm, ret = dec.decode(pk, ln)
        import ctypes
        L = ca.lib.load()
        L.opusgpu_kernel_timing_enable(1)
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            dec.reset()                           # every packet is the first of its own stream (fresh decoder)
            pcm, ret = dec.decode(pk, ln)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        assert (ret.cpu().numpy() == 960).all(), "decoder reported an error"
        NK = 10
        ksum = (ctypes.c_double * NK)()
        kcnt = (ctypes.c_int * NK)()
        ca.lib.check(L.opusgpu_kernel_timing_read(ksum, kcnt, NK), "opusgpu_kernel_timing_read")
        L.opusgpu_kernel_timing_enable(0)
        mean_len = float(ln.float().mean().item())
        STATE = int(L.opusgpu_celt_dec_state_size())
        per_frame = {7: ("celt_decode_lane_kernel", mean_len + 4 + 3840 + 600 + 8),      # packet -> X + energies/params
                     8: ("celt_decode_synth_kernel", 3840 + 2 * 4 * (1148 + 1148 + 1080)), # X, history shift, overlap-add
                     9: ("celt_decode_post_kernel", 2 * 4 * 2 * 960 + 3840)}             # out_syn r/w, pcm
        kern = []
        for i, (nm, b) in per_frame.items():
            if kcnt[i]:
                avg = ksum[i] / kcnt[i]
                kern.append({"kernel": nm, "avg_launch_ms": round(avg, 5), "algorithmic_bytes_per_launch": int(F * b),
                             "achieved": round(F * b / (avg * 1e-3) / 1e9, 2), "traffic": traffic_db.get(nm)})
        kern.sort(key=lambda k: -k["avg_launch_ms"])
        kname, kbytes, kms = kern[0]["kernel"], kern[0]["algorithmic_bytes_per_launch"], kern[0]["avg_launch_ms"]
        metric = "48kHz stereo 20ms CELT frames decoded/sec"
        workload = ("%d independent CELT-only 20 ms stereo packets per GPU (config #3's packets, mean %.1f B), each through a "
                    "fresh decoder (state reset included in the step), PCM bit-exact vs FIXED_POINT opus_decode()" % (F, mean_len))
        dtype = "int16/int32 fixed-point"
        extra = {"mean_packet_bytes": round(mean_len, 2), "other_kernels": kern[1:], "state_bytes_per_stream": STATE}
        pk_h, ln_h = pk[:4096].cpu().numpy(), ln[:4096].cpu().numpy()
        cpu = (lambda: cpu_baseline_decode(pk_h, ln_h))
    elif a.workload == "silk_deldec":
        # silk_NSQ_del_dec (the quantizer of complexity >= 4) over records captured at complexity 5/7/10
        F = a.frames or 65536
        steps = a.steps or 10
        warm = a.warmup if a.warmup is not None else 2
        g = np.load(os.path.join(ROOT, "tests", "golden", "silk_dd_golden.npz"))
        rec = {k[5:]: g[k] for k in g.files}
        m = rec["dd_in"].shape[0]
        rng = np.random.default_rng(6 + rank)
        di_h = np.tile(rec["dd_in"], (F // m + 1, 1))[:F].copy()
        di_h[:, 36:40].view(np.int32)[:, 0] = rng.integers(0, 4, size=F)              # dither seed per record
        di = torch.from_numpy(di_h).to(dev)
        st0 = torch.from_numpy(np.tile(rec["dd_state_in"], (F // m + 1, 1))[:F].copy()).to(dev)
        st = st0.clone()
        do = torch.empty((F, 324), dtype=torch.uint8, device=dev)
        for _ in range(warm):
            st.copy_(st0)
            ca.silk_NSQ_del_dec(di, st, do)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.silk_NSQ_del_dec(di, st, do)       # states keep evolving from step to step, as a stream would
            ev[k][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_nsq_del_dec_kernel"
        kbytes = F * (1648 + 2 * 4380 + 324)
        metric = "SILK 16kHz mono 20ms frames/sec (silk_NSQ_del_dec records)"
        workload = ("%d function-boundary records per GPU (84 captured from the reference encoder on 16 kHz mono voice at "
                    "32 kb/s, complexity 5/7/10 = 2/3/4 delayed-decision states, tiled), silk_NSQ_del_dec, bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {}
        cpu = (lambda: cpu_baseline_silk_dd(rec, 4096))
    elif a.workload == "silk":
        F = a.frames or 65536
        steps = a.steps or 20
        warm = a.warmup if a.warmup is not None else 3
        rec, bi, ni, st0 = silk_records(F, rank, dev)
        st = st0.clone()
        bo = torch.empty((F, 72), dtype=torch.uint8, device=dev)
        pulses = torch.empty((F, 320), dtype=torch.int8, device=dev)
        for _ in range(warm):
            st.copy_(st0)
            ca.silk_burg_modified(bi, bo)
            ca.silk_NSQ(ni, st, pulses)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.silk_burg_modified(bi, bo)
            ev[k][1].record()
            ca.silk_NSQ(ni, st, pulses)        # states keep evolving from step to step, as a stream would
            ev[k][2].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        burg_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        kname = "silk_nsq_kernel"
        kbytes = F * (1640 + 2 * 4380 + 320)
        metric = "SILK 16kHz mono 20ms frames/sec (silk_burg_modified + silk_NSQ records, config #4)"
        workload = ("configs[3]: %d function-boundary records per GPU (80 captured from the reference encoder on "
                    "16 kHz mono voice at 32 kb/s complexity 3, tiled), silk_burg_modified + silk_NSQ, bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {"burg_kernel_ms": round(burg_ms, 5), "burg_GBps": round(F * 856 / (burg_ms * 1e-3) / 1e9, 2)}
        cpu = (lambda: cpu_baseline_silk(rec, 8192))
    else:
        mixed = a.workload == "mixed"
        # mixed = BASELINE configs[4]: per GPU a shard of 131 072 units, 7/8 CELT frames (config #3's kind) and 1/8 SILK
        # records (config #4's kind); the 8-GPU job is then the 1 M-unit corpus
        FT = a.frames or (131072 if mixed else 65536)
        from concentus_amd.sharding import mixed_counts
        F, NS = mixed_counts(FT) if mixed else (FT, 0)
        steps = a.steps or (5 if mixed else 10)
        warm = a.warmup if a.warmup is not None else 2
        cfg = ca.default_config(2, 96000)          # opus_demo restricted-lowdelay 48000 2 96000, complexity 10, VBR
        rng = np.random.default_rng((5 if mixed else 3) + rank)      # seed 3 (config #3) / 5 (config #5), uniform int16 in [-8192, 8191]
        host = rng.integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)
        pcm = torch.from_numpy(host).to(dev)
        out = lens = None
        if mixed:
            _rec, s_bi, s_ni, s_st = silk_records(NS, rank, dev)
            s_bo = torch.empty((NS, 72), dtype=torch.uint8, device=dev)
            s_pulses = torch.empty((NS, 320), dtype=torch.int8, device=dev)
            silk_stream = torch.cuda.Stream(device=dev)

            def silk_part():
                # the SILK records are independent of the CELT frames: their two small kernels go to a stream of their own
                silk_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(silk_stream):
                    ca.silk_burg_modified(s_bi, s_bo)
                    ca.silk_NSQ(s_ni, s_st, s_pulses)

            def silk_join():
                torch.cuda.current_stream().wait_stream(silk_stream)
        for _ in range(warm):
            if mixed:
                silk_part()
            out, lens, _r = ca.encode_independent(pcm, cfg)
            if mixed:
                silk_join()
        torch.cuda.synchronize()
        from concentus_amd.sharding import gather_packets
        import ctypes
        L = ca.lib.load()
        L.opusgpu_kernel_timing_enable(1)          # HIP events around each kernel, on the launch stream
        barrier()
        t0 = time.perf_counter()
        side = [torch.cuda.Stream(device=dev) for _ in range(max(a.streams, 1))] if a.streams > 1 else None
        if side:
            for s_ in side:                        # workspaces of the side streams exist before the clock starts
                s_.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s_):
                    ca.encode_independent(pcm, cfg)
            torch.cuda.synchronize()
            L.opusgpu_kernel_timing_read((ctypes.c_double * 8)(), (ctypes.c_int * 8)(), 8)
            barrier()
            t0 = time.perf_counter()
        pending = []
        for k in range(steps):
            if side:
                # consecutive batches on alternating streams: the next batch's front kernels fill the CUs the
                # back kernel's last wavefronts leave idle
                with torch.cuda.stream(side[k % len(side)]):
                    out, lens, _r = ca.encode_independent(pcm, cfg)
            else:
                if mixed:
                    silk_part()
                out, lens, _r = ca.encode_independent(pcm, cfg)
                if mixed:
                    silk_join()
            if world > 1 and not a.no_gather:
                # the only exchange of the path: packets + lengths to rank 0 over RCCL/xGMI. Rows are trimmed to the
                # longest packet and the exchange of step k is left pending while step k+1 is encoded.
                if side:
                    torch.cuda.current_stream().wait_stream(side[k % len(side)])
                for pg in pending:
                    gathered = pg.wait()
                pending = [gather_packets(out, lens, _r, world, sizes=[F] * world, trim=True, async_op=True)]
                if mixed:
                    pending.append(gather_packets(s_pulses, s_bo, s_bo[:, :4], world, sizes=[NS] * world, async_op=True))
        for pg in pending:
            gathered = pg.wait()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        KNAMES = ["celt_front_kernel", "celt_back_kernel", "celt_back_lane_kernel", "celt_dc_reject_kernel",
                  "celt_front1_kernel",
                  "celt_transient_kernel" if os.environ.get("OPUSGPU_TRANSIENT_LANE") else "celt_transient_tile_kernel",
                  "celt_front2_kernel"]
        NK = len(KNAMES)
        ksum = (ctypes.c_double * NK)()
        kcnt = (ctypes.c_int * NK)()
        ca.lib.check(L.opusgpu_kernel_timing_read(ksum, kcnt, NK), "opusgpu_kernel_timing_read")
        L.opusgpu_kernel_timing_enable(0)
        lens_h = lens.cpu().numpy()
        assert (lens_h > 0).all(), "encoder reported an error"
        mean_len = float(lens_h.mean())
        # algorithmic bytes per frame and kernel: what each kernel must read and write once
        MID_BYTES = int(ca.encoder.MID_RECORD_BYTES)
        IN_BYTES = 2 * 1080 * 4
        per_frame = {
            "celt_front_kernel": PCM_BYTES + MID_BYTES,
            "celt_dc_reject_kernel": PCM_BYTES + PCM_BYTES,
            "celt_front1_kernel": PCM_BYTES + IN_BYTES + (MID_BYTES - PCM_BYTES),
            "celt_transient_kernel": IN_BYTES + 8,
            "celt_transient_tile_kernel": 2 * IN_BYTES + 8,       # two passes over the time signal by construction
            "celt_front2_kernel": IN_BYTES + MID_BYTES,
            "celt_back_kernel": MID_BYTES + mean_len + 8,
            "celt_back_lane_kernel": MID_BYTES + mean_len + 8,
        }
        kern = []
        for i, nm in enumerate(KNAMES):
            if kcnt[i]:
                avg = ksum[i] / kcnt[i]
                fpl = F * steps / kcnt[i]
                kern.append({"kernel": nm, "avg_launch_ms": round(avg, 5), "frames_per_launch": int(fpl),
                             "algorithmic_bytes_per_launch": int(fpl * per_frame[nm]),
                             "achieved": round(fpl * per_frame[nm] / (avg * 1e-3) / 1e9, 2), "traffic": traffic_db.get(nm)})
        kern.sort(key=lambda k: -k["avg_launch_ms"])
        dom = kern[0]
        kname, kbytes, kms = dom["kernel"], dom["algorithmic_bytes_per_launch"], dom["avg_launch_ms"]
        metric = "48kHz stereo 20ms CELT frames encoded/sec"
        workload = ("configs[2]: %d independent 48 kHz stereo 20 ms frames per GPU, full CELT encode "
                    "(MDCT + PVQ + range enc) 96 kb/s VBR complexity 10, packets bit-exact vs FIXED_POINT "
                    "opus_encode(); mean packet %.1f B" % (F, mean_len))
        dtype = "int16/int32 fixed-point"
        extra = {"mean_packet_bytes": round(mean_len, 2), "frames_per_launch": dom["frames_per_launch"],
                 "other_kernels": kern[1:], "realtime_factor": None}
        cpu = (lambda: cpu_baseline_celt(host[:4096], (2, 96000, 1, 0, 10, 16, 0, 1500)))
        if mixed:
            metric = "mixed CELT/SILK 20ms frames processed/sec (configs[4]: 7/8 CELT encode, 1/8 SILK burg+NSQ records)"
            workload = ("configs[4]: %d units per GPU = %d independent 48 kHz stereo CELT frames (full encode, 96 kb/s VBR "
                        "complexity 10, mean packet %.1f B) + %d SILK function-boundary records (silk_burg_modified + silk_NSQ), "
                        "bit-exact vs FIXED_POINT; the SILK kernels run on a side stream" % (FT, F, mean_len, NS))
            extra["celt_frames_per_gpu"], extra["silk_records_per_gpu"] = F, NS
            extra.pop("realtime_factor", None)
            F = FT

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    achieved = kbytes / (kms * 1e-3) / 1e9

    if rank == 0:
        value = F * world * steps / elapsed
        if "realtime_factor" in extra:
            extra["realtime_factor"] = round(value * 0.02, 1)        # 20 ms of audio per frame
        out_line = {
            "metric": metric,
            "value": round(value, 1),
            "unit": "records/s" if a.workload.startswith("silk") else "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warm,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": workload, "frames_per_gpu": F, "channels": 2,
                       "sharding": "frames block-partitioned across ranks, no data-path collective; packets gathered to rank 0"},
            "roofline": dict({"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic_db.get(kname),
                              "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": round(kms, 5)}, **extra),
        }
        if not a.no_cpu_baseline:
            try:
                out_line["cpu_baseline"] = cpu()
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out_line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference",
                                            "sample": "failed: %r" % (e,)}
        print(json.dumps(out_line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
