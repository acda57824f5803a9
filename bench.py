#!/usr/bin/env python3
"""bench.py -- throughput of the batched Opus frame path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--workload celt|celt_streams|mdct|silk|silk_deldec|decode|mixed] [--frames F]

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM.

  celt (default)  BASELINE.json configs[2]: 65 536 independent 48 kHz stereo 20 ms frames per GPU (each the first frame
                  of its own stream), full CELT encode (opus_encode(): dc_reject, pre-emphasis, pitch pre-filter, MDCT,
                  PVQ quant_all_bands, range coder) at 96 kb/s VBR complexity 10 -- the configuration the metric
                  "frames encoded/sec" is quoted on. Algorithmic bytes: 3 840 B PCM in + packet + 8 B.
  celt_streams    the same encoder in streams mode: 65 536 streams per GPU, every step encodes the NEXT 20 ms frame of
                  every stream (encoder state, 9 KB per stream, carried in HBM) -- no first-frame short-cuts.
  silk            BASELINE.json configs[3]: 65 536 distinct function-boundary records, silk_burg_modified + silk_NSQ
                  (16 kHz mono, order 16, 4 x 80-sample subframes; records captured from the reference encoder).
  silk_deldec     the same for silk_NSQ_del_dec (the quantiser of complexity >= 4).
  silk_frames     65 536 frames through the WHOLE chain on the device: pitch buffer -> find_pitch_lags -> noise_shape_analysis ->
                  find_pred_coefs -> process_gains -> prefilter -> NSQ_del_dec -> encode_indices + encode_pulses -> range-coder bytes (records between the kernels filled on the
                  device, concentus_amd/silk_chain.py); parity per frame against the reference's own results.
  silk_frames_cbr the same on frames captured from a constant-bitrate encoder, with silk_encode_frame_FIX's bitrate loop: a rate-control
                  step after every pass, frames over / under budget quantised + coded again (about four passes per frame)
  silk_streams    STREAMS MODE of the CBR chain: 65 536 streams, every step encodes two CONSECUTIVE frames of every stream with the
                  inter-frame state (x_buf, prevLag, smoothers, NLSFs, gain index, the prefilter / quantiser states ...) carried on the
                  device (opusgpu_silk_stream_carry_in / _out); the capture supplies the state before the first frame and, per frame,
                  only what is computed outside silk_encode_frame_FIX (samples, VAD results, maxBits, the packet's coder).
  silk_analysis   the five analysis calls of silk_encode_frame_FIX between the VAD and the quantiser (silk_find_pitch_lags_FIX,
                  silk_noise_shape_analysis_FIX, silk_find_pred_coefs_FIX, silk_process_gains_FIX, silk_prefilter_FIX), 65 536
                  distinct captured records each; value = frames/s through all five.
  silk_pred       silk_find_pred_coefs_FIX whole (LTP analysis + quantisation, silk_find_LPC_FIX, silk_process_NLSFs,
                  silk_residual_energy_FIX) over 65 536 distinct records (voiced and unvoiced frames).
  silk_nlsf       silk_process_NLSFs + silk_residual_energy_FIX (the tail of silk_find_pred_coefs_FIX) over 65 536 distinct records
  silk_lpc        silk_find_LPC_FIX (Burg + silk_A2NLSF, and the NLSF interpolation search at complexity >= 4) over 65 536
                  distinct records: the SILK analysis step that feeds / consumes silk_burg_modified.
  mixed           BASELINE.json configs[4]: per GPU 131 072 units = 7/8 CELT frames (as celt) + 1/8 SILK records (as
                  silk); with --gpus 8 that is the 1 M-unit corpus sharded over the node.
  decode          the packets of configs[2] through opusgpu_decode_batch (fresh decoder each).
  mdct            BASELINE.json configs[1]: 4 096 frames, clt_mdct_forward + clt_mdct_backward only
                  (33 600 algorithmic bytes per stereo frame) -- the HBM-bound slice.

N > 1: one rank per GPU. Either the driver launches the ranks (python -m torch.distributed.run ... bench.py --gpus N:
RANK / LOCAL_RANK / WORLD_SIZE in the environment), or `python bench.py --gpus N` alone starts them itself: the parent
process touches no GPU, spawns torch.distributed.run as a CHILD (never exec) and exits with its status. Every rank
encodes its own shard of F frames -- the corpus partitions with no data-path collective -- then the packets are gathered
to rank 0 over RCCL (inside the timed region, once per step; rows trimmed to the longest packet, the exchange of step k
overlapping the encode of step k+1), so scaling is "weak" and `value` is the whole-job frames/s. A world size that
differs from --gpus is an error.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel, timed with HIP events
on the launch stream), `cpu_baseline` (the reference's own C code timed on this box's host cores over a bounded
sample) and `parity_checked` (units of the LAST timed step compared bit-exactly with the reference after the clock
stopped; a mismatch is a failure, not a number).

--rehearse: no GPU and no codec work -- the ranks only rendezvous on gloo and run the packet gather on synthetic
slabs; `value` is null. It exists so the multi-rank launch path can be tested on a CPU-only machine.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4                    # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9                   # MI355X_MICROARCH.md: 2.4 GHz peak engine clock
BYTES_FWD = 2 * 1080 * 4 + 2 * 960 * 4                   # 16 320 B / stereo frame  
BYTES_BWD = 2 * 960 * 4 + 2 * 1080 * 4 + 2 * 120 * 4     # 17 280 B / stereo frame
PCM_BYTES = 960 * 2 * 2                                   # 3 840 B / stereo frame
WORKLOADS = ["celt", "celt_streams", "mdct", "silk", "silk_deldec", "silk_lpc", "silk_nlsf", "silk_pred", "silk_analysis", "silk_frames", "silk_frames_cbr", "silk_streams", "decode", "mixed"]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="celt", choices=WORKLOADS)
    ap.add_argument("--frames", type=int, default=None, help="frames (units) per GPU per step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("CONCENTUS_BENCH_STREAMS", "1")),
                    help="celt: HIP streams the consecutive batches (steps) alternate over (each with its own workspace)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the post-clock comparison with the reference")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL packet gather (N > 1)")
    ap.add_argument("--rehearse", action="store_true", help="CPU only: rendezvous on gloo + packet gather, no codec work")
    return ap.parse_args(argv)


# ---- multi-rank launch ---------------------------------------------------------------------------------------
def launch_ranks(a, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as children of this (GPU-free) process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def host_threads():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, 16)          # one GPU's share of the host (the box allots 16 per GPU)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _refdrv():
    path = os.path.join(ROOT, "oracle", "_ref", "librefdrv.so")
    return C.CDLL(path) if os.path.exists(path) else None


class RefCfg(C.Structure):
    _fields_ = [(k, C.c_int32) for k in "channels bitrate vbr constrained_vbr complexity lsb_depth loss_rate max_data_bytes".split()]


def ref_encode(pcm, cfgvals, fps, threads):
    """The compiled reference's opus_encode() over frames [n][960][2] (stream-major, fps frames per stream)."""
    drv = _refdrv()
    cfg = RefCfg(*cfgvals)
    n = pcm.shape[0]
    pcm = np.ascontiguousarray(pcm)
    out = np.zeros((n, 1280), np.uint8)
    lens = np.zeros(n, np.int32)
    rng = np.zeros(n, np.uint32)
    drv.refdrv_encode_frames(C.byref(cfg), _p(pcm), C.c_long(n), fps, _p(out), 1280, _p(lens), _p(rng), threads)
    return out, lens, rng


def check_packets(what, got_pk, got_len, got_rng, exp_pk, exp_len, exp_rng):
    if not np.array_equal(got_len, exp_len):
        raise SystemExit("PARITY FAILURE (%s): packet lengths differ at %s" % (what, np.nonzero(got_len != exp_len)[0][:8]))
    if not np.array_equal(got_rng.astype(np.uint32), exp_rng):
        raise SystemExit("PARITY FAILURE (%s): final ranges differ" % what)
    w = int(exp_len.max())
    mask = np.arange(w)[None, :] < exp_len[:, None]
    if not np.array_equal(np.where(mask, got_pk[:, :w], 0), np.where(mask, exp_pk[:, :w], 0)):
        raise SystemExit("PARITY FAILURE (%s): packet bytes differ" % what)


# ---- CPU baselines (the checker's code timed on the host; never on the product path) ----------------------------
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


def _no_ref(unit="frames/s"):
    return {"value": None, "unit": unit, "cores": 0, "kind": "reference", "cpu": cpu_model(),
            "sample": "oracle/_ref/librefdrv.so did not travel; no CPU baseline"}


def cpu_baseline_mdct(seed_sig):
    """Reference clt_mdct_forward_c + clt_mdct_backward_c (oracle/_ref, kind "reference"), or our C
    restatement (kind "port") if that library did not travel, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores = host_threads()
    n = 2048
    sig = np.ascontiguousarray(np.tile(seed_sig, (n // seed_sig.shape[0] + 1, 1, 1))[:n])
    freq = np.zeros((n, 2, 960), np.int32)
    rec = sig.copy()
    drv = _refdrv()
    if drv is not None:
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
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": kind, "cpu": cpu_model(),
            "sample": "%d stereo frames x clt_mdct_forward_c+clt_mdct_backward_c (shift 0) per pass, "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def cpu_baseline_celt(pcm_sample, cfgvals, fps=1):
    """The reference's own opus_encode() (unmodified opus-fix, FIXED_POINT, -O2; oracle/_ref) over a bounded sample of
    the same workload. fps == 1: independent frames = fresh opus_encoder_create + opus_demo ctl sequence + one
    opus_encode per frame, with the steady-state (streams) rate of the same library next to it;
    fps > 1: streams of fps frames, one encoder per stream."""
    if _refdrv() is None:
        return _no_ref()
    cores = host_threads()
    n = pcm_sample.shape[0]

    def run(threads):
        ref_encode(pcm_sample, cfgvals, fps, threads)
    one, multi = _time_cpu(run, n, cores)
    if fps == 1:
        sfps = 32
        ns = (n // sfps) * sfps

        def run_s(threads):
            ref_encode(pcm_sample[:ns], cfgvals, sfps, threads)
        _one_s, multi_s = _time_cpu(run_s, ns, cores)
        sample = ("%d independent frames per pass through opus-fix opus_encode() (create+ctl+encode per frame), repeated "
                  "~8 s on %d thread(s); 1 thread: %.0f frames/s; steady state (one encoder per %d-frame stream, same "
                  "frames): %.0f frames/s on %d threads" % (n, cores, one, sfps, multi_s, cores))
        extra = {"steady_state_value": round(multi_s, 1)}
    else:
        sample = ("%d streams x %d consecutive frames per pass through opus-fix opus_encode() (one encoder per stream), "
                  "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n // fps, fps, cores, one))
        extra = {}
    return dict({"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "reference", "cpu": cpu_model(),
                 "sample": sample}, **extra)


def cpu_baseline_decode(pk, ln):
    """The reference's own opus_decode() over a bounded sample: fresh opus_decoder_create + one opus_decode per packet."""
    drv = _refdrv()
    if drv is None:
        return _no_ref()
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
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "reference", "cpu": cpu_model(),
            "sample": "%d independent packets per pass through opus-fix opus_decode() (create+decode per packet), "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def _pool_run(work, n):
    from concurrent.futures import ThreadPoolExecutor

    def run(threads):
        if threads == 1:
            work(0, n)
            return
        per = (n + threads - 1) // threads
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda t: work(t * per, min(n, (t + 1) * per)), range(threads)))
    return run


def cpu_baseline_silk_encoder(complexity=7, vbr=0):
    """The reference's own opus_encode() as a 16 kHz mono VOIP SILK encoder (32 kb/s, 20 ms frames; constant bitrate unless vbr) on synthetic
    speech, one encoder per host thread (oracle/ref_driver.c refdrv_silk_encode_loop) -- the whole encoder, i.e. silk_encode_frame_FIX
    plus what surrounds it (input filters, resampler, VAD, packet assembly). kind "reference"."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import silk_corpus
    drv = _refdrv()
    if drv is None or not hasattr(drv, "refdrv_silk_encode_loop"):
        return _no_ref()
    drv.refdrv_silk_encode_loop.restype = C.c_long
    drv.refdrv_silk_encode_loop.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    cores = host_threads()
    nfr = 1000
    pcm = np.ascontiguousarray(silk_corpus.synth_voice(nfr * 320, 4242))
    rates = []
    for threads in (1, cores):
        loops, t = 1, 0.0
        while True:
            t0 = time.perf_counter()
            assert drv.refdrv_silk_encode_loop(_p(pcm), nfr, 320, 16000, 32000, vbr, complexity, loops, threads) > 0
            t = time.perf_counter() - t0
            if t > 3.0 or loops >= 64:
                break
            loops *= 2
        rates.append(nfr * loops * threads / t)
    return {"value": round(rates[1], 1), "unit": "frames/s", "cores": cores, "kind": "reference", "cpu": cpu_model(),
            "sample": "opus-fix opus_encode() as a 16 kHz mono VOIP %s 32 kb/s SILK encoder (complexity %d) over 1 000 frames of synthetic speech, "
                      "repeated >= 3 s, one encoder per thread at 1 and %d threads; the WHOLE encoder (input filters, VAD, packet assembly included); "
                      "1 thread: %.0f frames/s" % ("VBR" if vbr else "CBR", complexity, cores, rates[0])}


def cpu_baseline_silk(bi, ni, st0):
    """CPU baseline for the SILK records: our C restatement (oracle/oracle_silk.c, kind "port"; the reference's
    silk_NSQ_c needs its whole encoder state struct, so it is not driven directly), chunks on a thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    orc = oraclelib.lib()
    cores = host_threads()
    n = bi.shape[0]
    bo = np.zeros((n, 72), np.uint8)
    no = np.zeros((n, 320), np.uint8)

    def work(lo, hi):
        st = st0[lo:hi].copy()
        orc.orc_silk_burg_batch(C.c_void_p(bi.ctypes.data + lo * 784), C.c_void_p(bo.ctypes.data + lo * 72), hi - lo)
        orc.orc_silk_nsq_batch(C.c_void_p(ni.ctypes.data + lo * 1640), _p(st), C.c_void_p(no.ctypes.data + lo * 320), hi - lo)
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d records (silk_burg_modified + silk_NSQ each) per pass through oracle/oracle_silk.c, repeated ~8 s on "
                      "%d thread(s); 1 thread: %.0f records/s" % (n, cores, one)}


def cpu_baseline_silk_lpc(lin):
    """CPU baseline for silk_find_LPC_FIX records: the kernel sources compiled for the host (tests/emu, kind "port"), chunks on
    a thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    emu = emulib.lib()
    cores = host_threads()
    n = lin.shape[0]
    out = np.zeros((n, 40), np.uint8)

    def work(lo, hi):
        emu.emu_silk_find_lpc(C.c_void_p(lin.ctypes.data + lo * 832), C.c_void_p(out.ctypes.data + lo * 40), C.c_long(hi - lo))
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d silk_find_LPC_FIX records per pass through the host build of concentus_amd/csrc/silk_lpc_dev.h, repeated "
                      "~8 s on %d thread(s); 1 thread: %.0f records/s" % (n, cores, one)}


def cpu_baseline_silk_nlsf(nin, ein):
    """CPU baseline for silk_process_NLSFs + silk_residual_energy_FIX records: the kernel sources compiled for the host
    (tests/emu, kind "port"), chunks on a thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    emu = emulib.lib()
    cores = host_threads()
    n = nin.shape[0]
    out = np.zeros((n, 120), np.uint8)
    eout = np.zeros((n, 40), np.uint8)

    def work(lo, hi):
        emu.emu_silk_process_nlsfs(C.c_void_p(nin.ctypes.data + lo * 96), C.c_void_p(out.ctypes.data + lo * 120), C.c_long(hi - lo))
        emu.emu_silk_residual_energy(C.c_void_p(ein.ctypes.data + lo * 864), C.c_void_p(eout.ctypes.data + lo * 40), C.c_long(hi - lo))
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d (silk_process_NLSFs + silk_residual_energy_FIX) record pairs per pass through the host build of "
                      "concentus_amd/csrc/silk_nlsf_dev.h, repeated ~8 s on %d thread(s); 1 thread: %.0f records/s" % (n, cores, one)}


ANALYSIS_OPS = (   # (corpus kind, input key, output key, bytes compared, emu entry, python op, kernel)
    ("pitch", "pitch_in", "pitch_out", 1380, "emu_silk_find_pitch_lags", "silk_find_pitch_lags", "silk_find_pitch_lags_kernel"),
    ("shape", "shape_in", "shape_out", 380, "emu_silk_noise_shape_analysis", "silk_noise_shape_analysis", "silk_noise_shape_kernel"),
    ("fpc", "fpc_in", "fpc_out", 204, "emu_silk_find_pred_coefs", "silk_find_pred_coefs", "silk_find_pred_coefs_kernel"),
    ("gains", "gains_in", "gains_out", 52, "emu_silk_process_gains", "silk_process_gains", "silk_process_gains_kernel"),
    ("prefilter", "prefilter_in", "prefilter_out", 1280, "emu_silk_prefilter", "silk_prefilter", "silk_prefilter_kernel"),
)


def cpu_baseline_silk_analysis(recs):
    """CPU baseline for the SILK analysis chain: the kernel sources compiled for the host (tests/emu, kind "port"), the five
    functions one after the other on each chunk, chunks on a thread pool."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    emu = emulib.lib()
    cores = host_threads()
    n = recs["pitch_in"].shape[0]
    outs = {k: np.zeros((n, recs[ok].shape[1]), np.uint8) for _, _, ok, _, _, k, _ in ANALYSIS_OPS}
    st = np.array(recs["prefilter_state_in"])

    def work(lo, hi):
        for kind, ik, ok, _, entry, k, _ in ANALYSIS_OPS:
            i, o = recs[ik], outs[k]
            args = [C.c_void_p(i.ctypes.data + lo * i.shape[1])]
            if kind == "prefilter":
                args.append(C.c_void_p(st.ctypes.data + lo * st.shape[1]))
            args += [C.c_void_p(o.ctypes.data + lo * o.shape[1]), C.c_long(hi - lo)]
            getattr(emu, entry)(*args)
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d frames' worth of records per pass (find_pitch_lags, noise_shape_analysis, find_pred_coefs, process_gains, "
                      "prefilter) through the host build of the kernel sources, repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s"
                      % (n, cores, one)}


def cpu_baseline_silk_bits(bin_, ec0):
    """CPU baseline for silk_encode_indices + silk_encode_pulses records: the kernel sources compiled for the host (tests/emu, "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    emu = emulib.lib()
    cores = host_threads()
    n = bin_.shape[0]
    out = np.zeros((n, 16), np.uint8)

    def work(lo, hi):
        ec = ec0[lo:hi].copy()
        emu.emu_silk_encode_bits(C.c_void_p(bin_.ctypes.data + lo * 416), _p(ec), C.c_void_p(out.ctypes.data + lo * 16), C.c_long(hi - lo))
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d (silk_encode_indices + silk_encode_pulses) records per pass through the host build of "
                      "concentus_amd/csrc/silk_bits_dev.h, repeated ~8 s on %d thread(s); 1 thread: %.0f frames/s" % (n, cores, one)}


def cpu_baseline_silk_pred(fin):
    """CPU baseline for silk_find_pred_coefs_FIX records: the kernel sources compiled for the host (tests/emu, kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulib
    emu = emulib.lib()
    cores = host_threads()
    n = fin.shape[0]
    out = np.zeros((n, 208), np.uint8)

    def work(lo, hi):
        emu.emu_silk_find_pred_coefs(C.c_void_p(fin.ctypes.data + lo * 2688), C.c_void_p(out.ctypes.data + lo * 208), C.c_long(hi - lo))
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d silk_find_pred_coefs_FIX records per pass through the host build of concentus_amd/csrc/silk_pred_dev.h, "
                      "repeated ~8 s on %d thread(s); 1 thread: %.0f records/s" % (n, cores, one)}


def cpu_baseline_silk_dd(di, st0):
    """CPU baseline for the silk_NSQ_del_dec records: the C restatement (oracle/oracle_silk.c, kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    orc = oraclelib.lib()
    cores = host_threads()
    n = di.shape[0]
    do = np.zeros((n, 324), np.uint8)

    def work(lo, hi):
        st = st0[lo:hi].copy()
        orc.orc_silk_nsq_del_dec_batch(C.c_void_p(di.ctypes.data + lo * 1648), _p(st), C.c_void_p(do.ctypes.data + lo * 324), hi - lo)
    one, multi = _time_cpu(_pool_run(work, n), n, cores)
    return {"value": round(multi, 1), "unit": "records/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": "%d silk_NSQ_del_dec records per pass through oracle/oracle_silk.c, repeated ~8 s on %d thread(s); "
                      "1 thread: %.0f records/s" % (n, cores, one)}


# ---- SILK input records ------------------------------------------------------------------------------------------
def silk_records(F, kind, rank):
    """F function-boundary records (host arrays) + a description. Preferred: F DISTINCT records captured on this box
    from the unmodified reference encoder running on synthetic speech (tests/silk_corpus.py; needs the capture build
    of the reference, oracle/_ref/libopus_ref_silkcap.so, which travels with the snapshot) -- input generation, before
    any clock starts; the corpus also carries the reference's outputs, which the post-clock parity check uses.
    Fallback when the capture library is absent: the committed golden records tiled (and said so)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import silk_corpus
    if silk_corpus.available():
        c = silk_corpus.corpus(F, kind, seed=20260401 + 1000003 * rank)
        return {k: np.array(v) for k, v in c.items()}, \
            "%d distinct records captured from the reference encoder (synthetic 16 kHz mono speech, 32 kb/s VOIP, %s)" % (
                F, "complexity 3" if kind == "nsq" else "complexity 5/7/10 = 2/3/4 delayed-decision states")
    g = np.load(os.path.join(ROOT, "tests", "golden", "silk_golden.npz" if kind == "nsq" else "silk_dd_golden.npz"))
    rec = {k[5:]: g[k] for k in g.files}
    m = next(iter(rec.values())).shape[0]
    out = {k: np.ascontiguousarray(np.tile(v, (F // m + 1, 1))[:F]) for k, v in rec.items()}
    for k in ("burg_out", "nsq_out", "nsq_state_out", "dd_out", "dd_state_out"):
        out.pop(k, None)              # tiled inputs are perturbed below, the captured outputs no longer apply
    rng = np.random.default_rng(4 + rank)
    if kind == "nsq":
        out["burg_in"][:, :768].view(np.int16)[...] += rng.integers(-3, 4, size=(F, 384), dtype=np.int16)
    else:
        out["dd_in"][:, 36:40].view(np.int32)[:, 0] = rng.integers(0, 4, size=F)
    return out, "%d records = the %d committed golden records tiled (capture library absent)" % (F, m)


def rehearse(a, world, rank):
    """CPU-only rehearsal of the N-rank path: gloo rendezvous, world-size check, the packet gather on synthetic slabs."""
    import torch
    import torch.distributed as dist
    from concentus_amd.sharding import gather_packets
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus
    F = a.frames or 256
    g = torch.Generator().manual_seed(100 + rank)
    lens = torch.randint(100, 400, (F,), generator=g, dtype=torch.int32)
    out = torch.randint(0, 256, (F, 1500), generator=g, dtype=torch.uint8)
    rng = torch.randint(0, 1 << 30, (F,), generator=g, dtype=torch.int32)
    ok = True
    if world > 1:
        res = gather_packets(out, lens, rng, world, sizes=[F] * world, trim=True, async_op=True).wait()
        if rank == 0:
            ok = res[0].shape[0] == F * world and bool((res[1][:F] == lens).all()) and bool((res[2][:F] == rng).all())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "rehearsal (no codec work)", "value": None, "unit": "frames/s", "n_gpus": world,
                          "rehearsal": True, "gather_ok": ok, "frames_per_rank": F}))
    return 0 if ok else 1


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # no launcher around us: start the ranks ourselves, BEFORE anything in this process touches a GPU
        return launch_ranks(a, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d)\n"
                         % (a.gpus, world, a.gpus, a.gpus))
        return 2
    if a.rehearse:
        return rehearse(a, world, rank)

    import torch
    import torch.distributed as dist
    # CONCENTUS_BENCH_BACKEND=gloo + CONCENTUS_BENCH_ONE_DEVICE=1: a rehearsal of the N-rank path on a ONE-GPU box -- every rank
    # runs the real kernels on cuda:0 and the gather goes through gloo on host copies (RCCL needs one device per rank). Not a
    # measurement: the line then carries "rehearsal": true.
    backend = os.environ.get("CONCENTUS_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("CONCENTUS_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == a.gpus and dist.get_backend() == backend
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    host_gather = world > 1 and backend != "nccl"
    import concentus_amd as ca
    L = ca.lib.load()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def load_db(name):
        p = os.path.join(ROOT, "profiles", name)
        try:
            return json.load(open(p)) if os.path.exists(p) else {}
        except Exception:
            return {}
    traffic_db = load_db("traffic.json")
    pmc_db = load_db("pmc.json")           # per kernel: wave-instructions per launch at the default batch (tools/pmc_db.py)
    sample_threads = host_threads()
    parity = {"checked": 0, "note": "skipped (--no-parity)" if a.no_parity else ""}
    limiter = "hbm"
    unit_bytes = None                      # SURVEY 8d algorithmic bytes per unit of the workload

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
        if not a.no_parity and rank == 0:
            drv = _refdrv()
            if drv is not None:
                m = min(F, 512)
                drv.refdrv_mdct_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
                ssig = np.ascontiguousarray(host[:m])
                sfreq = np.zeros((m, 2, 960), np.int32)
                # the timed loop left rec = backward(freq) accumulated `steps + warm` times onto sig: check freq, and
                # one fresh backward on top of sig
                drv.refdrv_mdct_batch(_p(ssig), _p(sfreq), None, m * 2, 0, 1, sample_threads)
                srec = ssig.copy()
                drv.refdrv_mdct_batch(None, _p(sfreq), _p(srec), m * 2, 0, 2, sample_threads)
                rec2 = sig[:m].clone()
                ca.mdct_backward_batch(freq[:m].contiguous(), rec2, shift=0)
                torch.cuda.synchronize()
                if not (np.array_equal(freq[:m].cpu().numpy(), sfreq) and np.array_equal(rec2.cpu().numpy(), srec)):
                    raise SystemExit("PARITY FAILURE (mdct)")
                parity = {"checked": m, "note": "first %d frames of the last step: forward output + one backward vs clt_mdct_*_c of oracle/_ref" % m}
            else:
                parity["note"] = "oracle/_ref did not travel"
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
        ksum = (C.c_double * NK)()
        kcnt = (C.c_int * NK)()
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
        kname, kms = kern[0]["kernel"], kern[0]["avg_launch_ms"]
        unit_bytes = mean_len + 8 + PCM_BYTES                    # packet + (len, rng) in, PCM out
        kbytes = int(F * unit_bytes)
        limiter = "latency / VALU issue (lane per stream)"
        metric = "48kHz stereo 20ms CELT frames decoded/sec"
        workload = ("%d independent CELT-only 20 ms stereo packets per GPU (config #3's packets, mean %.1f B), each through a "
                    "fresh decoder (state reset included in the step), PCM bit-exact vs FIXED_POINT opus_decode()" % (F, mean_len))
        dtype = "int16/int32 fixed-point"
        extra = {"mean_packet_bytes": round(mean_len, 2), "kernels": kern, "state_bytes_per_stream": STATE}
        pk_h, ln_h = pk[:4096].cpu().numpy(), ln[:4096].cpu().numpy()
        cpu = (lambda: cpu_baseline_decode(pk_h, ln_h))
        if not a.no_parity and rank == 0:
            drv = _refdrv()
            if drv is not None:
                idx = np.arange(0, F, max(1, F // 1024))[:1024]
                t_idx = torch.from_numpy(idx).to(dev)
                spk = np.ascontiguousarray(pk[t_idx].cpu().numpy())
                sln = np.ascontiguousarray(ln[t_idx].cpu().numpy().astype(np.int32))
                m = len(idx)
                epcm = np.zeros((m, 960, 2), np.int16)
                erng = np.zeros(m, np.uint32)
                eret = np.zeros(m, np.int32)
                drv.refdrv_decode_frames(_p(spk), spk.shape[1], _p(sln), C.c_long(m), 1, _p(epcm), _p(erng), _p(eret), sample_threads)
                if not (np.array_equal(pcm[t_idx].cpu().numpy(), epcm)
                        and np.array_equal(dec.final_range[t_idx].cpu().numpy().view(np.uint32), erng)):
                    raise SystemExit("PARITY FAILURE (decode)")
                parity = {"checked": m, "note": "strided sample of the last step: PCM + final range vs opus_decode() of oracle/_ref"}
            else:
                parity["note"] = "oracle/_ref did not travel"
    elif a.workload == "silk_deldec":
        # silk_NSQ_del_dec (the quantizer of complexity >= 4)
        F = a.frames or 65536
        steps = a.steps or 10
        warm = a.warmup if a.warmup is not None else 2
        rec, rec_desc = silk_records(F, "dd", rank)
        di = torch.from_numpy(rec["dd_in"]).to(dev)
        st0 = torch.from_numpy(rec["dd_state_in"]).to(dev)
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
        limiter = "latency / VALU issue (per-sample recurrence)"
        metric = "SILK 16kHz mono 20ms frames/sec (silk_NSQ_del_dec records)"
        workload = "%s per GPU, silk_NSQ_del_dec, bit-exact vs FIXED_POINT" % rec_desc
        dtype = "int16/int32/int64 fixed-point"
        extra = {}
        m_cpu = min(F, 4096)
        cpu = (lambda: cpu_baseline_silk_dd(rec["dd_in"][:m_cpu], rec["dd_state_in"][:m_cpu]))
        if not a.no_parity and rank == 0:
            st.copy_(st0)                          # one more pass from the captured states, outside the clock
            ca.silk_NSQ_del_dec(di, st, do)
            torch.cuda.synchronize()
            if "dd_out" in rec:
                if not (np.array_equal(do.cpu().numpy(), rec["dd_out"]) and np.array_equal(st.cpu().numpy(), rec["dd_state_out"])):
                    raise SystemExit("PARITY FAILURE (silk_NSQ_del_dec)")
                parity = {"checked": F, "note": "every record (pulses, Seed, all of silk_nsq_state) vs the reference's own captured outputs"}
            else:
                parity["note"] = "capture library absent: tiled golden inputs, outputs not compared here"
    elif a.workload == "silk_lpc":
        F = a.frames or 65536
        steps = a.steps or 10
        warm = a.warmup if a.warmup is not None else 2
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        if not silk_corpus.available():
            raise SystemExit("silk_lpc needs oracle/_ref/libopus_ref_silkcap.so (records are captured from the reference encoder)")
        rec = {k: np.array(v) for k, v in silk_corpus.corpus(F, "lpc", seed=20260401 + 1000003 * rank).items()}
        li = torch.from_numpy(rec["lpc_in"]).to(dev)
        lo = torch.empty((F, 40), dtype=torch.uint8, device=dev)
        for _ in range(warm):
            ca.silk_find_LPC(li, lo)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.silk_find_LPC(li, lo)
            ev[k][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_find_lpc_kernel"
        kbytes = F * (832 + 40)
        limiter = "latency / VALU issue (serial recurrences per frame, divergent root search)"
        metric = "SILK 16kHz mono 20ms frames/sec (silk_find_LPC_FIX records)"
        workload = ("%d distinct records per GPU captured from the reference encoder (synthetic 16 kHz mono speech, 32 kb/s VOIP, "
                    "complexity 3/5/8/10 in turn), silk_find_LPC_FIX, bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {}
        m_cpu = min(F, 4096)
        cpu = (lambda: cpu_baseline_silk_lpc(np.ascontiguousarray(rec["lpc_in"][:m_cpu])))
        if not a.no_parity and rank == 0:
            if not np.array_equal(lo.cpu().numpy()[:, :36], rec["lpc_out"][:, :36]):
                raise SystemExit("PARITY FAILURE (silk_find_LPC)")
            parity = {"checked": F, "note": "every record (NLSF_Q15, NLSFInterpCoef_Q2) vs the reference's own captured outputs"}
    elif a.workload in ("silk_frames", "silk_frames_cbr"):
        cbr = a.workload == "silk_frames_cbr"      # constant bitrate: silk_encode_frame_FIX's bitrate loop runs on (almost) every frame
        F = a.frames or 65536
        steps = a.steps or 5
        warm = a.warmup if a.warmup is not None else 1
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS
        if not silk_corpus.available():
            raise SystemExit("silk_frames needs oracle/_ref/libopus_ref_silkcap.so (frames are captured from the reference encoder)")
        rec = {k: np.array(v) for k, v in silk_corpus.corpus(F, "chain_dd", seed=20260401 + 1000003 * rank,
                                                              variant="wb20cbr" if cbr else "wb20").items()}
        names = {"pitch_in": "c_pitch_in", "shape_in": "c_shape_in", "fpc_in": "c_fpc_in", "gains_in": "c_gains_in",
                 "prefilter_in": "c_prefilter_in", "q_in": "c_q_in", "bits_in": "c_bits_in"}
        host_in = {k: rec[v].copy() for k, v in names.items()}
        for name, (cls, fields) in CHAIN_FED_FIELDS.items():              # what the chain has to produce itself starts out as zero
            for f in fields:
                d = getattr(cls, f)
                host_in[name][:, d.offset:d.offset + d.size] = 0
        d_in = {k: torch.from_numpy(v).to(dev) for k, v in host_in.items()}
        pf0, nsq0, ec0 = (torch.from_numpy(rec[k]).to(dev) for k in ("c_prefilter_state_in", "c_q_state_in", "c_ec_in"))
        pf_st, nsq_st, ec_st = pf0.clone(), nsq0.clone(), ec0.clone()
        chain = SilkAnalysisChain(16, 4)
        rate0 = rate_st = None
        if cbr:
            from concentus_amd import silk as S
            ctl = np.zeros(F, dtype=np.dtype(S.RateCtl))
            fargs = rec["c_frame_args"].view(np.int32)
            ctl["condCoding"], ctl["maxBits"], ctl["useCBR"], ctl["nb_subfr"], ctl["frame_length"] = fargs[:, 0], fargs[:, 1], fargs[:, 2], 4, 320
            rate0 = torch.from_numpy(ctl.view(np.uint8).reshape(F, -1).copy()).to(dev)
            rate_st = rate0.clone()

        def one_step():
            pf_st.copy_(pf0)
            nsq_st.copy_(nsq0)
            ec_st.copy_(ec0)
            if cbr:
                rate_st.copy_(rate0)
            return chain.run(d_in["pitch_in"], d_in["shape_in"], d_in["fpc_in"], d_in["gains_in"], d_in["prefilter_in"], pf_st, d_in["q_in"],
                             nsq_st, True, bits_in=d_in["bits_in"], ec_state=ec_st, rate_ctl=rate_st)
        for _ in range(warm):
            one_step()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            res = one_step()
            ev[k][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_nsq_del_dec_kernel"           # the longest of the step's kernels; avg_launch_ms below is the WHOLE step (extra["note"])
        kbytes = F * (1408 + 324 + 4380 + 1116 + 1328)  # per frame: pitch buffer in, pulses + Seed out, the carried states and the coder rewritten
        limiter = "latency / VALU issue (seven lane-per-frame kernels of serial fixed-point recurrences; whole step timed, kernel = the longest)"
        metric = "SILK 16kHz mono 20ms frames/sec (analysis chain + silk_NSQ_del_dec + entropy coding, pitch buffer -> range-coder bytes)"
        if cbr:
            metric = "SILK 16kHz mono 20ms frames/sec, constant bitrate (silk_encode_frame_FIX after the VAD incl. its bitrate loop, pitch buffer -> range-coder bytes)"
        workload = ("%d frames per GPU, each with the records of ONE frame captured from the reference encoder (synthetic 16 kHz mono "
                    "speech, 32 kb/s VOIP, complexity 5/7/10 in turn); the seven kernels run back to back (find_pitch_lags, "
                    "noise_shape_analysis, find_pred_coefs, process_gains, prefilter, NSQ_del_dec, encode_indices + encode_pulses), records between them filled on "
                    "the device; bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {"note": "avg_launch_ms is the whole eight-kernel step (incl. the device-side record moves and the reset of the carried states and the coder)"}
        m_cpu = min(F, 2048)

        def cpu():
            sub = {k[2:]: np.ascontiguousarray(v[:m_cpu]) for k, v in rec.items() if k.startswith("c_") and not k.startswith("c_q")}
            a_ = cpu_baseline_silk_analysis(sub)
            b_ = cpu_baseline_silk_dd(np.ascontiguousarray(rec["c_q_in"][:m_cpu]), np.ascontiguousarray(rec["c_q_state_in"][:m_cpu]))
            c_ = cpu_baseline_silk_bits(np.ascontiguousarray(rec["c_bits_in"][:m_cpu]), np.ascontiguousarray(rec["c_ec_in"][:m_cpu]))
            v = 1.0 / (1.0 / a_["value"] + 1.0 / b_["value"] + 1.0 / c_["value"])
            port = {"value": round(v, 1), "unit": "frames/s", "cores": a_["cores"], "kind": "port", "cpu": a_["cpu"],
                    "sample": "analysis: " + a_["sample"] + "; quantiser: " + b_["sample"] + "; entropy coding: " + c_["sample"]
                              + "; combined as 1 / (1/a + 1/b + 1/c)"
                              + ("; ONE pass per frame (the loop's further quantiser + coder passes are not in this baseline)" if cbr else "")}
            # the baseline proper: the unmodified reference encoder at the same settings (it also does what surrounds the frame
            # function); the host build of the kernel sources stays beside it as "port"
            ref = cpu_baseline_silk_encoder(7, 0 if cbr else 1)
            if ref.get("value"):
                ref["port"] = port
                return ref
            return port
        if cbr:
            passes = rec["c_frame_args"].view(np.int32)[:, 3]
            workload += ("; CBR: after every pass one rate-control step, the frames over / under budget quantised + coded again from their "
                         "entry states (%.2f passes per frame on average, 1-7)" % float(passes.mean()))
            extra["passes_per_frame"] = round(float(passes.mean()), 3)
        if not a.no_parity and rank == 0:
            for key, want, nb in (("pitch_out", "c_pitch_out", 1380), ("shape_out", "c_shape_out", 380), ("fpc_out", "c_fpc_out", 204),
                                  ("gains_out", "c_gains_out", 52), ("prefilter_out", "c_prefilter_out", 1280)):
                if not np.array_equal(res[key].cpu().numpy()[:, :nb], rec[want][:, :nb]):
                    raise SystemExit("PARITY FAILURE (silk_frames: %s)" % key)
            if cbr:
                misc = rec["c_frame_misc"]
                got_ctl = rate_st.cpu().numpy().view(np.dtype(S.RateCtl))[:, 0]
                ok = (np.array_equal(res["pulses"].cpu().numpy().view(np.uint8), misc[:, :320])
                      and np.array_equal(res["Seed"].cpu().numpy(), misc[:, 328:332].copy().view(np.int32)[:, 0])
                      and np.array_equal(nsq_st.cpu().numpy(), rec["c_frame_nsq"]) and np.array_equal(pf_st.cpu().numpy(), rec["c_prefilter_state_out"])
                      and np.array_equal(ec_st.cpu().numpy(), rec["c_frame_ec"]) and np.array_equal(got_ctl["passes"], passes)
                      and np.array_equal(got_ctl["LastGainIndex"], misc[:, 324:328].copy().view(np.int32)[:, 0]))
                if not ok:
                    raise SystemExit("PARITY FAILURE (silk_frames_cbr: what silk_encode_frame_FIX leaves behind)")
                parity = {"checked": F, "note": "every frame: every stage's output record, and what silk_encode_frame_FIX leaves behind after its "
                                               "bitrate loop -- pulses, Seed, all of silk_nsq_state, LastGainIndex, the number of passes, the range "
                                               "coder (every field, every byte written) -- vs the reference"}
            ok = cbr or (np.array_equal(res["pulses"].cpu().numpy().view(np.uint8), rec["c_q_out"][:, :320])
                  and np.array_equal(res["Seed"].cpu().numpy(), rec["c_q_out"][:, 320:324].copy().view(np.int32)[:, 0])
                  and np.array_equal(nsq_st.cpu().numpy(), rec["c_q_state_out"]) and np.array_equal(pf_st.cpu().numpy(), rec["c_prefilter_state_out"])
                  and np.array_equal(ec_st.cpu().numpy(), rec["c_ec_out"]))
            if not ok:
                raise SystemExit("PARITY FAILURE (silk_frames: pulses / Seed / silk_nsq_state / prefilter state / range coder)")
            if not cbr:
                parity = {"checked": F, "note": "every frame: every stage's output record, pulses, Seed, all of silk_nsq_state and silk_prefilter_state_FIX, "
                                           "and the range coder (every field, every byte written) vs what the reference computed for the same frame"}
    elif a.workload == "silk_streams":
        NS = a.frames or 65536                     # streams per GPU
        T = 2                                      # consecutive frames per stream and step
        F = NS * T
        steps = a.steps or 3
        warm = a.warmup if a.warmup is not None else 1
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        from concentus_amd import silk as S
        from concentus_amd.silk_chain import SilkAnalysisChain, CHAIN_FED_FIELDS, CARRIED_FIELDS
        from test_silk_stream_gpu import stream_state_from_capture, zero_fields
        if not silk_corpus.available():
            raise SystemExit("silk_streams needs oracle/_ref/libopus_ref_silkcap.so (streams are captured from the reference encoder)")
        rec = {k: np.array(v) for k, v in silk_corpus.corpus(F, "chain_dd", seed=20260401 + 1000003 * rank, variant="wb20cbr", seg_frames=T).items()}
        names = {"pitch_in": "c_pitch_in", "shape_in": "c_shape_in", "fpc_in": "c_fpc_in", "gains_in": "c_gains_in",
                 "prefilter_in": "c_prefilter_in", "q_in": "c_q_in", "bits_in": "c_bits_in"}
        rows0 = np.arange(NS) * T
        chain = SilkAnalysisChain(16, 4)
        frames_dev = []
        for t in range(T):
            rows = rows0 + t
            host = {k: rec[v][rows].copy() for k, v in names.items()}
            SI = host["shape_in"].view(np.dtype(S.NoiseShapeIn))[:, 0]
            finp = np.ascontiguousarray(SI["x"][:, 160:160 + 320]).astype(np.int16)        # the frame's own samples (behind 2 x la_shape of history)
            zero_fields(host, CHAIN_FED_FIELDS)
            zero_fields(host, CARRIED_FIELDS)                                              # nothing inherited comes from the host
            ctl = np.zeros(NS, dtype=np.dtype(S.RateCtl))
            fargs = rec["c_frame_args"][rows].view(np.int32)
            ctl["condCoding"], ctl["maxBits"], ctl["useCBR"], ctl["nb_subfr"], ctl["frame_length"] = fargs[:, 0], fargs[:, 1], fargs[:, 2], 4, 320
            frames_dev.append({"in": {k: torch.from_numpy(v).to(dev) for k, v in host.items()}, "input": torch.from_numpy(finp).to(dev),
                               "ctl0": torch.from_numpy(ctl.view(np.uint8).reshape(NS, -1).copy()).to(dev),
                               "ec0": torch.from_numpy(rec["c_ec_in"][rows].copy()).to(dev)})
            frames_dev[-1]["ctl"], frames_dev[-1]["ec"] = frames_dev[-1]["ctl0"].clone(), frames_dev[-1]["ec0"].clone()
        st0 = torch.from_numpy(stream_state_from_capture(rec, rows0, S, True).view(np.uint8).reshape(NS, -1).copy()).to(dev)
        pf0, nsq0 = (torch.from_numpy(rec[k][rows0].copy()).to(dev) for k in ("c_prefilter_state_in", "c_q_state_in"))
        st, pf_st, nsq_st = st0.clone(), pf0.clone(), nsq0.clone()

        def one_step():
            st.copy_(st0); pf_st.copy_(pf0); nsq_st.copy_(nsq0)          # every step starts the streams again at their captured t = 0
            res = None
            for fd in frames_dev:
                fd["ctl"].copy_(fd["ctl0"]); fd["ec"].copy_(fd["ec0"])
                i = fd["in"]
                res = chain.run(i["pitch_in"], i["shape_in"], i["fpc_in"], i["gains_in"], i["prefilter_in"], pf_st, i["q_in"], nsq_st, True,
                                bits_in=i["bits_in"], ec_state=fd["ec"], rate_ctl=fd["ctl"], streams=st, frame_input=fd["input"])
            return res
        for _ in range(warm):
            one_step()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            res = one_step()
            ev[k][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_nsq_del_dec_kernel"
        kbytes = F * (640 + 324 + 1328)            # per frame: the samples in, pulses + Seed and the packet's coder out (the states stay on the device)
        limiter = "latency / VALU issue (lane-per-frame kernels of serial fixed-point recurrences; whole two-frame step timed)"
        metric = "SILK 16kHz mono 20ms frames/sec, streams mode (silk_encode_frame_FIX incl. its bitrate loop, inter-frame state carried on the device)"
        workload = ("%d streams per GPU x %d consecutive frames per step (16 kHz mono synthetic speech, 32 kb/s VOIP CBR, complexity 5/7/10 in turn): the state "
                    "a frame inherits is carried on the device from frame to frame, the capture of the reference supplies it only before the "
                    "first frame, and per frame what silk_encode_frame_FIX does not compute (samples, VAD results, maxBits, the packet's "
                    "coder); bit-exact vs FIXED_POINT" % (NS, T))
        dtype = "int16/int32/int64 fixed-point"
        extra = {"note": "avg_launch_ms is the whole step: %d frames of every stream, each the chain + bitrate loop + carry kernels" % T,
                 "streams": NS, "frames_per_stream_and_step": T}
        cpu = cpu_baseline_silk_encoder
        if not a.no_parity and rank == 0:
            rows = rows0 + (T - 1)
            misc = rec["c_frame_misc"][rows]
            got_ctl = frames_dev[-1]["ctl"].cpu().numpy().view(np.dtype(S.RateCtl))[:, 0]
            ok = (np.array_equal(res["pulses"].cpu().numpy().view(np.uint8), misc[:, :320])
                  and np.array_equal(nsq_st.cpu().numpy(), rec["c_frame_nsq"][rows]) and np.array_equal(pf_st.cpu().numpy(), rec["c_prefilter_state_out"][rows])
                  and np.array_equal(frames_dev[-1]["ec"].cpu().numpy(), rec["c_frame_ec"][rows])
                  and np.array_equal(frames_dev[0]["ec"].cpu().numpy(), rec["c_frame_ec"][rows0])
                  and np.array_equal(got_ctl["passes"], rec["c_frame_args"][rows].view(np.int32)[:, 3]))
            if not ok:
                raise SystemExit("PARITY FAILURE (silk_streams: the last frame of the streams, whose inherited state only the device supplied)")
            parity = {"checked": F, "note": "both frames of every stream: the range coder after the frame (every field, every payload byte); the "
                                           "second frame also pulses, all of silk_nsq_state / silk_prefilter_state_FIX and the number of passes -- "
                                           "its inherited state came from the device only"}
    elif a.workload == "silk_analysis":
        F = a.frames or 65536
        steps = a.steps or 5
        warm = a.warmup if a.warmup is not None else 1
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        if not silk_corpus.available():
            raise SystemExit("silk_analysis needs oracle/_ref/libopus_ref_silkcap.so (records are captured from the reference encoder)")
        rec = {}
        for kind, *_ in ANALYSIS_OPS:
            rec.update({k: np.array(v) for k, v in silk_corpus.corpus(F, kind, seed=20260401 + 1000003 * rank).items()})
        d_in = {ik: torch.from_numpy(rec[ik]).to(dev) for _, ik, *_ in ANALYSIS_OPS}
        d_out = {ok: torch.empty((F, rec[ok].shape[1]), dtype=torch.uint8, device=dev) for _, _, ok, *_ in ANALYSIS_OPS}
        pf_st0 = torch.from_numpy(rec["prefilter_state_in"]).to(dev)
        pf_st = pf_st0.clone()

        def run_all(events=None):
            for j, (kind, ik, ok, _, _, op, _) in enumerate(ANALYSIS_OPS):
                if events is not None:
                    events[j].record()
                if kind == "prefilter":
                    pf_st.copy_(pf_st0)
                    ca.silk_prefilter(d_in[ik], pf_st, d_out[ok])
                else:
                    getattr(ca, op)(d_in[ik], d_out[ok])
            if events is not None:
                events[len(ANALYSIS_OPS)].record()
        for _ in range(warm):
            run_all()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(ANALYSIS_OPS) + 1)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            run_all(ev[k])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        per = [float(np.mean([e[j].elapsed_time(e[j + 1]) for e in ev])) for j in range(len(ANALYSIS_OPS))]
        kern = [{"kernel": ANALYSIS_OPS[j][6], "avg_launch_ms": round(per[j], 4),
                 "bytes_per_record": int(rec[ANALYSIS_OPS[j][1]].shape[1] + rec[ANALYSIS_OPS[j][2]].shape[1])} for j in range(len(ANALYSIS_OPS))]
        jmax = int(np.argmax(per))
        kms, kname = per[jmax], ANALYSIS_OPS[jmax][6]
        kbytes = F * kern[jmax]["bytes_per_record"]
        limiter = "latency / VALU issue (serial fixed-point recurrences, one lane per frame; scratch-resident work arrays)"
        metric = "SILK 16kHz mono 20ms frames/sec (analysis chain: find_pitch_lags + noise_shape_analysis + find_pred_coefs + process_gains + prefilter)"
        workload = ("%d distinct records per function and per GPU captured from the reference encoder (synthetic 16 kHz mono speech, 32 kb/s "
                    "VOIP, complexity 3/5/8/10 in turn): the five analysis calls silk_encode_frame_FIX makes between the VAD and the "
                    "noise-shaping quantiser, each bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {"kernels": kern}
        m_cpu = min(F, 2048)
        cpu = (lambda: cpu_baseline_silk_analysis({k: np.ascontiguousarray(v[:m_cpu]) for k, v in rec.items()}))
        if not a.no_parity and rank == 0:
            for kind, ik, ok, nb, _, op, _ in ANALYSIS_OPS:
                if not np.array_equal(d_out[ok].cpu().numpy()[:, :nb], rec[ok][:, :nb]):
                    raise SystemExit("PARITY FAILURE (%s)" % op)
            if not np.array_equal(pf_st.cpu().numpy(), rec["prefilter_state_out"]):
                raise SystemExit("PARITY FAILURE (silk_prefilter state)")
            parity = {"checked": F * len(ANALYSIS_OPS), "note": "every record of all five functions (every field written, and the prefilter state) vs the reference's own captured outputs"}
    elif a.workload == "silk_pred":
        F = a.frames or 65536
        steps = a.steps or 10
        warm = a.warmup if a.warmup is not None else 2
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        if not silk_corpus.available():
            raise SystemExit("silk_pred needs oracle/_ref/libopus_ref_silkcap.so (records are captured from the reference encoder)")
        rec = {k: np.array(v) for k, v in silk_corpus.corpus(F, "fpc", seed=20260401 + 1000003 * rank).items()}
        fi = torch.from_numpy(rec["fpc_in"]).to(dev)
        fo = torch.empty((F, 208), dtype=torch.uint8, device=dev)
        for _ in range(warm):
            ca.silk_find_pred_coefs(fi, fo)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.silk_find_pred_coefs(fi, fo)
            ev[k][1].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_find_pred_coefs_kernel"
        kbytes = F * (2688 + 208)
        limiter = "latency / VALU issue (serial recurrences per frame: LDL solve, codebook searches, Burg, NLSF trellis)"
        metric = "SILK 16kHz mono 20ms frames/sec (silk_find_pred_coefs_FIX records)"
        workload = ("%d distinct records per GPU captured from the reference encoder (synthetic 16 kHz mono speech, 32 kb/s VOIP, "
                    "complexity 3/5/8/10 in turn, voiced and unvoiced frames), silk_find_pred_coefs_FIX whole, bit-exact vs "
                    "FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {}
        m_cpu = min(F, 4096)
        cpu = (lambda: cpu_baseline_silk_pred(np.ascontiguousarray(rec["fpc_in"][:m_cpu])))
        if not a.no_parity and rank == 0:
            if not np.array_equal(fo.cpu().numpy()[:, :204], rec["fpc_out"][:, :204]):
                raise SystemExit("PARITY FAILURE (silk_find_pred_coefs)")
            parity = {"checked": F, "note": "every record, every field silk_find_pred_coefs_FIX writes, vs the reference's own captured outputs"}
    elif a.workload == "silk_nlsf":
        F = a.frames or 65536
        steps = a.steps or 10
        warm = a.warmup if a.warmup is not None else 2
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import silk_corpus
        if not silk_corpus.available():
            raise SystemExit("silk_nlsf needs oracle/_ref/libopus_ref_silkcap.so (records are captured from the reference encoder)")
        rec = {k: np.array(v) for k, v in silk_corpus.corpus(F, "pred", seed=20260401 + 1000003 * rank).items()}
        ni, ei = torch.from_numpy(rec["nlsf_in"]).to(dev), torch.from_numpy(rec["resnrg_in"]).to(dev)
        no = torch.empty((F, 120), dtype=torch.uint8, device=dev)
        eo = torch.empty((F, 40), dtype=torch.uint8, device=dev)
        for _ in range(warm):
            ca.silk_process_NLSFs(ni, no)
            ca.silk_residual_energy(ei, eo)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            ca.silk_process_NLSFs(ni, no)
            ev[k][1].record()
            ca.silk_residual_energy(ei, eo)
            ev[k][2].record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        kms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        kname = "silk_process_nlsfs_kernel"
        kbytes = F * (96 + 120)
        limiter = "latency / VALU issue (32-entry insertion sort and a 16-step 4-state trellis per survivor, one lane per record)"
        metric = "SILK 16kHz mono 20ms frames/sec (silk_process_NLSFs + silk_residual_energy_FIX records)"
        workload = ("%d distinct records per GPU captured from the reference encoder (synthetic 16 kHz mono speech, 32 kb/s VOIP, "
                    "complexity 3/5/8/10 in turn), silk_process_NLSFs then silk_residual_energy_FIX (the tail of "
                    "silk_find_pred_coefs_FIX), bit-exact vs FIXED_POINT" % F)
        dtype = "int16/int32/int64 fixed-point"
        extra = {"residual_energy_kernel_ms": round(float(np.mean([e[1].elapsed_time(e[2]) for e in ev])), 4)}
        m_cpu = min(F, 4096)
        cpu = (lambda: cpu_baseline_silk_nlsf(np.ascontiguousarray(rec["nlsf_in"][:m_cpu]), np.ascontiguousarray(rec["resnrg_in"][:m_cpu])))
        if not a.no_parity and rank == 0:
            if not (np.array_equal(no.cpu().numpy()[:, :116], rec["nlsf_out"][:, :116])
                    and np.array_equal(eo.cpu().numpy()[:, :32], rec["resnrg_out"][:, :32])):
                raise SystemExit("PARITY FAILURE (silk_process_NLSFs / silk_residual_energy_FIX)")
            parity = {"checked": F, "note": "every record (NLSFIndices, quantised NLSFs, PredCoef_Q12; nrgs, nrgsQ) vs the reference's own captured outputs"}
    elif a.workload == "silk":
        F = a.frames or 65536
        steps = a.steps or 20
        warm = a.warmup if a.warmup is not None else 3
        rec, rec_desc = silk_records(F, "nsq", rank)
        bi, ni, st0 = (torch.from_numpy(rec[k]).to(dev) for k in ("burg_in", "nsq_in", "nsq_state_in"))
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
        limiter = "latency / VALU issue (per-sample recurrence)"
        metric = "SILK 16kHz mono 20ms frames/sec (silk_burg_modified + silk_NSQ records, config #4)"
        workload = "configs[3]: %s per GPU, silk_burg_modified + silk_NSQ, bit-exact vs FIXED_POINT" % rec_desc
        dtype = "int16/int32/int64 fixed-point"
        extra = {"burg_kernel_ms": round(burg_ms, 5), "burg_GBps": round(F * 856 / (burg_ms * 1e-3) / 1e9, 2)}
        m_cpu = min(F, 8192)
        cpu = (lambda: cpu_baseline_silk(rec["burg_in"][:m_cpu], rec["nsq_in"][:m_cpu], rec["nsq_state_in"][:m_cpu]))
        if not a.no_parity and rank == 0:
            st.copy_(st0)
            ca.silk_NSQ(ni, st, pulses)
            torch.cuda.synchronize()
            if "nsq_out" in rec:
                ok = (np.array_equal(bo.cpu().numpy(), rec["burg_out"])
                      and np.array_equal(pulses.cpu().numpy().view(np.uint8), rec["nsq_out"])
                      and np.array_equal(st.cpu().numpy(), rec["nsq_state_out"]))
                if not ok:
                    raise SystemExit("PARITY FAILURE (silk)")
                parity = {"checked": F, "note": "every record (A_Q16 / res_nrg, pulses, all of silk_nsq_state) vs the reference's own captured outputs"}
            else:
                parity["note"] = "capture library absent: tiled golden inputs, outputs not compared here"
    elif a.workload == "celt_streams":
        # streams mode (SURVEY 8d "Definition of independent frame", 8f row 1): S streams, one frame per stream per step
        F = a.frames or 65536
        steps = a.steps or 8
        warm = a.warmup if a.warmup is not None else 2
        T = steps + warm
        cfgvals = (2, 96000, 1, 0, 10, 16, 0, 1500)
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        pcm_all = torch.randint(-8192, 8192, (T, F, 960, 2), generator=g, dtype=torch.int16, device=dev)
        enc = ca.OpusEncoderBatch(F, device=dev).apply_opus_demo_ctls(96000, 1, 0, 10)
        for t in range(warm):
            out, lens = enc.encode(pcm_all[t])
        torch.cuda.synchronize()
        L.opusgpu_kernel_timing_enable(1)
        barrier()
        t0 = time.perf_counter()
        for t in range(warm, T):
            out, lens = enc.encode(pcm_all[t])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        _r = enc.final_range
        kern, mean_len = celt_kernel_table(ca, L, F, steps, lens, traffic_db)
        dom = kern[0]
        kname, kms = dom["kernel"], dom["avg_launch_ms"]
        unit_bytes = PCM_BYTES + mean_len + 8
        kbytes = int(dom["frames_per_launch"] * unit_bytes)
        limiter = "latency / VALU issue (lane per frame)"
        metric = "48kHz stereo 20ms CELT frames encoded/sec (streams mode)"
        workload = ("streams mode of configs[2]: %d streams per GPU x %d consecutive 48 kHz stereo 20 ms frames (%d warm-up + %d "
                    "timed; one frame of every stream per step, encoder state 9 024 B/stream in HBM), full CELT encode 96 kb/s "
                    "VBR complexity 10, packets bit-exact vs FIXED_POINT opus_encode(); mean packet %.1f B" % (F, T, warm, steps, mean_len))
        dtype = "int16/int32 fixed-point"
        extra = {"mean_packet_bytes": round(mean_len, 2), "kernels": kern, "realtime_factor": None}
        sidx = np.arange(0, F, max(1, F // 256))[:256]
        t_sidx = torch.from_numpy(sidx).to(dev)
        spcm = np.ascontiguousarray(pcm_all[:, t_sidx].permute(1, 0, 2, 3).cpu().numpy().reshape(len(sidx) * T, 960, 2))
        cpu = (lambda: cpu_baseline_celt(spcm, cfgvals, fps=T))
        if not a.no_parity and rank == 0:
            if _refdrv() is not None:
                epk, eln, erg = ref_encode(spcm, cfgvals, T, sample_threads)
                last = np.arange(len(sidx)) * T + (T - 1)
                check_packets("celt_streams", out[t_sidx].cpu().numpy(), lens[t_sidx].cpu().numpy(),
                              _r[t_sidx].cpu().numpy().view(np.uint32), epk[last], eln[last], erg[last])
                parity = {"checked": len(sidx), "note": "frame %d (the last timed one) of a strided sample of streams: packet bytes + "
                          "final range vs opus_encode() of oracle/_ref run over the same %d-frame streams" % (T - 1, T)}
            else:
                parity["note"] = "oracle/_ref did not travel"
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
        cfgvals = (2, 96000, 1, 0, 10, 16, 0, 1500)
        rng = np.random.default_rng((5 if mixed else 3) + rank)      # seed 3 (config #3) / 5 (config #5), uniform int16 in [-8192, 8191]
        host = rng.integers(-8192, 8192, size=(F, 960, 2), dtype=np.int16)
        pcm = torch.from_numpy(host).to(dev)
        out = lens = None
        if mixed:
            srec, srec_desc = silk_records(NS, "nsq", rank)
            s_bi, s_ni, s_st0 = (torch.from_numpy(srec[k]).to(dev) for k in ("burg_in", "nsq_in", "nsq_state_in"))
            s_st = s_st0.clone()
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
        from concentus_amd.sharding import gather_packets, trim_width, truncated
        # row width of the packet exchange: learnt ONCE from the warm-up output (one all_reduce + host read, before the
        # clock); the timed steps pass it as a number, so no step synchronises with the host for it. The lengths travel
        # with the rows and are checked after the clock (a longer packet would have been cut).
        gather_w = trim_width(lens if host_gather is False else lens.cpu(), out.shape[1]) if (world > 1 and not a.no_gather and warm > 0) else False
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
            L.opusgpu_kernel_timing_read((C.c_double * 8)(), (C.c_int * 8)(), 8)
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
                    pg.wait()
                hc = (lambda t: t.cpu()) if host_gather else (lambda t: t)
                pending = [gather_packets(hc(out), hc(lens), hc(_r), world, sizes=[F] * world, trim=gather_w, async_op=True)]
                if mixed:
                    pending.append(gather_packets(hc(s_pulses), hc(s_bo), hc(s_bo[:, :4]).contiguous(), world, sizes=[NS] * world, async_op=True))
        gathered = [pg.wait() for pg in pending]
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        if gathered and gathered[0] is not None and gather_w and truncated(gathered[0][1], pending[0].width):
            raise SystemExit("packet gather: a packet of the last step is longer than the row width learnt in the warm-up")
        kern, mean_len = celt_kernel_table(ca, L, F, steps, lens, traffic_db)
        dom = kern[0]
        kname, kms = dom["kernel"], dom["avg_launch_ms"]
        unit_bytes = PCM_BYTES + mean_len + 8      # PCM in + packet + (len, rng) out
        kbytes = int(dom["frames_per_launch"] * unit_bytes)
        limiter = "latency / VALU issue (lane per frame)"
        metric = "48kHz stereo 20ms CELT frames encoded/sec"
        workload = ("configs[2]: %d independent 48 kHz stereo 20 ms frames per GPU (each the first frame of its own stream), full "
                    "CELT encode (MDCT + PVQ + range enc) 96 kb/s VBR complexity 10, packets bit-exact vs FIXED_POINT "
                    "opus_encode(); mean packet %.1f B" % (F, mean_len))
        dtype = "int16/int32 fixed-point"
        extra = {"mean_packet_bytes": round(mean_len, 2), "frames_per_launch": dom["frames_per_launch"],
                 "kernels": kern, "realtime_factor": None}
        cpu = (lambda: cpu_baseline_celt(host[:4096], cfgvals))
        if not a.no_parity and rank == 0:
            if _refdrv() is not None:
                idx = np.arange(0, F, max(1, F // 2048))[:2048]
                t_idx = torch.from_numpy(idx).to(dev)
                epk, eln, erg = ref_encode(host[idx], cfgvals, 1, sample_threads)
                check_packets(a.workload, out[t_idx].cpu().numpy(), lens[t_idx].cpu().numpy(),
                              _r[t_idx].cpu().numpy().view(np.uint32), epk, eln, erg)
                parity = {"checked": len(idx), "note": "strided sample of the last timed step: packet bytes + final range vs "
                          "opus_encode() of oracle/_ref"}
            else:
                parity["note"] = "oracle/_ref did not travel"
        if mixed:
            if not a.no_parity and rank == 0 and "nsq_out" in srec:
                s_st.copy_(s_st0)
                ca.silk_NSQ(s_ni, s_st, s_pulses)
                torch.cuda.synchronize()
                ok = (np.array_equal(s_bo.cpu().numpy(), srec["burg_out"])
                      and np.array_equal(s_pulses.cpu().numpy().view(np.uint8), srec["nsq_out"])
                      and np.array_equal(s_st.cpu().numpy(), srec["nsq_state_out"]))
                if not ok:
                    raise SystemExit("PARITY FAILURE (mixed: SILK records)")
                parity["checked"] += NS
                parity["note"] += "; all %d SILK records vs the reference's captured outputs" % NS
            metric = "mixed CELT/SILK 20ms frames processed/sec (configs[4]: 7/8 CELT encode, 1/8 SILK burg+NSQ records)"
            workload = ("configs[4]: %d units per GPU = %d independent 48 kHz stereo CELT frames (full encode, 96 kb/s VBR "
                        "complexity 10, mean packet %.1f B) + %s (silk_burg_modified + silk_NSQ), "
                        "bit-exact vs FIXED_POINT; the SILK kernels run on a side stream" % (FT, F, mean_len, srec_desc))
            extra["celt_frames_per_gpu"], extra["silk_records_per_gpu"] = F, NS
            extra.pop("realtime_factor", None)
            F = FT

    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_gather else dev)
    per_rank_s = [elapsed]
    if world > 1:
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                  # what each rank's clock read: shows a slow rank in the line
        per_rank_s = [float(e.item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    dist_info = {"backend": (dist.get_backend() if world > 1 else None), "dist_world_size": (dist.get_world_size() if world > 1 else 1),
                 "device": torch.cuda.get_device_name(local)}
    achieved = kbytes / (kms * 1e-3) / 1e9

    if rank == 0:
        value = F * world * steps / elapsed
        if "realtime_factor" in extra:
            extra["realtime_factor"] = round(value * 0.02, 1)        # 20 ms of audio per frame
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic_db.get(kname),
                "algorithmic_bytes_per_launch": kbytes, "avg_launch_ms": round(kms, 5), "limiter": limiter}
        if unit_bytes is not None:
            roof["algorithmic_bytes_per_unit"] = round(unit_bytes, 1)
            roof["pipeline_achieved"] = round(value / world * unit_bytes / 1e9, 2)     # whole step, per GPU
            roof["pipeline_frac"] = round(value / world * unit_bytes / 1e9 / HBM_PEAK_GBS, 6)
        pm = pmc_db.get(kname)
        if pm and pm.get("SQ_INSTS_VALU"):
            # VALU-issue roofline of the dominant kernel: wave-instructions x 4 cycles each / (SIMDs x cycles of the launch);
            # instruction count from the committed PMC pass (profiles/pmc.json, same batch size), duration measured live
            scale = (extra.get("frames_per_launch") or pm.get("units_per_launch") or 1) / float(pm.get("units_per_launch") or 1)
            roof["valu_issue_frac"] = round(pm["SQ_INSTS_VALU"] * scale * 4.0 / (SIMDS * CLOCK_HZ * kms * 1e-3), 4)
            roof["valu_wave_insts_per_launch"] = int(pm["SQ_INSTS_VALU"] * scale)
            if pm.get("SQ_THREAD_CYCLES_VALU") and pm.get("SQ_ACTIVE_INST_VALU"):
                # lanes doing work per issued vector instruction / 64: thread-cycles over instruction-cycles of the same
                # PMC pass (tools/prof_celt.sh pmc3), both in quad-cycles. valu_issue_frac counts a wave-instruction as issued whatever
                # its EXEC mask holds; the product of the two is the share of the machine's vector lane-slots doing useful work.
                roof["valu_lane_util"] = round(pm["SQ_THREAD_CYCLES_VALU"] / (64.0 * pm["SQ_ACTIVE_INST_VALU"]), 4)
                roof["valu_useful_lane_frac"] = round(roof["valu_issue_frac"] * roof["valu_lane_util"], 4)
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
            "config": {"workload": workload, "frames_per_gpu": F,
                       # the SILK workloads are 16 kHz mono records / frames; mdct / celt / decode / mixed carry 48 kHz stereo
                       "channels": 1 if a.workload.startswith("silk") else 2,
                       "sharding": ("records block-partitioned across ranks, no data-path collective, nothing gathered (the outputs stay on their rank)"
                                    if a.workload.startswith("silk") or a.workload in ("mdct", "decode") else
                                    "frames block-partitioned across ranks, no data-path collective; packets gathered to rank 0")},
            **({"rehearsal": True} if (host_gather or one_device) else {}),
            # transport evidence for a multi-GPU record: which torch.distributed backend ran ("nccl" = RCCL), how many ranks
            # it saw, and every rank's own rate over the timed region (units/s from that rank's clock)
            **dist_info,
            "per_rank_units_per_s": [round(F * steps / e, 1) for e in per_rank_s],
            "parity_checked": parity["checked"],
            "parity_note": parity["note"],
            "roofline": dict(roof, **extra),
        }
        if not a.no_cpu_baseline:
            try:
                out_line["cpu_baseline"] = cpu()
                if out_line["cpu_baseline"].get("value"):
                    out_line["vs_cpu_baseline"] = round(value / out_line["cpu_baseline"]["value"], 2)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out_line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference",
                                            "sample": "failed: %r" % (e,)}
        print(json.dumps(out_line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def celt_kernel_table(ca, L, F, steps, lens, traffic_db):
    """Per-kernel averages of the CELT encode pipeline over the timed region (HIP events on the launch stream,
    opusgpu_kernel_timing_*), sorted by time; and the mean packet length of the last step."""
    KNAMES = ["celt_front_kernel", "celt_back_kernel", "celt_back_lane_kernel", "celt_dc_reject_kernel",
              "celt_front1_kernel",
              "celt_transient_kernel" if os.environ.get("OPUSGPU_TRANSIENT_LANE") else "celt_transient_tile_kernel",
              "celt_front2_kernel"]
    NK = len(KNAMES)
    ksum = (C.c_double * NK)()
    kcnt = (C.c_int * NK)()
    ca.lib.check(L.opusgpu_kernel_timing_read(ksum, kcnt, NK), "opusgpu_kernel_timing_read")
    L.opusgpu_kernel_timing_enable(0)
    lens_h = lens.cpu().numpy()
    assert (lens_h > 0).all(), "encoder reported an error"
    mean_len = float(lens_h.mean())
    # hand-off bytes per frame and kernel: what each kernel must read and write once
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
                         "handoff_bytes_per_launch": int(fpl * per_frame[nm]),
                         "handoff_GBps": round(fpl * per_frame[nm] / (avg * 1e-3) / 1e9, 2), "traffic": traffic_db.get(nm)})
    kern.sort(key=lambda k: -k["avg_launch_ms"])
    return kern, mean_len


if __name__ == "__main__":
    sys.exit(main())
