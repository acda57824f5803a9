"""GPU tier: the N-rank path of bench.py with the real kernels, rehearsed on ONE GPU -- two ranks started by bench.py itself
(no launcher), both computing on cuda:0, the packet gather through gloo on host copies (RCCL wants one device per rank, so the
RCCL transport itself is only exercised by the driver's multi-GPU run). Checks what the one-rank runs cannot: the shard seeds,
the per-rank input generation (two ranks building their SILK corpora at the same time), the gather of both workloads, the
max-over-ranks timing and the post-clock parity check under WORLD_SIZE 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(CONCENTUS_BENCH_BACKEND="gloo", CONCENTUS_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-3000:]
    return json.loads(lines[-1])


def test_two_ranks_celt_shards_on_one_gpu():
    d = _bench(["--gpus", "2", "--frames", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["parity_checked"] >= 1024
    assert d["value"] > 0 and d["config"]["frames_per_gpu"] == 4096


def test_two_ranks_mixed_shards_on_one_gpu():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libopus_ref_silkcap.so")):
        pytest.skip("capture library did not travel")
    d = _bench(["--gpus", "2", "--workload", "mixed", "--frames", "8192", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True
    assert d["roofline"]["silk_records_per_gpu"] == 1024 and d["parity_checked"] >= 1024 + 1024
