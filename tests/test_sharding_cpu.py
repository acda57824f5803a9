"""CPU tier, world_size 2 (gloo): the multi-GPU layout of the frame path -- block partition of the
frame range, every rank encodes only its shard (here: with the host-emulated kernel sources), packets
gathered to rank 0 -- reproduces the single-process result (= the golden packets)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_range_partitions_exactly():
    from concentus_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_mixed_corpus_counts():
    """configs[4]: 1 M units over 8 ranks -> 131 072 per rank = 114 688 CELT frames + 16 384 SILK records."""
    from concentus_amd.sharding import mixed_counts, shard_range
    lo, hi = shard_range(1 << 20, 3, 8)
    assert hi - lo == 131072 and mixed_counts(hi - lo) == (114688, 16384)
    for n in (0, 1, 7, 8, 9, 65535):
        c, s = mixed_counts(n)
        assert c + s == n and c == n * 7 // 8
    with pytest.raises(ValueError):
        mixed_counts(-1)


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emulib
    import encode_cases as ec
    from concentus_amd.sharding import gather_packets, shard_range, trim_width, truncated
    pcm, pk, ln, rg = ec.load_case("noise_vbr_indep")          # 24 frames; world 2 -> 12 + 12, world 5 -> uneven
    n = pcm.shape[0] - 1                                        # 23: uneven split
    lo, hi = shard_range(n, rank, world)
    emu = emulib.lib()
    cfg = emulib.Config(2, 96000, 1, 0, 10, 16, 0, 1500)
    m = hi - lo
    out = np.zeros((m, 1280), np.uint8)
    lens = np.zeros(m, np.int32)
    rng = np.zeros(m, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    shard = np.ascontiguousarray(pcm[lo:hi])
    emu.emu_celt_encode_frames(C.byref(cfg), None, p(shard), m, 1, p(out), 1280, p(lens), p(rng))
    t_out, t_lens, t_rng = torch.from_numpy(out), torch.from_numpy(lens), torch.from_numpy(rng.view(np.int32))
    res = gather_packets(t_out, t_lens, t_rng, world)
    # the bench's form: shard sizes known up front, rows trimmed to the longest packet, exchange left pending
    sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    pending = gather_packets(t_out, t_lens, t_rng, world, sizes=sizes, trim=True, async_op=True)
    res2 = pending.wait()
    # ... and the steady-state form: the width learnt once (one collective + host read), then passed as a number so that
    # the exchange itself synchronises with nothing; a width that is too small must be detectable from the lengths
    w = trim_width(t_lens, t_out.shape[1])
    p3 = gather_packets(t_out, t_lens, t_rng, world, sizes=sizes, trim=w, async_op=True)
    res3 = p3.wait()
    p4 = gather_packets(t_out, t_lens, t_rng, world, sizes=sizes, trim=64, async_op=True)
    res4 = p4.wait()
    if rank == 0:
        try:
            assert p3.width == w == pending.width and w % 16 == 0 and w >= int(ln[:n].max()) and not truncated(res3[1], p3.width)
            assert p4.width == 64 and res4[0].shape[1] == 64 and truncated(res4[1], p4.width)
            for what, rr in (("sharded", res), ("sharded, trimmed + async", res2), ("sharded, width given", res3)):
                o, l, r = (t.numpy() for t in rr)
                assert o.shape[0] == n and (o.shape[1] == 1280 if what == "sharded" else o.shape[1] >= int(ln[:n].max()))
                ec.assert_packets_equal(o, l, r.view(np.uint32), pk[:n], ln[:n], rg[:n], what)
            q.put("ok")
        except AssertionError as e:
            q.put("FAIL: %s" % e)
    else:
        assert res is None and res2 is None and res3 is None and res4 is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_sharded_encode_matches_golden(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert q.get(timeout=10) == "ok"


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start the ranks itself (child processes, before any GPU
    call) and report n_gpus 2. --rehearse keeps it CPU-only: gloo rendezvous + the packet gather, no codec work."""
    rc, line, err = _run_bench(["--gpus", "2", "--rehearse"])
    assert rc == 0, err[-2000:]
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["gather_ok"] is True and line["value"] is None


def test_bench_rejects_a_world_size_that_differs_from_gpus():
    rc, line, err = _run_bench(["--gpus", "2", "--rehearse"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and line is None and "WORLD_SIZE=1" in err
