"""Multi-GPU layout of the frame path: one process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI on MI355X), frames block-partitioned across ranks, NO collective on the data path; the only
exchange is the final packet gather to rank 0. Works with the "gloo" backend on CPU
tensors too, which is how the logic is tested without GPUs."""


def shard_range(n_frames, rank, world):
    """Contiguous block partition [lo, hi) of the frame index range; the first n % world ranks get one
    extra frame. Frames are independent units, so any partition is valid; contiguous blocks keep the
    packet slab of a rank contiguous for the gather."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def gather_packets(out, lens, rng, world=None, dst=0):
    """Gather the per-rank packet slabs / lengths / final ranges to rank `dst`.
    Shards may differ in size by one frame, so every rank pads to the largest shard first.
    Returns (out, lens, rng) concatenated in rank order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = world or dist.get_world_size()
    rank = dist.get_rank()
    n_local = torch.tensor([out.shape[0]], dtype=torch.int64, device=out.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)

    def pad(t):
        if t.shape[0] == n_max:
            return t.contiguous()
        p = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[:t.shape[0]] = t
        return p

    results = []
    for t in (out, lens, rng):
        t = pad(t)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
        if rank == dst:
            results.append(torch.cat([b[:sizes[r]] for r, b in enumerate(bufs)], dim=0))
    return tuple(results) if rank == dst else None


def mixed_counts(n_units, silk_eighths=1):
    """BASELINE configs[4] / SURVEY 8d config #5: a shard of n_units mixed units holds 7/8 CELT frames and 1/8 SILK
    records (silk_eighths / 8 in general). Returns (n_celt, n_silk) with n_celt + n_silk == n_units."""
    if n_units < 0 or not 0 <= silk_eighths <= 8:
        raise ValueError("mixed_counts")
    n_celt = n_units * (8 - silk_eighths) // 8
    return n_celt, n_units - n_celt
