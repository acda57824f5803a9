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


class PendingGather:
    """Handle of a gather started with async_op=True: wait() blocks the current stream on the exchange and returns
    (out, lens, rng) on the destination rank (None elsewhere). The exchange runs on the backend's own stream, so kernels
    launched between gather_packets(...) and wait() overlap it."""

    width = None                    # row width of the gathered slab (after trimming), set by gather_packets

    def __init__(self, works, finish):
        self._works, self._finish = works, finish

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        return self._finish()


def gather_packets(out, lens, rng, world=None, dst=0, sizes=None, trim=False, async_op=False):
    """Gather the per-rank packet slabs / lengths / final ranges to rank `dst`.
    Shards may differ in size by one frame, so every rank pads to the largest shard first. `sizes` = the shard sizes of
    all ranks when the caller knows them (they follow from shard_range); otherwise they are exchanged first.
    `trim`: send only the first W bytes of every packet row (rounded up to 16) instead of the whole stride -- a VBR slab is
    mostly padding (mean packet ~320 B in a 1500-byte row). The gathered slab then has that trimmed row width.
      trim=True   W = the longest packet of this exchange over all ranks: one all_reduce(MAX) and one device->host read --
                  a host synchronisation per call, meant for the first exchange of a job (see trim_width());
      trim=<int>  W given by the caller -- the CBR packet size, or trim_width() of an earlier exchange: NO host read, nothing
                  on the critical path of the step. The lengths travel with the rows, so the receiver can always tell whether a
                  packet was longer than W (PendingGather.truncated / the `lens` returned): a caller that re-uses an earlier
                  width on new input must check that and repeat the exchange of that step with trim=False.
    Returns (out, lens, rng) concatenated in rank order on `dst`, None elsewhere; with async_op=True a PendingGather."""
    import torch
    import torch.distributed as dist
    world = world or dist.get_world_size()
    rank = dist.get_rank()
    if sizes is None:
        n_local = torch.tensor([out.shape[0]], dtype=torch.int64, device=out.device)
        got = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(got, n_local)
        sizes = [int(s.item()) for s in got]
    sizes = list(sizes)
    if len(sizes) != world or sizes[rank] != out.shape[0]:
        raise ValueError("gather_packets: sizes do not describe this job")
    n_max = max(sizes)
    width = out.shape[1] if out.dim() == 2 else None
    if trim is not False and trim is not None and width is not None:
        if trim is True:
            w = trim_width(lens, width)
        else:
            w = min(width, max(16, (int(trim) + 15) & ~15))   # the caller's bound: no collective, no host read
        out = out[:, :w]
        width = w

    def pad(t):
        if t.shape[0] == n_max:
            return t.contiguous()
        p = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[:t.shape[0]] = t
        return p

    works, bufs_all = [], []
    for t in (out, lens, rng):
        t = pad(t)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        w = dist.gather(t, bufs, dst=dst, async_op=async_op)
        if async_op:
            works.append(w)
        bufs_all.append((t, bufs))      # keep the send buffers alive until the exchange has completed

    def finish():
        if rank != dst:
            return None
        # with trim the gathered slab keeps the trimmed row width (>= every packet length): re-expanding 8 x 65 536 rows
        # to the 1500-byte stride on the destination would cost more than the exchange itself
        return tuple(torch.cat([b[:sizes[r]] for r, b in enumerate(bufs)], dim=0) for _t, bufs in bufs_all)

    if async_op:
        pg = PendingGather(works, finish)
        pg.width = width
        return pg
    return finish()


def trim_width(lens, stride):
    """Row width (a multiple of 16, at most `stride`) that holds the longest packet of `lens` on ANY rank: one
    all_reduce(MAX) + one device->host read. Call it once -- on the first (warm-up) step of a job or per change of
    input -- and pass the result as gather_packets(trim=<int>) afterwards."""
    import torch
    import torch.distributed as dist
    w = (lens.max() if lens.numel() else torch.zeros((), dtype=lens.dtype, device=lens.device)).to(torch.int64).reshape(1)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return min(int(stride), max(16, (int(w.item()) + 15) & ~15))


def truncated(lens, width):
    """True if any packet of a gathered (or local) `lens` is longer than the row width it travelled with."""
    return bool(lens.numel()) and int(lens.max().item()) > int(width)


def mixed_counts(n_units, silk_eighths=1):
    """BASELINE configs[4] / SURVEY 8d config #5: a shard of n_units mixed units holds 7/8 CELT frames and 1/8 SILK
    records (silk_eighths / 8 in general). Returns (n_celt, n_silk) with n_celt + n_silk == n_units."""
    if n_units < 0 or not 0 <= silk_eighths <= 8:
        raise ValueError("mixed_counts")
    n_celt = n_units * (8 - silk_eighths) // 8
    return n_celt, n_units - n_celt
