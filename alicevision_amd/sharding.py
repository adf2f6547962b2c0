"""Camera sharding and pyramid exchange across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference shards cameras over devices with one OpenMP thread per device and NO communication: every device re-reads
and re-uploads the neighbour images it needs (computeOnMultiGPUs.cpp:15-69, DepthMapEstimator.cpp:224-232).  Here each
image's Lab pyramid is built once, by the rank that OWNS the view, and handed to the other ranks over xGMI; depth maps are
independent afterwards (no collective in the compute phase).  The C++ product has the same design inside one process
(host/device.hpp: PyramidExchange, hipMemcpyPeerAsync); this module is its one-process-per-GPU form, used by bench.py and
tests/test_sharding.py.
"""


def cameras_of_rank(cams, rank, world, contiguous=False):
    """Reference cameras computed by `rank`.
    round-robin (BASELINE.json north_star, default): cams[rank::world];
    contiguous=True reproduces computeOnMultiGPUs.cpp:49-63: [rank*n/world, (rank+1)*n/world)."""
    cams = list(cams)
    if world <= 1:
        return cams
    if contiguous:
        n = len(cams)
        return cams[(rank * n) // world:((rank + 1) * n) // world]
    return cams[rank::world]


def owner_of_view(view_index, world):
    """rank that decodes / uploads / builds the pyramid of a view"""
    return view_index % world


class ViewExchange:
    """Pyramid storage of one rank: `bufs[v]` is the byte tensor of view v's pyramid on this rank.  A view is BUILT only by its owner
    (`owner_of_view`); everyone else RECEIVES it:

      setup()            every view's pyramid is broadcast once from its owner (neighbour views broadcast once, BASELINE north_star);
      publish_round(vs)  steady state of a streaming job: every rank q has just (re)built the pyramid of view vs[q] (its own); one
                         all-gather moves each of them into `bufs[vs[q]]` of every other rank, so that later depth maps on any rank
                         use the RECEIVED bytes as their neighbour pyramids.
    `dist` is torch.distributed (nccl = RCCL on the GPUs, gloo in the CPU tests) or None for a single rank."""

    def __init__(self, bufs, rank, world, dist):
        self.bufs, self.rank, self.world, self.dist = bufs, rank, world, dist
        self.bytes_received = 0

    def owns(self, v):
        return owner_of_view(v, self.world) == self.rank

    def setup(self):
        if self.dist is None or self.world == 1:
            return
        for v, b in enumerate(self.bufs):
            self.dist.broadcast(b, src=owner_of_view(v, self.world))
            if not self.owns(v):
                self.bytes_received += b.numel() * b.element_size()

    def publish_round(self, views):
        """views[q] = the view rank q rebuilt in this round: every rank contributes one (an all-gather has no empty slots; a rank without new
        work republishes any view it owns).  All pyramids have the same size."""
        if self.dist is None or self.world == 1:
            return
        assert len(views) == self.world and all(v is not None for v in views) and self.owns(views[self.rank])
        mine = self.bufs[views[self.rank]]
        # gather straight into the destination pyramids; my own slot is a scratch (its source is the input tensor)
        if not hasattr(self, "_scratch") or self._scratch.shape != mine.shape:
            import torch
            self._scratch = torch.empty_like(mine)
        out = [self._scratch if q == self.rank else self.bufs[views[q]] for q in range(self.world)]
        self.dist.all_gather(out, mine)
        self.bytes_received += (self.world - 1) * mine.numel() * mine.element_size()


def exchange_pyramid(buf, src, dist, all_ranks=False):
    """One-shot helpers kept for callers that hold a single buffer: broadcast of `buf` from `src`, or (all_ranks) an all-gather returning
    the list of every rank's buffer."""
    if dist is None or dist.get_world_size() == 1:
        return [buf]
    if not all_ranks:
        dist.broadcast(buf, src=src)
        return [buf]
    import torch
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(out, buf)
    return out
