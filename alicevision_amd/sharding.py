"""Camera sharding and pyramid exchange across the GPUs of one node (one process per GPU, torch.distributed over RCCL).

The reference shards cameras over devices with one OpenMP thread per device and NO communication: every device re-reads
and re-uploads the neighbour images it needs (computeOnMultiGPUs.cpp:15-69, DepthMapEstimator.cpp:224-232).  Here each
image's Lab pyramid is built once, by the rank that owns the view, and handed to the other ranks over xGMI; depth maps are
independent afterwards (no collective in the compute phase).
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


def exchange_pyramid(buf, src, dist, all_ranks=False):
    """Make the pyramid bytes held by rank `src` available on every rank.

    all_ranks=False: one-to-all broadcast of `buf` from `src` (set-up phase: each view has one owner).
    all_ranks=True : every rank contributes its own `buf` at the same time (steady state: each rank has just rebuilt the
                     pyramid of its current reference camera); implemented as an all-gather into a staging list so that a
                     rank also receives what the others rebuilt.  Returns the gathered list (index = rank)."""
    if dist is None or dist.get_world_size() == 1:
        return [buf]
    if not all_ranks:
        dist.broadcast(buf, src=src)
        return [buf]
    import torch
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(out, buf)
    return out
