"""Multi-GPU plumbing on CPU: camera sharding and the pyramid exchange over torch.distributed (gloo, world_size 2)."""
import os
import sys

import numpy as np
import pytest

from alicevision_amd.sharding import cameras_of_rank, owner_of_view

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(20, 8), (100, 8), (7, 2), (3, 4), (1, 8)])
def test_shards_partition_the_camera_list(n, world):
    cams = list(range(100, 100 + n))
    for contiguous in (False, True):
        parts = [cameras_of_rank(cams, r, world, contiguous) for r in range(world)]
        assert sorted(sum(parts, [])) == cams
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # contiguous mode is computeOnMultiGPUs.cpp:49-63: [r*n/w, (r+1)*n/w)
    for r in range(world):
        assert cameras_of_rank(cams, r, world, True) == cams[(r * n) // world:((r + 1) * n) // world]
    assert {owner_of_view(v, world) for v in range(max(n, world))} <= set(range(world))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from alicevision_amd.sharding import exchange_pyramid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # set-up phase: every view's pyramid is broadcast from its owner
        for v in range(5):
            want = torch.arange(1000, dtype=torch.uint8) * (v + 1)
            buf = want.clone() if v % world == rank else torch.zeros(1000, dtype=torch.uint8)
            exchange_pyramid(buf, src=v % world, dist=dist)
            ok &= bool(torch.equal(buf, want))
        # steady state: every rank has rebuilt one pyramid; all-gather
        mine = torch.full((64,), rank + 7, dtype=torch.uint8)
        got = exchange_pyramid(mine, src=rank, dist=dist, all_ranks=True)
        ok &= len(got) == world and all(bool((g == r + 7).all()) for r, g in enumerate(got))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_pyramid_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_exchange_is_identity_without_a_process_group():
    import torch
    from alicevision_amd.sharding import exchange_pyramid
    b = torch.arange(10, dtype=torch.uint8)
    assert exchange_pyramid(b, 0, None)[0] is b
