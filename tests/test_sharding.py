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
    """rank-local part of the world-2 test.  The compute stand-in on CPU is the oracle (tests may use it); what is under test is the
    sharding: a view is built ONLY by its owner, every other rank computes from the bytes it RECEIVED."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from alicevision_amd import abi
    from alicevision_amd.sharding import ViewExchange, cameras_of_rank, exchange_pyramid, owner_of_view
    from alicevision_amd.synthetic import make_scene, plane_depths
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        V, W, H = 3, 96, 64
        sc = make_scene(V, W, H, seed=4)
        imgs = sc.images.numpy()
        sgm, rp = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=2)
        mode = abi.FILTER_CUDA_FIXED8

        def build(img):
            return oracle.HostPyramid(img, 1, 128, mode)

        # every rank builds ONLY the views it owns, into its slots of the exchange arena; the others start as zeros
        nbytes = build(imgs[0]).desc.bytes
        ex = ViewExchange(V, nbytes, rank, world, dist)
        bufs = [ex.buffer(v) for v in range(V)]
        for v in range(V):
            if owner_of_view(v, world) == rank:
                bufs[v].copy_(torch.from_numpy(build(imgs[v]).buf.copy()))
        ex.setup()
        ok &= ex.collectives == (V + world - 1) // world  # one collective per ROW of views, not one per view
        ok &= ex.bytes_received == nbytes * sum(1 for v in range(V) if owner_of_view(v, world) != rank)

        def depth_map(buffers, rc):
            pyr = [oracle.HostPyramid.from_bytes(W, H, 1, 128, mode, b.numpy()) for b in buffers]
            o = oracle.OracleDepthMap(None, sc.K, sc.R, sc.C, sgm, rp, filter_mode=mode, pyramids=pyr)
            o.run_sgm(rc, [v for v in range(V) if v != rc], plane_depths(sc, 8))
            return o.run_refine(rc, [v for v in range(V) if v != rc])

        # a depth map of one of MY reference cameras from received neighbour pyramids == the same from locally built ones
        rc = cameras_of_rank(list(range(V)), rank, world)[0]
        got = depth_map(bufs, rc)
        want = depth_map([torch.from_numpy(build(imgs[v]).buf.copy()) for v in range(V)], rc)
        ok &= bool(np.array_equal(got, want)) and bool((want[..., 0] > 0).mean() > 0.3)

        # steady state: every rank rebuilds the pyramid of a view it owns (a changed image); one all-gather brings the others' new pyramids
        # into the STAGING row — the arena keeps the old bytes until commit()
        mine = cameras_of_rank(list(range(V)), rank, world)[0]
        bufs[mine].copy_(torch.from_numpy(build(imgs[mine] * np.float32(0.8)).buf))
        views = [cameras_of_rank(list(range(V)), r, world)[0] for r in range(world)]
        ex.publish_async(views)
        for r, v in enumerate(views):
            if r != rank:
                ok &= bool(torch.equal(bufs[v], torch.from_numpy(build(imgs[v]).buf)))  # still the set-up bytes
        ex.commit()
        for r, v in enumerate(views):
            ok &= bool(torch.equal(bufs[v], torch.from_numpy(build(imgs[v] * np.float32(0.8)).buf)))
        # the one-shot helpers
        one = torch.full((64,), rank + 7, dtype=torch.uint8)
        got = exchange_pyramid(one, src=rank, dist=dist, all_ranks=True)
        ok &= len(got) == world and all(bool((g == r + 7).all()) for r, g in enumerate(got))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_depth_map_from_received_pyramids_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_exchange_is_identity_without_a_process_group():
    import torch
    from alicevision_amd.sharding import exchange_pyramid
    b = torch.arange(10, dtype=torch.uint8)
    assert exchange_pyramid(b, 0, None)[0] is b


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (one process per rank, rendezvous on
    127.0.0.1): the launch, the process group, the round-robin ownership and the barrier / MAX-over-ranks timing protocol run here on the
    CPU with --dry-run (gloo, no GPU work) and print ONE JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "cfg1", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["steps"] == 4
    assert out["views_owned_total"] == out["views"] == 3 and out["reference_cameras_total"] == 3  # every view / camera owned exactly once
    assert len(out["rank0_cameras"]) == 4 and set(out["rank0_cameras"]) == {0, 2}  # rank 0 of 2 owns views 0 and 2
    # and the command it becomes is the driver's own
    sys.path.insert(0, root)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], port=29511)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[9].endswith("bench.py") and cmd[10:] == ["--gpus", "4", "--steps", "2"]
