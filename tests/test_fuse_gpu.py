"""GPU parity of the depth-map filtering kernels (include/avdm_fuse.h) against the CPU restatement (oracle/avdm_fuse_oracle.c):
bit-exact modal-count maps and filtered maps — the kernels run the reference's double arithmetic in its order (-ffp-contract=off,
IEEE division / square root)."""
import numpy as np
import pytest
import torch

from fuse_scene import camera_structs, make_fuse_scene

pytestmark = pytest.mark.gpu


def _gpu_nmod(fs, rc, order, depths, tol=2.0, ball=0, ball_wsp=0, pitched=False):
    from alicevision_amd import fuse
    cams = camera_structs(fs, fuse.fuse_camera)
    dev = torch.device("cuda:0")

    def up(a):
        t = torch.from_numpy(a).to(dev)
        if pitched:  # row pitch larger than the width
            buf = torch.full((a.shape[0], a.shape[1] + 24), float("nan"), dtype=t.dtype, device=dev)
            buf[:, :a.shape[1]] = t
            return buf[:, :a.shape[1]]
        return t

    tds = [None if d is None else up(d) for d in depths]
    out = fuse.filter_groups(up(fs.depth[rc]), up(fs.sim[rc]), cams[rc], [cams[i] for i in order], tds, tol, ball, ball_wsp)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _cpu_nmod(fs, rc, order, depths, tol=2.0, ball=0, ball_wsp=0):
    from oracle import fuse_oracle as fo
    cams = camera_structs(fs, fo.fuse_cam)
    return fo.filter_groups_rc(fs.depth[rc], fs.sim[rc], cams[rc], [cams[i] for i in order], depths, tol, ball, ball_wsp)


@pytest.mark.parametrize("cfg", [
    dict(n=5, w=160, h=120, seed=7, noise=1e-4, outliers=0.05, weak=0.2, masked=0.03, ball=0, wsp=0, tol=2.0),
    dict(n=4, w=203, h=117, seed=3, noise=3e-4, outliers=0.1, weak=0.4, masked=0.0, ball=1, wsp=2, tol=2.0),
    dict(n=6, w=96, h=131, seed=12, noise=0.0, outliers=0.0, weak=0.0, masked=0.0, ball=0, wsp=0, tol=0.5),
    dict(n=3, w=320, h=240, seed=21, noise=1e-3, outliers=0.2, weak=0.5, masked=0.1, ball=2, wsp=0, tol=3.0),
])
def test_modal_count_map_is_bit_exact(oracle_lib, cfg):
    fs = make_fuse_scene(cfg["n"], cfg["w"], cfg["h"], seed=cfg["seed"], noise=cfg["noise"], outliers=cfg["outliers"], weak=cfg["weak"],
                         masked=cfg["masked"])
    for rc in (0, cfg["n"] - 1):
        order = [i for i in range(cfg["n"]) if i != rc]
        depths = [fs.depth[i] for i in order]
        want = _cpu_nmod(fs, rc, order, depths, cfg["tol"], cfg["ball"], cfg["wsp"])
        got = _gpu_nmod(fs, rc, order, depths, cfg["tol"], cfg["ball"], cfg["wsp"])
        assert want.max() > 0
        assert np.array_equal(got, want), f"{(got != want).sum()} of {want.size} pixels differ (rc {rc})"


def test_carry_over_missing_maps_and_pitched_rows(oracle_lib):
    fs = make_fuse_scene(5, 160, 120, seed=9, noise=1e-4, outliers=0.05)
    order = [3, 1, 4, 2]
    empty = np.full_like(fs.depth[1], -1.0)
    for depths in ([fs.depth[3], None, fs.depth[4], empty], [empty, fs.depth[1], None, fs.depth[2]], [None, None, None, None], []):
        o = order[:len(depths)]
        want = _cpu_nmod(fs, 0, o, depths)
        got = _gpu_nmod(fs, 0, o, depths, pitched=True)
        assert np.array_equal(got, want)


def test_different_image_sizes_per_camera(oracle_lib):
    """T cameras whose maps have another size than the reference camera's (each camera carries its own width / height)."""
    from alicevision_amd import fuse
    from oracle import fuse_oracle as fo
    a = make_fuse_scene(3, 160, 120, seed=4, noise=1e-4)
    b = make_fuse_scene(3, 240, 180, seed=4, noise=1e-4)  # same cameras, 1.5 x the resolution
    for maker, run in ((fo.fuse_cam, "cpu"), (fuse.fuse_camera, "gpu")):
        ca, cb = camera_structs(a, maker), camera_structs(b, maker)
        if run == "cpu":
            want = fo.filter_groups_rc(a.depth[0], a.sim[0], ca[0], [cb[1], ca[2]], [b.depth[1], a.depth[2]])
        else:
            dev = torch.device("cuda:0")
            got = fuse.filter_groups(torch.from_numpy(a.depth[0]).to(dev), torch.from_numpy(a.sim[0]).to(dev), ca[0], [cb[1], ca[2]],
                                     [torch.from_numpy(b.depth[1]).to(dev), torch.from_numpy(a.depth[2]).to(dev)]).cpu().numpy()
    assert want.max() == 2 and np.array_equal(got, want)


def test_filter_depth_maps_is_bit_exact(oracle_lib):
    from alicevision_amd import fuse
    from oracle import fuse_oracle as fo
    rng = np.random.RandomState(5)
    h, w = 97, 213
    depth = rng.uniform(1, 9, (h, w)).astype(np.float32)
    depth[rng.uniform(size=(h, w)) < 0.1] = -1.0
    depth[rng.uniform(size=(h, w)) < 0.1] = -2.0
    sim = rng.uniform(-1, 1, (h, w)).astype(np.float32)
    weak = rng.uniform(size=(h, w)) < 0.4
    sim[weak] += 2.0
    sim[rng.uniform(size=(h, w)) < 0.05] = 1.0
    nmod = rng.randint(0, 7, (h, w)).astype(np.uint8)
    dev = torch.device("cuda:0")
    for mn, mw in ((3, 4), (2, 3), (1, 1), (5, 2)):
        wd, ws = fo.filter_depth_maps_rc(depth, sim, nmod, mn, mw)
        gd, gs = torch.from_numpy(depth).to(dev), torch.from_numpy(sim).to(dev)
        fuse.filter_depth_maps(gd, gs, torch.from_numpy(nmod).to(dev), mn, mw)
        assert np.array_equal(gd.cpu().numpy(), wd) and np.array_equal(gs.cpu().numpy(), ws)


def test_bad_arguments_are_reported(oracle_lib):
    from alicevision_amd import abi, fuse
    fs = make_fuse_scene(2, 64, 48, seed=1)
    cams = camera_structs(fs, fuse.fuse_camera)
    dev = torch.device("cuda:0")
    d = torch.from_numpy(fs.depth[0]).to(dev)
    with pytest.raises(abi.AvdmError):
        fuse.filter_groups(d, d, cams[0], [cams[1]], [d], 2.0, -1, 0)
    with pytest.raises(ValueError):
        fuse.filter_groups(d[:, :32], d, cams[0], [cams[1]], [d])


def test_full_size_maps_property():
    """12 MP: on exact depth maps of a smooth surface every interior pixel is consistent in every T camera (size-independent
    property; the oracle would take minutes here), and the kernel time is reported."""
    from alicevision_amd import fuse
    n, w, h = 4, 4000, 3000
    fs = make_fuse_scene(n, w, h, seed=2, device="cuda:0")
    cams = camera_structs(fs, fuse.fuse_camera)
    dev = torch.device("cuda:0")
    maps = [torch.from_numpy(e).to(dev) for e in fs.exact]
    sim = torch.full((h, w), -0.5, dtype=torch.float32, device=dev)
    out = fuse.filter_groups(maps[0], sim, cams[0], cams[1:], maps[1:])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fuse.filter_groups(maps[0], sim, cams[0], cams[1:], maps[1:], out=out)
    e1.record()
    torch.cuda.synchronize()
    inner = out[40:-40, 40:-40]
    frac = float((inner == n - 1).float().mean().item())
    print(f"filter_groups 12 MP x {n - 1} T cameras: {e0.elapsed_time(e1):.2f} ms, {frac:.4f} of the interior consistent in all")
    assert frac > 0.97 and int(out.max().item()) == n - 1
    d, s = maps[0].clone(), sim.clone()
    fuse.filter_depth_maps(d, s, out)
    assert float((d[40:-40, 40:-40] > 0).float().mean().item()) > 0.97
