"""CPU tests of the depth-map filtering restatement (oracle/avdm_fuse_oracle.c <- fuseCut/Fuser.cpp:66-304).  The reference has no
tests or golden vectors for this step: here the restatement is checked against the geometry it encodes; tests/test_fuse_ref.py pins it,
bit for bit, to the reference's own functions compiled for the CPU."""
import numpy as np
import pytest

from fuse_scene import camera_structs, make_fuse_scene
from oracle import fuse_oracle as fo


@pytest.fixture(scope="module")
def exact_scene(oracle_lib):
    return make_fuse_scene(n_views=5, width=160, height=120, seed=7)


def _interior(fs, margin=12):
    m = np.zeros((fs.height, fs.width), bool)
    m[margin:-margin, margin:-margin] = True
    return m


def test_pixel_size_is_the_footprint_of_a_pixel(exact_scene):
    fs = exact_scene
    cams = camera_structs(fs, fo.fuse_cam)
    # a point on the optical axis of camera 0 at distance z: mean of the lateral footprint of a pixel (z / f) and of the depth step
    # that moves the projection in the T camera by one pixel along the epipolar line (~ z^2 / (f b) for a baseline b)
    f = fs.K[0, 0]
    b = np.linalg.norm(fs.C[0] - fs.C[1])
    for z in (2.0, 4.0, 7.5):
        p = fs.C[0] + fs.R[0][2] * z
        got = fo.pixel_size_plane_sweep_alpha(p, cams[0], cams[1])
        assert got == pytest.approx(0.5 * (z / f + z * z / (f * b)), rel=0.25)
    # a camera pair without baseline: the epipolar construction degenerates (NaN path), nothing is ever consistent
    assert not (fo.pixel_size_plane_sweep_alpha(fs.C[0] + fs.R[0][2] * 4.0, cams[0], cams[0]) > 0)


def test_exact_maps_are_consistent_in_every_t_camera(exact_scene):
    fs = exact_scene
    cams = camera_structs(fs, fo.fuse_cam)
    nmod = fo.filter_groups_rc(fs.exact[0], fs.sim[0], cams[0], cams[1:], fs.exact[1:])
    inner = _interior(fs)
    # every pixel of the interior is hit by (almost) every T camera: the rounding of the projection leaves isolated gaps
    assert (nmod[inner] == fs.n - 1).mean() > 0.97
    assert nmod.max() == fs.n - 1
    # border of 2 pixels is never touched (isPixelInImage with g_border = 2)
    assert nmod[:2].max() == 0 and nmod[-2:].max() == 0 and nmod[:, :2].max() == 0 and nmod[:, -2:].max() == 0


def test_wrong_depths_are_not_consistent(exact_scene):
    fs = exact_scene
    cams = camera_structs(fs, fo.fuse_cam)
    d = fs.exact[0].copy()
    d[40:60, 50:90] *= 1.2
    nmod = fo.filter_groups_rc(d, fs.sim[0], cams[0], cams[1:], fs.exact[1:])
    assert nmod[42:58, 52:88].max() == 0
    assert (nmod[70:100, 50:90] == fs.n - 1).mean() > 0.97


def test_hit_counters_carry_over_to_later_t_cameras(exact_scene):
    """StaticVector::resize_with keeps the counters (StaticVector.hpp:70): a T camera without any hit still counts once an
    earlier T camera had one."""
    fs = exact_scene
    cams = camera_structs(fs, fo.fuse_cam)
    empty = np.full_like(fs.exact[1], -1.0)
    inner = _interior(fs)
    # last T camera has no valid depth at all: the pixel keeps the hits of the previous ones -> still counted
    last_empty = fo.filter_groups_rc(fs.exact[0], fs.sim[0], cams[0], cams[1:], fs.exact[1:-1] + [empty])
    assert (last_empty[inner] == fs.n - 1).mean() > 0.97
    # first T camera empty: nothing to carry yet -> one camera less
    first_empty = fo.filter_groups_rc(fs.exact[0], fs.sim[0], cams[0], cams[1:], [empty] + fs.exact[2:])
    assert (first_empty[inner] == fs.n - 2).mean() > 0.97
    # a T camera without a depth map is skipped altogether (Fuser.cpp:189)
    missing = fo.filter_groups_rc(fs.exact[0], fs.sim[0], cams[0], cams[1:], fs.exact[1:-1] + [None])
    assert (missing[inner] == fs.n - 2).mean() > 0.97


def test_ball_sizes_widen_the_support(exact_scene):
    fs = make_fuse_scene(n_views=4, width=120, height=90, seed=11, noise=2e-4, weak=0.3)
    cams = camera_structs(fs, fo.fuse_cam)
    n0 = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:], 2.0, 0, 0)
    n1 = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:], 2.0, 1, 1)
    nw = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:], 2.0, 0, 2)
    assert (n1 >= n0).all() and (nw >= n0).all()
    assert n1.astype(int).sum() > n0.astype(int).sum() and nw.astype(int).sum() > n0.astype(int).sum()
    # a larger tolerance never loses a hit
    n4 = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:], 4.0, 0, 0)
    assert (n4 >= n0).all()


def test_filter_depth_maps_truth_table(oracle_lib):
    # columns: depth, sim, nmod -> depth', sim'   (minNumOfModals 3, minNumOfModalsWSP2SSP 4)
    rows = [
        (-2.0, 0.3, 0, -2.0, 0.3),    # masked: untouched
        (-2.5, 1.5, 9, -2.5, 1.5),
        (5.0, -0.5, 2, 5.0, -0.5),    # strong, consistent in >= 2 T cameras: kept
        (5.0, -0.5, 1, -1.0, 1.0),    # strong, too few: removed
        (5.0, 1.4, 3, 5.0, 1.4 - 2),  # weak, consistent in 3: promoted
        (5.0, 1.4, 2, 5.0, 1.4),      # weak, consistent in 2: kept weak
        (5.0, 1.4, 1, -1.0, 1.0),     # weak, consistent in one: removed
        (5.0, 1.0, 0, -1.0, 1.0),
        (-1.0, 1.0, 0, -1.0, 1.0),    # already invalid
        (5.0, 2.9, 5, 5.0, 2.9 - 2),  # promoted, still strong (< 1)
        (5.0, 3.5, 5, 5.0, 1.5),      # promoted to a value that is still weak: kept (nmod > 1)
        (5.0, 3.5, 1, -1.0, 1.0),
    ]
    f32 = np.float32
    depth = np.array([r[0] for r in rows], f32).reshape(1, -1)
    sim = np.array([r[1] for r in rows], f32).reshape(1, -1)
    nmod = np.array([r[2] for r in rows], np.uint8).reshape(1, -1)
    want_d = np.array([r[3] for r in rows], f32)
    # "s - 2" rows are formed in float32 like the reference does
    want_s = np.array([f32(r[1]) - f32(2.0) if abs(r[4] - (r[1] - 2)) < 1e-9 else f32(r[4]) for r in rows], f32)
    d, s = fo.filter_depth_maps_rc(depth, sim, nmod)
    assert np.array_equal(d.ravel(), want_d)
    assert np.array_equal(s.ravel(), want_s)
    # other thresholds: minNumOfModals 2 keeps a strong point seen by one T camera, WSP2SSP 3 promotes with two
    d2, s2 = fo.filter_depth_maps_rc(depth, sim, nmod, 2, 3)
    assert d2[0, 3] == 5.0 and s2[0, 5] == f32(1.4) - f32(2.0)


def test_filtering_removes_outliers_and_keeps_the_surface(oracle_lib):
    fs = make_fuse_scene(n_views=6, width=160, height=120, seed=5, noise=1e-4, outliers=0.05)
    cams = camera_structs(fs, fo.fuse_cam)
    nmod = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:])
    d, s = fo.filter_depth_maps_rc(fs.depth[0], fs.sim[0], nmod)
    inner = _interior(fs)
    good = np.abs(fs.depth[0] - fs.exact[0]) < 2e-3 * fs.exact[0]
    # at this resolution (f = 150 px, baselines ~0.3) twice the pixel size is ~10 % of the depth
    bad = (fs.depth[0] > 0) & (np.abs(fs.depth[0] - fs.exact[0]) > 0.25 * fs.exact[0])
    assert bad[inner].sum() > 100
    assert (d[inner & good] > 0).mean() > 0.95
    # (one chance hit in an early T camera counts for every later one — the carry-over — so a few outliers do survive)
    assert (d[inner & bad] > 0).mean() < 0.08
    assert np.array_equal(d[d > 0], fs.depth[0][d > 0])  # surviving depths are untouched
