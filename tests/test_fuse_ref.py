"""The depth-map filtering restatement pinned to the REFERENCE'S OWN functions (CPU): oracle/avdm_fuse_oracle.c against
oracle/_ref/libavdm_host_ref.so, which is fuseCut::Fuser::updateInSurr / filterGroupsRC / filterDepthMapsRC, MultiViewParams'
projection and pixel-size functions, common.cpp's epipolar helpers and mvsData's geometry compiled from the reference's text
(oracle/ref/Makefile).  Same arrays into both, results compared with ==.

Needs the library (built in this container by __graft_entry__.build() / `make -C oracle/ref`; it travels to the GPU box with the snapshot)."""
import numpy as np
import pytest

from fuse_scene import make_fuse_scene
from oracle import fuse_oracle as fo
from oracle import fuse_ref as fr

pytestmark = pytest.mark.skipif(not fr.available(), reason="oracle/_ref/libavdm_host_ref.so not built (no reference tree)")


def reference_cameras(fs):
    """the scene's cameras with iCamArr / CArr derived from P the way the reference derives them (fuse_ref.camera_from_projection)"""
    out = []
    for i in range(fs.n):
        P, _, _ = fo.camera_arrays(fs.K, fs.R[i], fs.C[i])
        out.append(fr.camera_from_projection(P, fs.width, fs.height))
    return out


@pytest.fixture(scope="module")
def scenes(oracle_lib):
    return {
        "exact": make_fuse_scene(n_views=5, width=160, height=120, seed=7),
        "defects": make_fuse_scene(n_views=6, width=144, height=112, seed=3, noise=2e-3, outliers=0.05, masked=0.03, weak=0.2),
    }


def test_pixel_size_equals_reference(scenes):
    """getCamPixelSizePlaneSweepAlpha (MultiViewParams.cpp:437-448) with everything below it — getCamPixelSizeRcTc,
    getTarEpipolarDirectedLine, get2dLineImageIntersection, triangulateMatch, lineLineIntersect, decomposeProjectionMatrix,
    pointLineDistance3D — bit for bit, over points in front of, beside and behind the cameras and the degenerate pair"""
    fs = scenes["exact"]
    cams = reference_cameras(fs)
    rng = np.random.RandomState(11)
    n_same = n_nan = 0
    for k in range(4000):
        i, j = rng.randint(fs.n), rng.randint(fs.n)
        p = np.array([rng.uniform(-3.0, 3.0), rng.uniform(-3.0, 3.0), rng.uniform(-1.0, 9.0)])
        a = fo.pixel_size_plane_sweep_alpha(p, cams[i], cams[j])
        b = fr.pixel_size_plane_sweep_alpha(p, cams[i], cams[j])
        if np.isnan(a) or np.isnan(b):
            assert np.isnan(a) and np.isnan(b), (k, i, j, p, a, b)
            n_nan += 1
        else:
            assert a == b, (k, i, j, p, a, b)
            n_same += 1
    # the rest: i == j pairs and points whose epipolar line misses the T image — the NaN path of the reference, reproduced as NaN
    assert n_same > 2000 and n_nan > 500, (n_same, n_nan)


@pytest.mark.parametrize("name", ["exact", "defects"])
@pytest.mark.parametrize("balls", [(0, 0), (1, 2)])
def test_filter_groups_equals_reference(scenes, name, balls):
    """Fuser::filterGroupsRC (Fuser.cpp:144-231) incl. updateInSurr (:66-121): the modal-count map of every camera against its neighbours,
    with one neighbour lacking a depth map, identical byte for byte — the never-reset hit counters (StaticVector::resize_with) included"""
    fs = scenes[name]
    cams = reference_cameras(fs)
    for rc in range(fs.n):
        tcs = [t for t in range(fs.n) if t != rc]
        maps = [fs.depth[t] for t in tcs]
        if rc % 2 == 1:
            maps[1] = None  # a T camera without a depth map: skipped by both (Fuser.cpp:189)
        args = (fs.depth[rc], fs.sim[rc], cams[rc], [cams[t] for t in tcs], maps)
        kw = dict(pix_tolerance_factor=2.0, pix_size_ball=balls[0], pix_size_ball_wsp=balls[1])
        a = fo.filter_groups_rc(*args, **kw)
        b = fr.filter_groups_rc(*args, **kw)
        assert np.array_equal(a, b), (rc, int((a != b).sum()))
        assert a.max() >= 2


def test_filter_depth_maps_equals_reference(scenes):
    """Fuser::filterDepthMapsRC (Fuser.cpp:250-304): every combination of modal count, support class and mask, and the maps of a scene"""
    d, s, m = [], [], []
    for nm in range(0, 7):
        for sim in (-0.9, -0.2, 0.999, 1.0, 1.3, 2.5):
            for dep in (-2.0, -2.5, -1.0, 0.0, 3.7):
                d.append(dep), s.append(sim), m.append(nm)
    d, s, m = np.array(d, np.float32), np.array(s, np.float32), np.array(m, np.uint8)
    for mn, mw in ((3, 4), (2, 2), (1, 7), (4, 3)):
        a = fo.filter_depth_maps_rc(d, s, m, mn, mw)
        b = fr.filter_depth_maps_rc(d, s, m, mn, mw)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (mn, mw)
    fs = scenes["defects"]
    cams = reference_cameras(fs)
    nmod = fo.filter_groups_rc(fs.depth[0], fs.sim[0], cams[0], cams[1:], fs.depth[1:])
    a = fo.filter_depth_maps_rc(fs.depth[0], fs.sim[0], nmod)
    b = fr.filter_depth_maps_rc(fs.depth[0], fs.sim[0], nmod)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert (a[0] == -1.0).mean() > 0.01 and (a[0] > 0).mean() > 0.5
