"""The knife-edge rows on the CPU: the border test the similarity kernels run on the device (alicevision_amd/csrc/avdm_knife.h) is compiled for the
HOST, text unchanged, and held voxel by voxel to the oracle's LITERAL evaluation — which tests/test_oracle_ref.py pins to the reference's own
kernels.  (On the GPU the same equality shows as `similarity_volume_levels.validity_differs == 0` in tests/test_gpu_parity.py.)

How the oracle's R-side decision is isolated: its literal mode tests the re-projected patch centre, its exact mode (avo_set_exact_rc_pixel(1)) the
pixel itself; everything else — T-side test, alpha, arithmetic — is common.  So for every voxel
        valid_literal == valid_exact AND knife(pixel, plane)
must hold if knife() is the reference's R-side test; on the rows where the pixel lies exactly on the margin the two modes differ by the
reference's coin flips, elsewhere they agree and knife() must say "inside"."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from alicevision_amd import abi
from alicevision_amd.synthetic import make_scene, plane_depths

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def knife(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("knife") / "libknife_host.so")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "knife_host.cpp"),
                    "-o", out], check=True)
    lib = C.CDLL(out)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.knife_sgm_mask.argtypes = [C.POINTER(abi.Camera), vp, vp, i32, vp, i32, f32, f32, f32, vp]
    lib.knife_refine_mask.argtypes = [C.POINTER(abi.Camera), vp, vp, vp, vp, i32, i32, f32, f32, f32, vp]
    lib.knife_sgm_mask.restype = lib.knife_refine_mask.restype = None
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("W,H,seed", [(102, 70, 5), (134, 102, 9)])
def test_knife_edge_border_test_equals_the_reference_arithmetic_voxel_by_voxel(knife, W, H, seed):
    from oracle import oracle
    lib = oracle.load()
    # W / 2 and H / 2 odd: the far margin of the SGM stage (level width - 1 - 6) is an even coordinate too, i.e. a stage pixel lies exactly on it
    sc = make_scene(3, W, H, seed=seed)
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=0)
    depths = np.ascontiguousarray(plane_depths(sc, 24), np.float32)
    Z = len(depths)
    o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref)

    def run(exact):
        lib.avo_set_exact_rc_pixel(1 if exact else 0)
        try:
            if not exact:
                o.run_sgm(0, [1], depths, optimize=True)  # the SGM depths the Refine stage sweeps around: from the literal run, for both modes
            else:
                keep = (o.sgm_depth_thickness.copy(), o.sgm_depth_sim.copy())
                o.run_sgm(0, [1], depths, optimize=True)
                o.sgm_depth_thickness, o.sgm_depth_sim = keep
            raw = o.best_raw[..., :Z].copy()
            o.run_refine(0, [1])
            return raw != 255, o.refine_volume[..., : 2 * ref.halfNbDepths + 1].astype(np.float32) != 0.0, o.sgm_upscaled.copy()
        finally:
            lib.avo_set_exact_rc_pixel(0)

    sgm_lit, ref_lit, up = run(False)
    sgm_ex, ref_ex, up2 = run(True)
    assert np.array_equal(up, up2)

    # ---- SGM stage: scale 2, stepXY 2, wsh 4
    ds = sgm.scale * sgm.stepXY
    Y, X = sgm_lit.shape[:2]
    vy, vx = np.mgrid[0:Y, 0:X]
    xs = np.ascontiguousarray((vx * sgm.stepXY).astype(np.float32).ravel())
    ys = np.ascontiguousarray((vy * sgm.stepXY).astype(np.float32).ravel())
    cam = o.cam(0, sgm.scale)
    lw, lh = (W + sgm.scale - 1) // sgm.scale, (H + sgm.scale - 1) // sgm.scale
    dd = float(sgm.wsh + 2)
    mask = np.empty((Y * X, Z), np.uint8)
    knife.knife_sgm_mask(C.byref(cam), _p(xs), _p(ys), Y * X, _p(depths), Z, dd, float(lw - 1), float(lh - 1), _p(mask))
    mask = mask.reshape(Y, X, Z).astype(bool)
    on_edge = (xs == dd) | (xs == lw - 1 - dd) | (ys == dd) | (ys == lh - 1 - dd)
    assert on_edge.sum() > 0.8 * 2 * (X + Y - 28)  # all four margins are hit exactly
    assert np.array_equal(sgm_lit, sgm_ex & mask), "SGM stage: %d voxels" % int((sgm_lit != (sgm_ex & mask)).sum())
    flips = (sgm_lit != sgm_ex).reshape(Y * X, Z)
    assert flips[~on_edge].sum() == 0 and 0.05 < flips[on_edge].mean() < 0.95  # the reference's coin: only on the knife-edge rows, and a real coin there

    # ---- Refine stage: scale 1, stepXY 1, wsh 3, 31 planes around the SGM depth
    Yr, Xr, Zr = ref_lit.shape
    vy, vx = np.mgrid[0:Yr, 0:Xr]
    xs = np.ascontiguousarray(vx.astype(np.float32).ravel())
    ys = np.ascontiguousarray(vy.astype(np.float32).ravel())
    cam = o.cam(0, ref.scale)
    dd = float(ref.wsh + 2)
    dpt, pix = np.ascontiguousarray(up[..., 0].ravel()), np.ascontiguousarray(up[..., 1].ravel())
    mask = np.empty((Yr * Xr, Zr), np.uint8)
    knife.knife_refine_mask(C.byref(cam), _p(xs), _p(ys), _p(dpt), _p(pix), Yr * Xr, Zr, dd, float(W - 1), float(H - 1), _p(mask))
    mask = mask.reshape(Yr, Xr, Zr).astype(bool)
    active = (up[..., 0] > 0)[..., None]
    assert np.array_equal(ref_lit, ref_ex & (mask | ~active)), "Refine stage: %d voxels" % int((ref_lit != (ref_ex & (mask | ~active))).sum())
    on_edge = ((xs == dd) | (xs == W - 1 - dd) | (ys == dd) | (ys == H - 1 - dd)).reshape(Yr, Xr)
    flips = ref_lit != ref_ex
    assert flips[~on_edge].sum() == 0
    # with the default parameters the Refine stage's knife-edge rows (pixel 5 / W - 6) lie inside the band the SGM stage leaves without a depth
    # (its own margin is 6 level-1 pixels = 12 pixels here), so they are inactive and the coin is never thrown there ...
    assert not active[on_edge].any() and flips.sum() == 0

    # ... it is with a narrower SGM margin: Refine at scale 2 next to an SGM stage of wsh 1 (margin 3 level pixels at scale 2 / step 1 = 6 full-size
    # pixels < Refine's (3 + 2) * 2 = 10) — the same comparison, now with active knife-edge rows
    sgm2, ref2 = abi.SgmParams.default(wsh=1, stepXY=1), abi.RefineParams.default(scale=2, optimizationNbIterations=0)
    o2 = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm2, ref2)
    out = {}
    for exact in (False, True):
        lib.avo_set_exact_rc_pixel(1 if exact else 0)
        try:
            if not exact:
                o2.run_sgm(0, [1], depths)
            o2.run_refine(0, [1])
            out[exact] = (o2.refine_volume[..., : 2 * ref2.halfNbDepths + 1].astype(np.float32) != 0.0, o2.sgm_upscaled.copy())
        finally:
            lib.avo_set_exact_rc_pixel(0)
    (r_lit, up), (r_ex, up_b) = out[False], out[True]
    assert np.array_equal(up, up_b)
    Yr, Xr, Zr = r_lit.shape
    vy, vx = np.mgrid[0:Yr, 0:Xr]
    xs = np.ascontiguousarray(vx.astype(np.float32).ravel())
    ys = np.ascontiguousarray(vy.astype(np.float32).ravel())
    cam = o2.cam(0, ref2.scale)
    dd = float(ref2.wsh + 2)
    lw, lh = (W + 1) // 2, (H + 1) // 2
    dpt, pix = np.ascontiguousarray(up[..., 0].ravel()), np.ascontiguousarray(up[..., 1].ravel())
    mask = np.empty((Yr * Xr, Zr), np.uint8)
    knife.knife_refine_mask(C.byref(cam), _p(xs), _p(ys), _p(dpt), _p(pix), Yr * Xr, Zr, dd, float(lw - 1), float(lh - 1), _p(mask))
    mask = mask.reshape(Yr, Xr, Zr).astype(bool)
    active = (up[..., 0] > 0)[..., None]
    assert np.array_equal(r_lit, r_ex & (mask | ~active)), "Refine stage (scale 2): %d voxels" % int((r_lit != (r_ex & (mask | ~active))).sum())
    on_edge = ((xs == dd) | (xs == lw - 1 - dd) | (ys == dd) | (ys == lh - 1 - dd)).reshape(Yr, Xr)
    flips = r_lit != r_ex
    assert flips[~on_edge].sum() == 0 and active[on_edge].any() and flips[on_edge].sum() > 0
