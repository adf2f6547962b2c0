"""CPU tests of the parity oracle (oracle/avdm_oracle.c): known-answer checks against independent restatements written from
the algorithm description (numpy / pure Python, small cases), analytic properties, and the committed golden fixtures.

The reference holds no test or golden vector for depthMap; these checks stand beside tests/test_oracle_ref.py, which
pins the oracle bit for bit to the reference's own kernels compiled for the CPU (oracle/_ref)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from alicevision_amd import abi
from alicevision_amd.synthetic import make_scene, plane_depths

from common import make_oracle, small_case

HERE = os.path.dirname(os.path.abspath(__file__))


def test_half_conversion_matches_ieee(oracle_lib):
    # every binary16 value round-trips; fp32 -> fp16 agrees with numpy's IEEE round-to-nearest-even on a dense sample
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    back = np.array([oracle_lib.avo_half_to_float(int(h)) for h in allh[::7]], np.float32)
    assert np.array_equal(back[~np.isnan(back)], f[::7][~np.isnan(f[::7])])
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-300, 300, 20000), rng.uniform(-1e-4, 1e-4, 5000), [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, 5.96e-8, 2.98e-8]])
    x = x.astype(np.float32)
    got = np.array([oracle_lib.avo_float_to_half(float(v)) for v in x], np.uint16)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, want)


def test_exp_p2_is_accurate(oracle_lib):
    xs = np.linspace(-30.0, 30.0, 4001)
    got = np.array([oracle_lib.avo_exp_p2(float(x)) for x in xs])
    want = np.exp(xs.astype(np.float32).astype(np.float64))
    assert np.max(np.abs(got / want - 1.0)) < 3e-7  # ~2 ulp, the error bound CUDA documents for expf


def _lab_reference(rgb255):
    """independent float64 restatement of rgb2xyz / xyz2lab (color.cuh:65-70,124-141): linear RGB in 0..255 -> CIELAB * 2.55"""
    r, g, b = [rgb255[..., i] / 255.0 for i in range(3)]
    x = (0.4124564 * r + 0.3575761 * g + 0.1804375 * b) * 100.0
    y = (0.2126729 * r + 0.7151522 * g + 0.0721750 * b) * 100.0
    z = (0.0193339 * r + 0.1191920 * g + 0.9503041 * b) * 100.0

    def f(t):
        return np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16.0 / 116.0)
    fx, fy, fz = f(x / 95.047), f(y / 100.0), f(z / 108.883)
    L = 116.0 * fy - 16.0
    a = 500.0 * (fx - fy)
    bb = 200.0 * (fy - fz)
    return np.stack([L * 2.55, a * 2.55, bb * 2.55], -1)


def test_rgb2lab_against_float64_formula(oracle_lib):
    from oracle import oracle
    rng = np.random.RandomState(1)
    img = rng.uniform(0.02, 0.98, size=(8, 16, 4)).astype(np.float32)
    img[..., 3] = 1.0
    h16 = np.empty((8, 16, 4), np.uint16)
    oracle_lib.avo_image_rgba_f32_to_f16x255(oracle.ptr(h16), 16 * 8, oracle.ptr(img), 16 * 16, 16, 8)
    rgb255 = h16.view(np.float16).astype(np.float64)[..., :3]  # what the Lab kernel really sees (fp16-rounded 0..255)
    oracle_lib.avo_rgb2lab(oracle.ptr(h16), 16 * 8, 16, 8)
    got = h16.view(np.float16).astype(np.float64)
    want = _lab_reference(rgb255)
    # fp16 storage of the result: quantum 0.125 below 256, fp32 pow/cbrt inside
    assert np.max(np.abs(got[..., :3] - want)) <= 0.13
    assert np.all(got[..., 3] == 255.0)


def test_texture_unit_texel_centres_and_midpoints(oracle_lib):
    from oracle import oracle
    sc, sgm, ref, _ = small_case(width=64, height=48)
    for mode in (abi.FILTER_EXACT, abi.FILTER_CUDA_FIXED8):
        hp = oracle.HostPyramid(sc.images[0].numpy(), 1, 8, mode)
        for l in range(hp.desc.levels):
            lev = hp.level(l).astype(np.float32)
            H, W = lev.shape[:2]
            buf = (C.c_float * 4)()
            for (x, y) in [(0, 0), (W - 1, H - 1), (W // 2, H // 3), (3 % W, 5 % H)]:
                oracle_lib.avo_tex2dlod(C.byref(hp.desc), (x + 0.5) / W, (y + 0.5) / H, float(l), C.byref(buf))
                assert np.allclose(buf[:], lev[y, x], atol=1e-4)  # texel centre -> the texel
            if W > 2 and H > 2:
                x, y = W // 2, H // 2
                oracle_lib.avo_tex2dlod(C.byref(hp.desc), (x + 1.0) / W, (y + 1.0) / H, float(l), C.byref(buf))
                want = 0.25 * (lev[y, x] + lev[y, x + 1] + lev[y + 1, x] + lev[y + 1, x + 1])
                assert np.allclose(buf[:], want, atol=1e-3)
            # clamp addressing outside [0, 1]
            oracle_lib.avo_tex2dlod(C.byref(hp.desc), -0.3, 1.7, float(l), C.byref(buf))
            assert np.allclose(buf[:], lev[H - 1, 0], atol=1e-4)


def _sgm_reference(vin, Z, axes, P1, p2map_of_axis):
    """pure-Python restatement of the 4-path aggregation from its description: float32 arithmetic in the stated
    order, truncations where the reference truncates.  p2map_of_axis(axis, rev) -> array P2[y][x] for the slice being processed."""
    f32 = np.float32
    Y, X, _ = vin.shape
    out = np.full_like(vin, 9)
    k = 0
    for axis in axes:
        for rev in (False, True):
            A, B = (Y, X) if axis == "X" else (X, Y)

            def vox(a, b):
                return (a, b) if axis == "X" else (b, a)  # -> (y, x)
            P2 = p2map_of_axis(axis, rev)
            for a in range(A):
                y0, x0 = vox(a, 0)
                prev = vin[y0, x0, :Z].astype(np.uint32)
                out[y0, x0, :Z] = 255
                for ib in range(1, B):
                    b = B - 1 - ib if rev else ib
                    yy, xx = vox(a, b)
                    best = f32(prev.min())
                    cur = np.empty(Z, np.uint32)
                    for z in range(Z):
                        pc = f32(255.0)
                        if 1 <= z < Z - 1:
                            mc = min(f32(prev[z]), f32(prev[z - 1]) + P1, f32(prev[z + 1]) + P1, best + f32(P2[yy, xx]))
                            pc = f32(f32(f32(vin[yy, xx, z]) + mc) - best)
                        cur[z] = np.uint32(pc)
                        pc = min(f32(255.0), max(f32(0.0), pc))
                        out[yy, xx, z] = np.uint8(f32(f32(f32(out[yy, xx, z]) * f32(k) + pc) / f32(k + 1)))
                    prev = cur
            k += 1
    return out


@pytest.mark.parametrize("Z,axes", [(7, "YX"), (12, "X"), (9, "XY")])
def test_sgm_aggregation_against_python_restatement(oracle_lib, Z, axes):
    from oracle import oracle
    rng = np.random.RandomState(Z)
    X, Y = 9, 7
    Zp = (Z + 3) // 4 * 4
    vin = rng.randint(0, 256, size=(Y, X, Zp)).astype(np.uint8)
    sc, sgm, ref, _ = small_case(width=64, height=48, filteringAxes=axes.encode(), p2Weighting=-37.5)  # fixed P2: no image involved
    o = make_oracle(sc, sgm, ref)
    got = np.full_like(vin, 9)
    roi = abi.ROI.make(2, 2 + X, 1, 1 + Y)
    oracle_lib.avo_volume_optimize(oracle.ptr(got), oracle.ptr(vin), X * Zp, Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
    want = _sgm_reference(vin, Z, axes, np.float32(sgm.p1), lambda axis, rev: np.full((Y, X), 37.5, np.float32))
    assert np.array_equal(got[..., :Z], want[..., :Z])
    assert np.all(got[..., Z:] == 9)  # padding planes untouched


def test_sgm_adaptive_p2_is_between_its_bounds_and_uses_the_image(oracle_lib):
    """with the colour-adaptive P2 the result must differ from both constant-P2 extremes somewhere, and agree with them nowhere
    beyond what P2 in [80, 255] allows: a sanity property of the sigmoid (kernels.cuh:715-720)"""
    from oracle import oracle
    rng = np.random.RandomState(3)
    sc, sgm, ref, _ = small_case(width=96, height=64)
    o = make_oracle(sc, sgm, ref)
    X, Y, Z = 24, 16, 12
    vin = rng.randint(0, 200, size=(Y, X, Z)).astype(np.uint8)
    roi = abi.ROI.make(0, X, 0, Y)
    res = {}
    for name, p2 in (("adaptive", 100.0), ("lo", -80.0), ("hi", -255.0)):
        sp = abi.SgmParams.default(p2Weighting=p2)
        out = np.zeros_like(vin)
        oracle_lib.avo_volume_optimize(oracle.ptr(out), oracle.ptr(vin), X * Z, Z, X, Y, C.byref(o.pyr[0].desc), C.byref(sp), Z, roi)
        res[name] = out.astype(np.int32)
    assert (res["adaptive"] != res["lo"]).any() and (res["adaptive"] != res["hi"]).any()
    lo, hi = np.minimum(res["lo"], res["hi"]), np.maximum(res["lo"], res["hi"])
    # path costs are (nearly) monotone in P2 — the subtracted per-slice minimum moves too — so only a statistical bound is asserted
    assert ((res["adaptive"] >= lo - 2) & (res["adaptive"] <= hi + 2)).mean() > 0.97


def test_ncc_identity_and_decorrelation(oracle_lib):
    """R == T with identical cameras: every valid voxel must reach the best similarity level 0 (sim = -1) on the true plane...
    here the two cameras coincide, so EVERY plane reprojects to the same pixel and scores 0; against an unrelated texture the
    score must be far from 0."""
    from oracle import oracle
    sc = make_scene(2, 128, 96, seed=2)
    sc.C[1] = sc.C[0].copy() + np.array([1e-3, 0.0, 0.0])  # coincident up to 1 mm (exactly coincident centres give 0/0 patch axes)
    sc.R[1] = sc.R[0].copy()
    imgs = sc.images.numpy().copy()
    imgs[1] = imgs[0]
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default()
    o = oracle.OracleDepthMap(imgs, sc.K, sc.R, sc.C, sgm, ref)
    depths = plane_depths(sc, 8)
    o.run_sgm(0, [1], depths, optimize=False)
    best = o.best_raw[..., :8]
    valid = best < 255
    assert valid.mean() > 0.3
    assert np.percentile(best[valid], 99) <= 2
    rng = np.random.RandomState(0)
    imgs2 = imgs.copy()
    imgs2[1, ..., :3] = rng.uniform(0.05, 0.95, size=imgs2[1, ..., :3].shape)
    o2 = oracle.OracleDepthMap(imgs2, sc.K, sc.R, sc.C, sgm, ref)
    o2.run_sgm(0, [1], depths, optimize=False)
    b2 = o2.best_raw[..., :8]
    assert np.median(b2[b2 < 255]) > 60


def test_retrieve_best_depth_on_a_known_volume(oracle_lib):
    from oracle import oracle
    sc, sgm, ref, depths = small_case(width=64, height=48, n_planes=10)
    o = make_oracle(sc, sgm, ref)
    roi = o.droi(sgm.scale * sgm.stepXY)
    X, Y, Z = roi.width, roi.height, 10
    Zp = 12
    vol = np.full((Y, X, Zp), 200, np.uint8)
    zz = (np.arange(X)[None, :] + np.arange(Y)[:, None]) % Z
    for y in range(Y):
        for x in range(X):
            vol[y, x, zz[y, x]] = 17
    vol[0, 0, :] = 255       # nothing valid -> (-1, -1)
    vol[1, 1, 3] = 17
    vol[1, 1, 6] = 17        # tie: first minimum wins
    zz[1, 1] = min(3, zz[1, 1]) if zz[1, 1] in (3, 6) else min(zz[1, 1], 3)
    dt = np.empty((Y, X, 2), np.float32)
    ds = np.empty((Y, X, 2), np.float32)
    d32 = np.ascontiguousarray(depths, np.float32)
    rc1 = oracle.camera_fill(sc.K, sc.R[0], sc.C[0], 1)
    oracle_lib.avo_volume_retrieve_best_depth(oracle.ptr(dt), X * 8, oracle.ptr(ds), X * 8, oracle.ptr(d32), oracle.ptr(vol), X * Zp, Zp, Z, C.byref(rc1),
                                              C.byref(sgm), abi.Range(0, Z), roi)
    assert tuple(dt[0, 0]) == (-1.0, -1.0) and tuple(ds[0, 0]) == (-1.0, 1.0)
    # depth = distance from C to the plane along the pixel ray; at the principal point it equals the plane depth
    Kinv = np.linalg.inv(sc.K)
    ds_step = sgm.scale * sgm.stepXY
    for (y, x) in [(1, 1), (5, 7), (Y - 1, X - 1), (Y // 2, X // 2)]:
        ray = Kinv @ np.array([x * ds_step, y * ds_step, 1.0])
        want = depths[zz[y, x]] * np.linalg.norm(ray) / ray[2]
        assert abs(dt[y, x, 0] - want) < 2e-5 * want
        assert abs(ds[y, x, 1] - (17.0 / 255.0 * 2.0 - 1.0)) < 1e-6


@pytest.mark.parametrize("name,mode", [("relief_192x144_fixed8", abi.FILTER_CUDA_FIXED8), ("relief_192x144_exact", abi.FILTER_EXACT)])
def test_golden_fixture(oracle_lib, name, mode):
    """the committed outputs — produced by the REFERENCE's own kernels through oracle/_ref (tests/golden/make_golden.py) — are reproduced bit for
    bit by the oracle, and they match the analytic surface"""
    from oracle import oracle
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    sc = make_scene(3, 192, 144, seed=3)
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=10)
    depths = plane_depths(sc, 24)
    assert np.array_equal(depths, g["depths"])
    o = oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref, filter_mode=mode)
    o.run_sgm(0, [1, 2], depths)
    out = o.run_refine(0, [1, 2])
    assert np.array_equal(o.pyr[0].level(1)[..., 0], g["level1_L"])
    assert np.array_equal(o.second[..., :24], g["second"])
    assert np.array_equal(o.filtered[..., :24], g["filtered"])
    assert np.array_equal(o.sgm_depth_thickness, g["sgm_depth_thickness"])
    assert np.array_equal(o.refine_volume.view(np.uint16)[::4, ::4, :31], g["refine_volume_s4"])
    assert np.array_equal(o.refined, g["refined"])
    assert np.allclose(out, g["optimized"], rtol=0, atol=1e-6)
    # analytic check: the estimate follows the known surface (plane spacing here is ~0.03, relief amplitude 0.2)
    d, gt = g["optimized"][..., 0], g["gt_depth"]
    v = d > 0
    assert v.mean() > 0.6
    assert np.median(np.abs(d - gt)[v]) < 0.06
    assert np.corrcoef(d[v], gt[v])[0, 1] > 0.85


def test_normal_map_oracle_matches_numpy_pca_and_analytic_plane(oracle_lib):
    """avo_depth_sim_map_compute_normal: on the depth map of a slanted plane the PCA normal is the plane normal; on a relief surface it
    equals numpy's eigh-based PCA of the same 7 x 7 neighbourhoods (independent restatement of mapKernels.cuh:393-477)."""
    import ctypes as C
    from alicevision_amd import abi
    from oracle import oracle
    sc = make_scene(1, 96, 72, seed=4)
    cam = abi.camera_fill(sc.K, sc.R[0], sc.C[0], 1)
    H, W = 72, 96
    roi = abi.ROI.make(0, W, 0, H)
    Kinv = np.linalg.inv(sc.K)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    rays = np.stack([u, v, np.ones_like(u)], -1) @ Kinv.T @ sc.R[0]   # world ray directions (R^T K^-1 x), rows
    rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
    # plane n.X = d in world coordinates
    n_true = np.array([0.2, -0.1, 1.0])
    n_true /= np.linalg.norm(n_true)
    d = 4.0
    depth = ((d - sc.C[0] @ n_true) / (rays @ n_true)).astype(np.float32)
    dm = np.zeros((H, W, 2), np.float32)
    dm[..., 0] = depth
    dm[5, 7, 0] = -1.0  # a hole
    out = np.zeros((H, W, 3), np.float32)
    oracle_lib.avo_depth_sim_map_compute_normal(oracle.ptr(out), W * 12, oracle.ptr(dm), W * 8, C.byref(cam), 1, roi)
    assert np.all(out[5, 7] == -1.0)
    ok = np.ones((H, W), bool)
    ok[5, 7] = False
    cosang = np.abs(out[ok] @ n_true)
    assert cosang.min() > 1.0 - 1e-6, cosang.min()
    # oriented towards the camera: n . (C - p) > 0
    P = sc.C[0] + rays * depth[..., None]
    assert np.all(np.einsum("ijk,ijk->ij", out, sc.C[0] - P)[ok] > 0)

    # relief surface: compare with numpy PCA
    dm2 = np.zeros((H, W, 2), np.float32)
    dm2[..., 0] = (4.0 + 0.2 * np.sin(0.21 * u) * np.cos(0.17 * v)).astype(np.float32)
    oracle_lib.avo_depth_sim_map_compute_normal(oracle.ptr(out), W * 12, oracle.ptr(dm2), W * 8, C.byref(cam), 1, roi)
    rng = np.random.RandomState(0)
    P2 = (sc.C[0] + rays * dm2[..., :1].astype(np.float64)).astype(np.float32).astype(np.float64)
    for _ in range(60):
        y, x = rng.randint(0, H), rng.randint(0, W)
        pts = P2[max(y - 3, 0):y + 4, max(x - 3, 0):x + 4].reshape(-1, 3)
        w_, vecs = np.linalg.eigh(np.cov(pts.T, bias=True))
        assert abs(out[y, x] @ vecs[:, 0]) > 1.0 - 1e-4, (y, x)


# ------------------------------------------------------------------------------------------------ image ingest: --downscale resize
def _np_lanczos3(x):
    x = np.abs(np.asarray(x, np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = 3.0 * np.sin(np.pi * x) * np.sin(np.pi * x / 3.0) / (np.pi * np.pi * x * x)
    return np.where(x > 3.0, 0.0, np.where(x < 1e-4, 1.0, v))


@pytest.mark.parametrize("dst,src", [(32, 64), (33, 67), (25, 77), (48, 192), (21, 21)])
def test_resize_taps_follow_the_published_filter(dst, src):
    """imageAlgo::resizeImage -> OpenImageIO resize with its default filter (lanczos3 when shrinking): the oracle's tap tables against an
    independent double-precision numpy evaluation of the published formula (src_xf = (x + 0.5) * src / dst, radius ceil(3 / ratio) source
    pixels, weights lanczos3(ratio * (i - rad - (frac - 0.5))) normalised to 1)."""
    from oracle import oracle
    lib = oracle.load()
    taps = lib.avo_image_resize_taps(dst, src, None, None)
    ratio = dst / src
    rad = int(np.ceil(3.0 / ratio))
    assert taps == 2 * rad + 1
    w = np.zeros((dst, taps), np.float32)
    first = np.zeros(dst, np.int32)
    lib.avo_image_resize_taps(dst, src, oracle.ptr(w), oracle.ptr(first))
    for d in range(dst):
        xf = (d + 0.5) / dst * src
        xi = int(np.floor(xf))
        frac = xf - xi
        ref = _np_lanczos3(ratio * (np.arange(taps) - rad - (frac - 0.5)))
        ref = ref / ref.sum()
        assert first[d] == xi - rad
        assert np.abs(w[d] - ref).max() < 2e-6, (d, np.abs(w[d] - ref).max())
    assert np.abs(w.sum(1) - 1.0).max() < 1e-6


def test_resize_properties():
    from oracle import oracle
    lib = oracle.load()
    rng = np.random.default_rng(3)
    h, w = 61, 83
    # a constant stays constant; an affine ramp stays the same affine function of the continuous coordinate away from the (clamped) borders
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    src = np.stack([np.full((h, w), 0.3, np.float32), 0.01 * xx + 0.02 * yy, rng.random((h, w), dtype=np.float32), np.ones((h, w), np.float32)], -1)
    src = np.ascontiguousarray(src)
    dw, dh = w // 2, h // 2
    dst = np.zeros((dh, dw, 4), np.float32)
    assert lib.avo_image_resize(oracle.ptr(dst), dw * 16, dw, dh, oracle.ptr(src), w * 16, w, h, 4) == 0
    assert np.abs(dst[..., 0] - 0.3).max() < 1e-6 and np.abs(dst[..., 3] - 1.0).max() < 1e-6
    cy, cx = np.mgrid[0:dh, 0:dw].astype(np.float64)
    sx, sy = (cx + 0.5) * w / dw - 0.5, (cy + 0.5) * h / dh - 0.5  # continuous source coordinate of a destination pixel centre
    want = 0.01 * sx + 0.02 * sy
    inner = (slice(4, dh - 4), slice(4, dw - 4))
    assert np.abs(dst[..., 1] - want)[inner].max() < 3e-4  # discrete lanczos taps only preserve the mean exactly
    # low-pass: the noise channel loses variance
    assert dst[..., 2].std() < 0.6 * src[..., 2].std()
    # enlarging is not this path
    big = np.zeros((h * 2, w * 2, 4), np.float32)
    assert lib.avo_image_resize(oracle.ptr(big), w * 2 * 16, w * 2, h * 2, oracle.ptr(src), w * 16, w, h, 4) != 0


# ------------------------------------------------------------------------------------------------ image ingest: undistortion
def _np_undistort(src, W, H, fx, fy, ox, oy, model, k, fill):
    """camera::UndistortImage restated independently in numpy (vectorised, double): distorted position of every pixel, the
    truncating `contains`, the float-converted sample position, the 2 x 2 sampler with dropped out-of-range neighbours."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    ppx, ppy = ox + W * 0.5, oy + H * 0.5
    cx, cy = (xx - ppx) / fx, (yy - ppy) / fy
    if model == 1:
        coeff = 1.0 + k[0] * (cx * cx + cy * cy)
    else:
        r = np.sqrt(cx * cx + cy * cy)
        r2 = r * r
        r4 = r2 * r2
        r6 = r4 * r2
        coeff = 1.0 + k[0] * r2 + k[1] * r4 + k[2] * r6
        if model == 3:
            coeff = coeff / (1.0 + k[0] + k[1] + k[2])
    dx, dy = cx * coeff * fx + ppx, cy * coeff * fy + ppy
    inside = (np.trunc(dx) >= 0) & (np.trunc(dx) < W) & (np.trunc(dy) >= 0) & (np.trunc(dy) < H)
    xf, yf = dx.astype(np.float32).astype(np.float64), dy.astype(np.float32).astype(np.float64)
    gx, gy = np.floor(xf), np.floor(yf)
    ax, ay = xf - gx, yf - gy
    out = np.zeros((H, W, 4), np.float64)
    tw = np.zeros((H, W), np.float64)
    for i, wy in ((0, 1.0 - ay), (1, ay)):
        for j, wx in ((0, 1.0 - ax), (1, ax)):
            ic, jc = (gy + i).astype(np.int64), (gx + j).astype(np.int64)
            ok = (ic >= 0) & (ic < H) & (jc >= 0) & (jc < W)
            wgt = np.where(ok, wx * wy, 0.0)
            out += src[np.clip(ic, 0, H - 1), np.clip(jc, 0, W - 1)].astype(np.float64) * wgt[..., None]
            tw += wgt
    near = src[np.clip(gy.astype(np.int64), 0, H - 1), np.clip(gx.astype(np.int64), 0, W - 1)]
    res = np.where((tw != 1.0)[..., None], out / np.where(tw == 0, 1.0, tw)[..., None], out)
    res = np.where((tw <= 0.2)[..., None], near, res).astype(np.float32)
    return np.where(inside[..., None], res, np.asarray(fill, np.float32)[None, None, :])


@pytest.mark.parametrize("model,k", [(1, (0.08, 0.0, 0.0)), (2, (0.1, -0.05, 0.01)), (2, (-0.25, 0.08, 0.0)), (3, (0.05, 0.02, -0.01))])
def test_undistort_matches_independent_restatement(model, k):
    from oracle import oracle
    lib = oracle.load()
    rng = np.random.default_rng(model * 7 + 1)
    H, W = 75, 101
    src = rng.random((H, W, 4), dtype=np.float32)
    fx, fy, ox, oy = 90.0, 88.0, 2.25, -1.5
    cam = abi.Intrinsic(width=W, height=H, scale_x=fx, scale_y=fy, offset_x=ox, offset_y=oy, distortion_model=model, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0.25, 0.5, 0.75, 0.0)
    got = np.zeros_like(src)
    assert lib.avo_image_undistort(oracle.ptr(got), W * 16, oracle.ptr(src), W * 16, C.byref(cam), C.byref(fill)) == 0
    want = _np_undistort(src, W, H, fx, fy, ox, oy, model, k, list(fill))
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    # barrel / pincushion distortion leaves part of the frame without a source pixel: the fill colour shows up (or not) accordingly
    filled = np.all(got == np.asarray(list(fill), np.float32), axis=-1).mean()
    assert 0.0 <= filled < 0.5


def test_undistort_without_distortion_is_a_copy():
    from oracle import oracle
    lib = oracle.load()
    src = np.random.default_rng(2).random((20, 30, 4), dtype=np.float32)
    cam = abi.Intrinsic(width=30, height=20, scale_x=40.0, scale_y=40.0, offset_x=0.0, offset_y=0.0, distortion_model=0, k=(C.c_double * 3)(0, 0, 0))
    fill = (C.c_float * 4)(0, 0, 0, 0)
    got = np.zeros_like(src)
    assert lib.avo_image_undistort(oracle.ptr(got), 30 * 16, oracle.ptr(src), 30 * 16, C.byref(cam), C.byref(fill)) == 0
    assert np.array_equal(got, src)


@pytest.mark.parametrize("w,h,dw,dh", [(64, 48, 32, 24), (101, 77, 50, 38), (400, 300, 100, 75), (640, 480, 213, 160)])
def test_image_resize_agrees_with_an_independent_lanczos3(oracle_lib, w, h, dw, dh):
    """avo_image_resize restates OpenImageIO's default resize filter, which cannot be run here (parity unpinned).  An INDEPENDENT
    implementation of the same published filter — Pillow's LANCZOS: lanczos3 stretched by the shrink factor, sampled at pixel centres,
    weights normalised — must give the same interior to float rounding, integer and fractional ratios alike; the treatment of taps that
    leave the image is OpenImageIO's own (read from its source) and is what stays unchecked."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import oracle
    olib = oracle.load()
    rng = np.random.RandomState(w + dh)
    src = rng.rand(h, w, 4).astype(np.float32)
    dst = np.zeros((dh, dw, 4), np.float32)
    assert olib.avo_image_resize(oracle.ptr(dst), dw * 16, dw, dh, oracle.ptr(src), w * 16, w, h, 4) == 0
    pil = np.stack([np.asarray(Image.fromarray(src[..., c], mode="F").resize((dw, dh), Image.LANCZOS)) for c in range(4)], axis=-1)
    b = 4
    assert np.abs(dst - pil)[b:-b, b:-b].max() < 5e-5


@pytest.mark.parametrize("bits", [8, 16])
def test_image_decode_integer_oracle(bits):
    """avo_image_decode_integer — image::readImage(..., LINEAR) for an integer file as mvsUtils::loadImage receives it (fileIO.cpp:386-446):
    samples / max, OpenImageIO's sRGB decoding on the colour channels only, grey replicated, missing alpha = 1 — against an independent
    double-precision numpy evaluation of the published formula, for every sample value and every channel layout"""
    from oracle import oracle
    lib = oracle.load()
    n = 1 << bits
    dt = np.uint8 if bits == 8 else np.uint16
    vals = np.arange(n, dtype=np.int64)
    x = vals / float(n - 1)
    lin = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)
    rng = np.random.default_rng(bits)
    for ch in (1, 2, 3, 4):
        h = 3
        src = rng.integers(0, n, size=(h, n, ch)).astype(dt)
        src[0, :, 0] = vals.astype(dt)  # every value at least once
        dst = np.full((h, n, 4), -1.0, np.float32)
        assert lib.avo_image_decode_integer(oracle.ptr(dst), n * 16, oracle.ptr(src), n * ch * (bits // 8), n, h, ch, bits, 1) == 0
        col = [0, 0, 0] if ch < 3 else [0, 1, 2]
        for k in range(3):
            want = lin[src[..., col[k]].astype(np.int64)]
            assert np.abs(dst[..., k] - want).max() < 2e-7 + 2e-7 * 1.0, (ch, k)
        if ch in (2, 4):
            assert np.array_equal(dst[..., 3], (src[..., ch - 1].astype(np.float32) * np.float32(1.0 / (n - 1))))
        else:
            assert np.all(dst[..., 3] == 1.0)
        # without the colour-space conversion: plain scaling
        assert lib.avo_image_decode_integer(oracle.ptr(dst), n * 16, oracle.ptr(src), n * ch * (bits // 8), n, h, ch, bits, 0) == 0
        assert np.array_equal(dst[..., 0], src[..., 0].astype(np.float32) * np.float32(1.0 / (n - 1)))
    assert lib.avo_image_decode_integer(oracle.ptr(dst), n * 16, oracle.ptr(src), n, n, 1, 5, bits, 1) != 0


# ------------------------------------------------------------------------------------------------ JPEG (image ingest)
JPEG_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")


def _jpeg_oracle_decode(path):
    import ctypes as C
    from alicevision_amd import abi, jpeg_io
    from oracle import oracle
    j = jpeg_io.read_coefficients(path)
    comps = j.descriptors([c["coef"].ctypes.data for c in j.components])
    out = np.zeros((j.height, j.width, 3), np.uint8)
    olib = oracle.load()
    olib.avo_image_decode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(abi.JpegComponent), C.c_int, C.c_int, C.c_int, C.c_int]
    rc = olib.avo_image_decode_jpeg(out.ctypes.data, 3 * j.width, j.width, j.height, comps, len(j.components), j.hmax, j.vmax, 0 if j.stored_as_rgb else 1)
    assert rc == 0
    return out, j


def test_jpeg_entropy_decoder_and_oracle_equal_libjpeg_turbo_on_the_golden_files():
    """host/jpeg.cpp (markers, Huffman, sequential and progressive scans, restart markers) + the oracle's restatement of libjpeg's inverse
    DCT, fancy up-sampling and colour conversion == the pixels libjpeg-turbo decoded from the same files (tests/golden/jpeg: 4:4:4 /
    4:2:2 / 4:2:0, baseline / progressive, restarts, optimised tables, grey, RGB-stored, sizes down to 1 x 1, quality 1 .. 100)."""
    __import__("common").build_host()
    exp = np.load(os.path.join(JPEG_GOLDEN, "expected.npz"))
    assert len(exp.files) >= 17
    seen_progressive = seen_rgb = 0
    for name in exp.files:
        got, j = _jpeg_oracle_decode(os.path.join(JPEG_GOLDEN, name + ".jpg"))
        assert np.array_equal(got, exp[name]), name
        seen_progressive += j.progressive
        seen_rgb += j.stored_as_rgb
    assert seen_progressive >= 3 and seen_rgb == 1


def test_jpeg_decoder_against_pillow_when_available(tmp_path):
    """the same comparison on files made now (when this host has Pillow): a larger image per sampling mode, and malformed input"""
    Image = pytest.importorskip("PIL.Image")
    __import__("common").build_host()
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:301, 0:413]
    a = np.clip(np.stack([128 + 100 * np.sin(x / 17.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 13.0 + y / 9.0), 60 + x * 0.4 + 20 * np.sin(y / 2.0)], -1) +
                rng.normal(0, 6, (301, 413, 3)), 0, 255).astype(np.uint8)
    for sub, prog in ((0, False), (1, True), (2, False), (2, True)):
        p = str(tmp_path / "t.jpg")
        Image.fromarray(a).save(p, quality=88, subsampling=sub, progressive=prog)
        got, j = _jpeg_oracle_decode(p)
        assert np.array_equal(got, np.array(Image.open(p))) and j.progressive == prog
    from alicevision_amd import jpeg_io
    data = open(p, "rb").read()
    open(p, "wb").write(b"\x89PNG" + data[4:])
    with pytest.raises(RuntimeError, match="not a JPEG"):
        jpeg_io.read_coefficients(p)
    Image.new("CMYK", (16, 16), (10, 20, 30, 40)).save(p, quality=90)
    with pytest.raises(RuntimeError, match="CMYK"):
        jpeg_io.read_coefficients(p)
    sof = data.index(b"\xff\xc2")
    open(p, "wb").write(data[:sof + 1] + b"\xc9" + data[sof + 2:])  # SOF9: arithmetic coding
    with pytest.raises(RuntimeError, match="arithmetic"):
        jpeg_io.read_coefficients(p)
