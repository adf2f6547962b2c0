"""Python face of oracle/_ref/libavdm_ref.so — the REFERENCE's own kernel-launch layer compiled for the CPU (oracle/ref/).

TEST INFRASTRUCTURE ONLY (same rule as oracle.py: tests/, smoke(), bench.py's cpu_baseline leg).  The library is built from the
reference sources where they lie (/root/reference, this container only: `make -C oracle/ref`); on the GPU box the prebuilt .so that
travelled with the snapshot is used, and everything degrades to "not available" when it is absent.

`RefDepthMap` sequences the reference's wrappers for one tile exactly like `oracle.OracleDepthMap` sequences the C restatement
(Sgm::sgmRc, Sgm::smoothThicknessMap, Refine::refineRc), with the same attributes, so that tests can compare the two field by field.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from alicevision_amd import abi  # struct layouts only (include/avdm.h)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libavdm_ref.so")
REFERENCE_TREE = "/root/reference/src/aliceVision/depthMap/cuda/planeSweeping/deviceSimilarityVolume.cu"
P = C.POINTER
vp, i32, i64, f32, u8 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ubyte

_SIG = {
    "avr_set_filter_mode": (None, [i32]),
    "avr_camera_set": (i32, [i32, P(abi.Camera)]),
    "avr_image_create": (vp, [vp, i32, i32, i32, i32, i32]),
    "avr_image_destroy": (None, [vp]),
    "avr_image_tex2dlod": (None, [vp, vp, i32, vp]),
    "avr_image_read_level": (None, [vp, i32, i32, i32, vp]),
    "avr_image_level": (f32, [vp, i32]),
    "avr_image_dimensions": (None, [vp, i32, P(i32), P(i32)]),
    "avr_build_custom_patch_pattern": (i32, [i32, P(abi.PatchSubpartParams), i32, P(abi.PatchPattern)]),
    "avr_cache_clear": (None, []),
    "avr_cache_register_image": (None, [i32, vp]),
    "avr_cache_register_camera": (None, [i32, i32, i32]),
    "avr_tile_run": (i32, [i32, i32, P(i32), i32, i32, P(i32), P(abi.SgmParams), i32, P(abi.RefineParams), i32, i32, vp, i32, P(i32), i32, vp, vp, vp, vp, vp]),
    "avr_volume_initialize_u8": (None, [vp, i64, i32, i32, i32, i32, u8]),
    "avr_volume_update_uninitialized": (None, [vp, vp, i64, i32, i32, i32, i32]),
    "avr_volume_compute_similarity": (None, [vp, vp, i64, i32, i32, i32, i32, vp, i32, i32, i32, vp, vp, P(abi.SgmParams), abi.Range, abi.ROI]),
    "avr_volume_refine_similarity": (None, [vp, i64, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, vp, vp, P(abi.RefineParams), abi.Range,
                                            abi.ROI]),
    "avr_volume_optimize": (None, [vp, vp, i64, i32, i32, i32, i32, vp, P(abi.SgmParams), i32, abi.ROI]),
    "avr_volume_retrieve_best_depth": (None, [vp, i32, vp, i32, vp, i32, vp, i64, i32, i32, i32, i32, i32, P(abi.SgmParams), abi.Range, abi.ROI]),
    "avr_volume_refine_best_depth": (None, [vp, i32, vp, i32, vp, i64, i32, i32, i32, i32, P(abi.RefineParams), abi.ROI]),
    "avr_depth_sim_map_copy_depth_only": (None, [vp, i32, vp, i32, i32, i32, f32]),
    "avr_normal_map_upscale": (None, [vp, i32, i32, i32, vp, i32, i32, i32, abi.ROI]),
    "avr_depth_thickness_smooth_thickness": (None, [vp, i32, i32, i32, P(abi.SgmParams), P(abi.RefineParams), abi.ROI]),
    "avr_compute_sgm_upscaled_depth_pixsize_map": (None, [vp, i32, i32, i32, vp, i32, i32, i32, i32, vp, P(abi.RefineParams), abi.ROI]),
    "avr_depth_sim_map_compute_normal": (None, [vp, i32, vp, i32, i32, i32, i32, i32, abi.ROI]),
    "avr_depth_sim_map_optimize_gradient_descent": (None, [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, i32, vp, P(abi.RefineParams),
                                                           abi.ROI]),
    "avr_rgb2lab": (None, [vp, i32, vp]),
    "avr_cost_yk_from_lab": (None, [vp, vp, i32, f32, f32, vp]),
    "avr_sim_stat_wsim": (None, [vp, i32, i32, vp]),
    "avr_sigmoid": (None, [vp, i32, f32, f32, f32, f32, vp, vp]),
    "avr_stat3d_plane": (None, [vp, i32, i32, vp, vp]),
    "avr_project3d": (None, [vp, vp, i32, vp]),
}

_lib = None
_variants = {}
# second, equally faithful evaluations of the same sources (oracle/ref/Makefile): "fm" = the fast intrinsics with the error model the CUDA
# programming guide documents, "fma" = contraction of a * b + c into FMAs like nvcc's default, "cuda" = both
VARIANTS = ("fm", "fma", "cuda")


def variant_path(variant):
    return LIB_PATH if not variant else os.path.join(HERE, "_ref", "libavdm_ref_%s.so" % variant)


def available(variant=""):
    """the library exists (prebuilt) or can be built here (reference tree present)"""
    return os.path.exists(variant_path(variant)) or os.path.exists(REFERENCE_TREE)


def build():
    from oracle.oracle import locked_make
    locked_make(os.path.join(HERE, "ref"))
    return LIB_PATH


def load(variant=""):
    global _lib
    if variant not in _variants:
        if os.path.exists(REFERENCE_TREE):
            build()  # no-op when up to date
        path = variant_path(variant)
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref (%s) is not built and /root/reference is absent" % os.path.basename(path))
        lib = C.CDLL(path)  # RTLD_LOCAL; the variants are linked -Bsymbolic: each one keeps its own constant memory / shim state
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _variants[variant] = lib
        if not variant:
            _lib = lib
    return _variants[variant]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def ceil_div(a, b):
    return (a + b - 1) // b


class RefImage:
    """DeviceMipmapImage (the reference's) filled from a float RGBA image."""

    def __init__(self, rgba, min_downscale, max_downscale, variant=""):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w = rgba.shape[:2]
        self.width0, self.height0 = w, h
        self.min_downscale = min_downscale
        self.lib = load(variant)
        self.h = self.lib.avr_image_create(ptr(rgba), w * 16, w, h, min_downscale, max_downscale)
        self.levels = int(np.log2(max_downscale // min_downscale)) + 1

    def __del__(self):
        if getattr(self, "h", None) and getattr(self, "lib", None) is not None:
            self.lib.avr_image_destroy(self.h)
            self.h = None

    def level(self, l):
        """texels of mip level l as float32 (values are exact fp16 numbers)"""
        w, h = ceil_div(self.width0, self.min_downscale), ceil_div(self.height0, self.min_downscale)
        for _ in range(l):
            w, h = w // 2, h // 2
        out = np.empty((h, w, 4), np.float32)
        self.lib.avr_image_read_level(self.h, l, w, h, ptr(out))
        return out

    def tex2dlod(self, uvl):
        uvl = np.ascontiguousarray(uvl, np.float32)
        out = np.empty((len(uvl), 4), np.float32)
        self.lib.avr_image_tex2dlod(self.h, ptr(uvl), len(uvl), ptr(out))
        return out


class RefDepthMap:
    """One tile of one R camera through the reference's SGM + Refine wrappers on the CPU (same interface as OracleDepthMap)."""

    def __init__(self, images, K, Rs, Cs, sgm, refine, filter_mode=abi.FILTER_CUDA_FIXED8, roi=None, variant=""):
        self.lib = load(variant)
        self.lib.avr_set_filter_mode(filter_mode)
        self.filter_mode = filter_mode
        self.sgm, self.refine = sgm, refine
        n, H, W = images.shape[:3]
        self.W, self.H = W, H
        min_ds = min(sgm.scale, refine.scale)
        max_ds = max(sgm.scale, refine.scale) * 64  # DepthMapEstimator.cpp:324-325
        self.img = [RefImage(images[i], min_ds, max_ds, variant) for i in range(n)]
        self.K, self.Rs, self.Cs = K, Rs, Cs
        self.roi = roi if roi is not None else (0, W, 0, H)
        self._slots = {}

    def slot(self, i, scale):
        """constant-memory slot of camera i at `scale` (the camera block itself comes from the oracle's avo_camera_fill, which
        tests/test_oracle_ref.py checks separately)"""
        key = (i, scale)
        if key not in self._slots:
            from oracle import oracle
            s = len(self._slots)
            cam = oracle.camera_fill(self.K, self.Rs[i], self.Cs[i], scale)
            assert self.lib.avr_camera_set(s, C.byref(cam)) == 0
            self._slots[key] = s
        return self._slots[key]

    def droi(self, ds):
        x0, x1, y0, y1 = self.roi
        return abi.ROI.make(x0 // ds, ceil_div(x1, ds), y0 // ds, ceil_div(y1, ds))

    def run_sgm(self, rc, tcs, depths, tc_ranges=None, optimize=True):
        lib, sp = self.lib, self.sgm
        lib.avr_set_filter_mode(self.filter_mode)
        roi = self.droi(sp.scale * sp.stepXY)
        X, Y, Z = roi.width, roi.height, len(depths)
        Zp = ceil_div(Z, 4) * 4
        self.vol_dims = (X, Y, Z, Zp)
        best = np.full((Y, X, Zp), 255, np.uint8)
        second = np.full((Y, X, Zp), 255, np.uint8)
        py, pxx = X * Zp, Zp
        lib.avr_volume_initialize_u8(ptr(best), py, pxx, X, Y, Zp, 255)
        lib.avr_volume_initialize_u8(ptr(second), py, pxx, X, Y, Zp, 255)
        depths = np.ascontiguousarray(depths, np.float32)
        rcS = self.slot(rc, sp.scale)
        for ti, tc in enumerate(tcs):
            r = tc_ranges[ti] if tc_ranges else (0, Z)
            lib.avr_volume_compute_similarity(ptr(best), ptr(second), py, pxx, X, Y, Zp, ptr(depths), Z, rcS, self.slot(tc, sp.scale),
                                              self.img[rc].h, self.img[tc].h, C.byref(sp), abi.Range(r[0], r[1]), roi)
        self.best_raw = best.copy()
        # the reference updates the whole allocated volume (Sgm.cpp:271-273); planes >= Z stay 255 either way
        lib.avr_volume_update_uninitialized(ptr(best), ptr(second), py, pxx, X, Y, Zp)
        self.second = second
        if optimize:
            lib.avr_volume_optimize(ptr(best), ptr(second), py, pxx, X, Y, Zp, self.img[rc].h, C.byref(sp), Z, roi)
        else:
            best[...] = second
        self.filtered = best
        dt = np.empty((Y, X, 2), np.float32)
        dsm = np.empty((Y, X, 2), np.float32)
        # volDimZ = Z like OracleDepthMap (the reference passes the allocated depth: ADVICE r1, DESIGN "deliberate deviations")
        vol = np.ascontiguousarray(best[..., :Z]) if Zp != Z else best
        lib.avr_volume_retrieve_best_depth(ptr(dt), X * 8, ptr(dsm), X * 8, ptr(depths), Z, ptr(vol), X * vol.shape[2], vol.shape[2], X, Y, Z,
                                           self.slot(rc, 1), C.byref(sp), abi.Range(0, Z), roi)
        self.sgm_depth_thickness = dt
        self.sgm_depth_sim = dsm
        return dt, dsm

    def run_refine(self, rc, tcs, refine_enabled=True, optimize_enabled=True):
        lib, sp, rp = self.lib, self.sgm, self.refine
        lib.avr_set_filter_mode(self.filter_mode)
        roiS, roiR = self.droi(sp.scale * sp.stepXY), self.droi(rp.scale * rp.stepXY)
        XS, YS = roiS.width, roiS.height
        dt = self.sgm_depth_thickness.copy()
        lib.avr_depth_thickness_smooth_thickness(ptr(dt), XS * 8, XS, YS, C.byref(sp), C.byref(rp), roiS)
        self.sgm_depth_thickness_smooth = dt
        X, Y = roiR.width, roiR.height
        rcS = self.slot(rc, rp.scale)
        up = np.empty((Y, X, 2), np.float32)
        lib.avr_compute_sgm_upscaled_depth_pixsize_map(ptr(up), X * 8, X, Y, ptr(dt), XS * 8, XS, YS, rcS, self.img[rc].h, C.byref(rp), roiR)
        self.sgm_upscaled = up
        Zr = rp.halfNbDepths * 2 + 1
        refined = np.empty((Y, X, 2), np.float32)
        if refine_enabled:
            vol = np.zeros((Y, X, Zr), np.float16)
            py, pxx = X * Zr * 2, Zr * 2
            for tc in tcs:
                lib.avr_volume_refine_similarity(ptr(vol), py, pxx, X, Y, Zr, ptr(up), X * 8, None, 0, rcS, self.slot(tc, rp.scale), self.img[rc].h,
                                                 self.img[tc].h, C.byref(rp), abi.Range(0, Zr), roiR)
            self.refine_volume = vol
            lib.avr_volume_refine_best_depth(ptr(refined), X * 8, ptr(up), X * 8, ptr(vol), py, pxx, X, Y, Zr, C.byref(rp), roiR)
        else:
            lib.avr_depth_sim_map_copy_depth_only(ptr(refined), X * 8, ptr(up), X * 8, X, Y, 1.0)
        self.refined = refined
        if optimize_enabled and rp.optimizationNbIterations > 0:
            opt = np.empty((Y, X, 2), np.float32)
            var = np.zeros((Y, X), np.float32)
            tmp = np.zeros((Y, X), np.float32)
            lib.avr_depth_sim_map_optimize_gradient_descent(ptr(opt), X * 8, ptr(var), X * 4, ptr(tmp), X * 4, X, Y, ptr(up), X * 8, ptr(refined),
                                                            X * 8, rcS, self.img[rc].h, C.byref(rp), roiR)
            self.img_variance = var
        else:
            opt = refined.copy()
        self.optimized = opt
        return opt


class RefTile(RefDepthMap):
    """One tile through the reference's OWN host classes: depthMap/Sgm.cpp and depthMap/Refine.cpp compiled whole and unchanged
    (oracle/ref/tile_driver.cpp) — their constructors' buffer sizes, Sgm::sgmRc, Sgm::smoothThicknessMap, Refine::refineRc, i.e. the
    sequence of wrapper calls and every argument the reference passes — where RefDepthMap sequences the wrappers from Python.
    The pyramids and camera blocks are the ones RefDepthMap builds (DeviceMipmapImage::fill of the reference; the camera block from
    the oracle's avo_camera_fill), registered with the stand-in DeviceCache."""

    def run_tile(self, rc, tcs, depths, limits, tile_buffer=(1024, 1024), max_depths=1500, compute_normal=False, refine=True, use_refine_fuse=True,
                 use_color_optimization=True):
        lib, sp, rp = self.lib, self.sgm, self.refine
        lib.avr_set_filter_mode(self.filter_mode)
        lib.avr_cache_clear()
        for c in [rc] + list(tcs):
            lib.avr_cache_register_image(c, self.img[c].h)
            for scale in sorted({sp.scale, rp.scale, 1}):
                lib.avr_cache_register_camera(c, scale, self.slot(c, scale))
        roiS, roiR = self.droi(sp.scale * sp.stepXY), self.droi(rp.scale * rp.stepXY)
        XS, YS, XR, YR = roiS.width, roiS.height, roiR.width, roiR.height
        dt = np.zeros((YS, XS, 2), np.float32)
        dsm = np.zeros((YS, XS, 2), np.float32)
        dts = np.zeros((YS, XS, 2), np.float32)
        nrm = np.zeros((YS, XS, 3), np.float32) if compute_normal else None
        out = np.zeros((YR, XR, 2), np.float32) if refine else None
        depths = np.ascontiguousarray(depths, np.float32)
        tc = np.asarray(list(tcs), np.int32)
        lim = np.ascontiguousarray(np.asarray(limits, np.int32).reshape(-1, 2))
        roi = np.asarray(self.roi, np.int32)
        st = lib.avr_tile_run(int(tile_buffer[0]), int(tile_buffer[1]), roi.ctypes.data_as(P(i32)), int(rc), len(tc), tc.ctypes.data_as(P(i32)), C.byref(sp),
                              int(max_depths), C.byref(rp), int(bool(use_refine_fuse)), int(bool(use_color_optimization)), ptr(depths), len(depths),
                              lim.ctypes.data_as(P(i32)), int(bool(compute_normal)), ptr(dt), ptr(dsm), ptr(dts), ptr(nrm) if compute_normal else None,
                              ptr(out) if refine else None)
        if st != 0:
            raise RuntimeError("avr_tile_run: the reference threw (see stderr)")
        self.sgm_depth_thickness, self.sgm_depth_sim, self.sgm_depth_thickness_smooth, self.sgm_normal, self.optimized = dt, dsm, dts, nrm, out
        return out
