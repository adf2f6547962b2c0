"""Python face of the CPU parity oracle (oracle/libavdm_oracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module (pinned to the reference's own kernels through
oracle/_ref, tests/test_oracle_ref.py; what stays unpinned is listed in the header of avdm_oracle.c).  Struct layouts are shared with the product ABI (alicevision_amd/abi.py mirrors include/avdm.h);
all buffers are numpy arrays in host memory.

`OracleDepthMap` restates the per-tile control flow of the reference:
  Sgm::sgmRc (Sgm.cpp:117-188, 203-325), Sgm::smoothThicknessMap (:190-201), Refine::refineRc (Refine.cpp:97-176, 178-272).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from alicevision_amd import abi  # struct layouts only (include/avdm.h)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libavdm_oracle.so")
P = C.POINTER
vp, i32, i64, f32, u8 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ubyte

_SIG = {
    "avo_float_to_half": (C.c_ushort, [f32]),
    "avo_half_to_float": (f32, [C.c_ushort]),
    "avo_exp_p2": (f32, [f32]),
    "avo_set_ncc_precision": (None, [i32]),
    "avo_set_exact_rc_pixel": (None, [i32]),
    "avo_build_custom_patch_pattern": (i32, [i32, P(abi.PatchSubpartParams), i32, P(abi.PatchPattern)]),
    "avo_tex2dlod": (None, [P(abi.Pyramid), f32, f32, f32, P(f32 * 4)]),
    "avo_pyramid_layout": (i32, [P(abi.Pyramid), i32, i32, i32, i32, i32]),
    "avo_image_rgba_f32_to_f16x255": (None, [vp, i32, vp, i32, i32, i32]),
    "avo_image_resize": (i32, [vp, i32, i32, i32, vp, i32, i32, i32, i32]),
    "avo_image_decode_integer": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32]),
    "avo_image_undistort": (i32, [vp, i32, vp, i32, P(abi.Intrinsic), P(C.c_float * 4)]),
    "avo_image_resize_taps": (i32, [i32, i32, vp, vp]),
    "avo_rgb2lab": (None, [vp, i32, i32, i32]),
    "avo_downscale_with_gaussian_blur": (None, [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32]),
    "avo_pyramid_build_levels": (None, [P(abi.Pyramid)]),
    "avo_pyramid_fill": (i32, [P(abi.Pyramid), vp, i32]),
    "avo_camera_fill": (None, [P(abi.Camera), P(C.c_double * 9), P(C.c_double * 9), P(C.c_double * 3), i32]),
    "avo_volume_initialize_u8": (None, [vp, i64, i32, i32, i32, i32, u8]),
    "avo_volume_initialize_f16": (None, [vp, i64, i32, i32, i32, i32, f32]),
    "avo_volume_add_f16": (None, [vp, vp, i64, i32, i32, i32, i32]),
    "avo_volume_update_uninitialized": (None, [vp, vp, i64, i32, i32, i32, i32]),
    "avo_volume_compute_similarity": (None, [vp, vp, i64, i32, vp, P(abi.Camera), P(abi.Camera), P(abi.Pyramid), P(abi.Pyramid),
                                             P(abi.SgmParams), abi.Range, abi.ROI]),
    "avo_volume_refine_similarity": (None, [vp, i64, i32, i32, vp, i32, vp, i32, P(abi.Camera), P(abi.Camera), P(abi.Pyramid), P(abi.Pyramid),
                                            P(abi.RefineParams), abi.Range, abi.ROI]),
    "avo_volume_optimize": (None, [vp, vp, i64, i32, i32, i32, P(abi.Pyramid), P(abi.SgmParams), i32, abi.ROI]),
    "avo_volume_retrieve_best_depth": (None, [vp, i32, vp, i32, vp, vp, i64, i32, i32, P(abi.Camera), P(abi.SgmParams), abi.Range, abi.ROI]),
    "avo_volume_refine_best_depth": (None, [vp, i32, vp, i32, vp, i64, i32, i32, P(abi.RefineParams), abi.ROI]),
    "avo_depth_sim_map_copy_depth_only": (None, [vp, i32, vp, i32, i32, i32, f32]),
    "avo_normal_map_upscale": (None, [vp, i32, vp, i32, f32, abi.ROI]),
    "avo_depth_sim_map_compute_normal": (None, [vp, i32, vp, i32, P(abi.Camera), i32, abi.ROI]),
    "avo_depth_thickness_smooth_thickness": (None, [vp, i32, P(abi.SgmParams), P(abi.RefineParams), abi.ROI]),
    "avo_compute_sgm_upscaled_depth_pixsize_map": (None, [vp, i32, vp, i32, P(abi.Camera), P(abi.Pyramid), P(abi.RefineParams), f32, abi.ROI]),
    "avo_depth_sim_map_optimize_gradient_descent": (None, [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, P(abi.Camera), P(abi.Pyramid),
                                                           P(abi.RefineParams), abi.ROI]),
}

_lib = None


def locked_make(directory, *args):
    """`make` under an exclusive file lock: the test suite runs in several processes (pytest-xdist) that all want the checker built"""
    import fcntl
    import hashlib
    import tempfile
    # a STABLE name per directory (Python randomises str hashes per process: hash() gave every process its own lock file and no exclusion at all)
    tag = hashlib.sha1(os.path.abspath(directory).encode()).hexdigest()[:16]
    with open(os.path.join(tempfile.gettempdir(), "avdm_make_%s.lock" % tag), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-C", directory, "-s"] + list(args), check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def build():
    locked_make(HERE)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class well_posed:
    """Context manager: the reference's formulas with the ill-conditioned part evaluated in its well-posed form — double-precision NCC sums,
    the R centre colour fetched at the exact pixel.  The R-side BORDER TEST stays the reference's (on the re-projected patch centre, its
    per-voxel coin flips on the knife-edge rows included: since round 4 the GPU kernels evaluate that test with the reference's own
    operations there).  exact_border=True is the mode of rounds 1-3 (border test on the exact pixel as well).
    See the comments in avdm_oracle.c and DESIGN.md."""

    def __init__(self, exact_border=False):
        self.mode = 1 if exact_border else 2

    def __enter__(self):
        load().avo_set_ncc_precision(1)
        load().avo_set_exact_rc_pixel(self.mode)

    def __exit__(self, *a):
        load().avo_set_ncc_precision(0)
        load().avo_set_exact_rc_pixel(0)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def camera_fill(K, R, Cc, downscale):
    cam = abi.Camera()
    Ka = (C.c_double * 9)(*[float(v) for v in np.asarray(K).reshape(-1)])
    Ra = (C.c_double * 9)(*[float(v) for v in np.asarray(R).reshape(-1)])
    Ca = (C.c_double * 3)(*[float(v) for v in np.asarray(Cc).reshape(-1)])
    load().avo_camera_fill(C.byref(cam), C.byref(Ka), C.byref(Ra), C.byref(Ca), int(downscale))
    return cam


class HostPyramid:
    """fp16 Lab mip pyramid in host memory (DeviceMipmapImage restated)."""

    def __init__(self, rgba, min_downscale, max_downscale, filter_mode):
        lib = load()
        h, w = rgba.shape[:2]
        self.desc = abi.Pyramid()
        assert lib.avo_pyramid_layout(C.byref(self.desc), w, h, min_downscale, max_downscale, filter_mode) == 0
        self.buf = np.zeros(self.desc.bytes, dtype=np.uint8)
        self.desc.base = self.buf.ctypes.data
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        assert lib.avo_pyramid_fill(C.byref(self.desc), ptr(rgba), w * 16) == 0

    @classmethod
    def from_bytes(cls, width, height, min_downscale, max_downscale, filter_mode, raw):
        """a pyramid whose bytes were built elsewhere (e.g. received from the rank that owns the view)"""
        self = cls.__new__(cls)
        self.desc = abi.Pyramid()
        assert load().avo_pyramid_layout(C.byref(self.desc), width, height, min_downscale, max_downscale, filter_mode) == 0
        self.buf = np.ascontiguousarray(raw, dtype=np.uint8)
        assert self.buf.size == self.desc.bytes
        self.desc.base = self.buf.ctypes.data
        return self

    def level(self, l):
        d = self.desc
        raw = self.buf[d.offset[l]:d.offset[l] + d.pitch[l] * d.height[l]].reshape(d.height[l], d.pitch[l])
        return raw[:, :d.width[l] * 8].view(np.float16).reshape(d.height[l], d.width[l], 4)


def ceil_div(a, b):
    return (a + b - 1) // b


class OracleDepthMap:
    """One tile of one R camera through SGM + Refine on the CPU.  roi = full-resolution (process) pixel ROI."""

    def __init__(self, images, K, Rs, Cs, sgm, refine, filter_mode=abi.FILTER_CUDA_FIXED8, roi=None, pyramids=None):
        self.lib = load()
        self.sgm, self.refine = sgm, refine
        if pyramids is not None:  # prebuilt (or received) pyramids instead of images
            self.pyr = list(pyramids)
            W, H = self.pyr[0].desc.width0, self.pyr[0].desc.height0
        else:
            n, H, W = images.shape[:3]
            min_ds = min(sgm.scale, refine.scale)
            max_ds = max(sgm.scale, refine.scale) * 64  # DepthMapEstimator.cpp:324-325
            self.pyr = [HostPyramid(images[i], min_ds, max_ds, filter_mode) for i in range(n)]
        self.W, self.H = W, H
        self.K, self.Rs, self.Cs = K, Rs, Cs
        self.roi = roi if roi is not None else (0, W, 0, H)

    def cam(self, i, scale):
        return camera_fill(self.K, self.Rs[i], self.Cs[i], scale)

    def droi(self, ds):
        # downscaleROI (mvsData/ROI.hpp:181): begin floor-divided, end ceil-divided
        x0, x1, y0, y1 = self.roi
        return abi.ROI.make(x0 // ds, ceil_div(x1, ds), y0 // ds, ceil_div(y1, ds))

    # ---- Sgm::sgmRc ----
    def run_sgm(self, rc, tcs, depths, tc_ranges=None, optimize=True, tile_buffer=None):
        """tile_buffer = (bufferWidth, bufferHeight) of the tile workflow.  The reference allocates its volumes for the tile BUFFER
        (Sgm.cpp:37-72), fills them with 255, sweeps the tile's ROI into their corner — and AGGREGATES OVER THE ALLOCATED EXTENT
        (cuda_volumeAggregatePath takes its dimensions from the volume, deviceSimilarityVolume.cu:278-283): the reverse paths, seeded
        from slice 0 (the quirk restated in avo_volume_optimize), cross the 255-filled remainder before they enter the ROI, so the
        result inside the ROI depends on the buffer size.  None: a buffer the size of this tile's ROI."""
        lib, sp = self.lib, self.sgm
        ds = sp.scale * sp.stepXY
        roi = self.droi(ds)
        X, Y, Z = roi.width, roi.height, len(depths)
        AX, AY = (X, Y) if tile_buffer is None else (ceil_div(tile_buffer[0], ds), ceil_div(tile_buffer[1], ds))
        assert AX >= X and AY >= Y, "the tile does not fit its buffer"
        Zp = ceil_div(Z, 4) * 4
        self.vol_dims = (X, Y, Z, Zp)
        best = np.empty((AY, AX, Zp), np.uint8)
        second = np.empty((AY, AX, Zp), np.uint8)
        py, pxx = AX * Zp, Zp
        lib.avo_volume_initialize_u8(ptr(best), py, pxx, AX, AY, Zp, 255)
        lib.avo_volume_initialize_u8(ptr(second), py, pxx, AX, AY, Zp, 255)
        depths = np.ascontiguousarray(depths, np.float32)
        rcCam = self.cam(rc, sp.scale)
        for ti, tc in enumerate(tcs):
            tcCam = self.cam(tc, sp.scale)
            r = tc_ranges[ti] if tc_ranges else (0, Z)
            lib.avo_volume_compute_similarity(ptr(best), ptr(second), py, pxx, ptr(depths), C.byref(rcCam), C.byref(tcCam),
                                              C.byref(self.pyr[rc].desc), C.byref(self.pyr[tc].desc), C.byref(sp), abi.Range(r[0], r[1]), roi)
        self.best_raw = best[:Y, :X].copy()
        lib.avo_volume_update_uninitialized(ptr(best), ptr(second), py, pxx, AX, AY, Z)
        self.second = second[:Y, :X]
        self.second_buffer = second  # the whole laid-out volume (tile in its corner, 255 elsewhere): what the aggregation walks
        if optimize:
            lib.avo_volume_optimize(ptr(best), ptr(second), py, pxx, AX, AY, C.byref(self.pyr[rc].desc), C.byref(sp), Z, roi)
        else:
            best[...] = second
        self.filtered = best[:Y, :X]
        self.filtered_buffer = best
        dt = np.empty((Y, X, 2), np.float32)
        dsm = np.empty((Y, X, 2), np.float32)
        rc1 = self.cam(rc, 1)
        lib.avo_volume_retrieve_best_depth(ptr(dt), X * 8, ptr(dsm), X * 8, ptr(depths), ptr(best), py, pxx, Z, C.byref(rc1), C.byref(sp),
                                           abi.Range(0, Z), roi)
        self.sgm_depth_thickness = dt
        self.sgm_depth_sim = dsm
        return dt, dsm

    # ---- Sgm::smoothThicknessMap + Refine::refineRc ----
    def run_refine(self, rc, tcs, refine_enabled=True, optimize_enabled=True, tile_buffer=None):
        """tile_buffer = (bufferWidth, bufferHeight) of the tile workflow: the reference allocates its maps for the tile BUFFER and takes the
        upscale ratio from those allocated widths (deviceDepthSimilarityMap.cu:115-117); None: a buffer the size of this tile's ROI"""
        lib, sp, rp = self.lib, self.sgm, self.refine
        dsS = sp.scale * sp.stepXY
        dsR = rp.scale * rp.stepXY
        roiS, roiR = self.droi(dsS), self.droi(dsR)
        dt = self.sgm_depth_thickness.copy()
        lib.avo_depth_thickness_smooth_thickness(ptr(dt), roiS.width * 8, C.byref(sp), C.byref(rp), roiS)
        self.sgm_depth_thickness_smooth = dt
        X, Y = roiR.width, roiR.height
        rcCam = self.cam(rc, rp.scale)
        up = np.empty((Y, X, 2), np.float32)
        if tile_buffer is None:
            ratio = np.float32(roiS.width) / np.float32(X)  # allocated widths == ROI widths
        else:
            ratio = np.float32(ceil_div(tile_buffer[0], dsS)) / np.float32(ceil_div(tile_buffer[0], dsR))
        lib.avo_compute_sgm_upscaled_depth_pixsize_map(ptr(up), X * 8, ptr(dt), roiS.width * 8, C.byref(rcCam), C.byref(self.pyr[rc].desc),
                                                       C.byref(rp), ratio, roiR)
        self.sgm_upscaled = up
        Zr = rp.halfNbDepths * 2 + 1
        Zrp = ceil_div(Zr, 8) * 8
        refined = np.empty((Y, X, 2), np.float32)
        if refine_enabled:
            vol = np.zeros((Y, X, Zrp), np.float16)
            py, pxx = X * Zrp * 2, Zrp * 2
            lib.avo_volume_initialize_f16(ptr(vol), py, pxx, X, Y, Zr, 0.0)
            for tc in tcs:
                tcCam = self.cam(tc, rp.scale)
                lib.avo_volume_refine_similarity(ptr(vol), py, pxx, Zr, ptr(up), X * 8, None, 0, C.byref(rcCam), C.byref(tcCam),
                                                 C.byref(self.pyr[rc].desc), C.byref(self.pyr[tc].desc), C.byref(rp), abi.Range(0, Zr), roiR)
            self.refine_volume = vol
            lib.avo_volume_refine_best_depth(ptr(refined), X * 8, ptr(up), X * 8, ptr(vol), py, pxx, Zr, C.byref(rp), roiR)
        else:
            lib.avo_depth_sim_map_copy_depth_only(ptr(refined), X * 8, ptr(up), X * 8, X, Y, 1.0)
        self.refined = refined
        if optimize_enabled and rp.optimizationNbIterations > 0:
            opt = np.empty((Y, X, 2), np.float32)
            var = np.empty((Y, X), np.float32)
            tmp = np.empty((Y, X), np.float32)
            lib.avo_depth_sim_map_optimize_gradient_descent(ptr(opt), X * 8, ptr(var), X * 4, ptr(tmp), X * 4, X, Y, ptr(up), X * 8,
                                                            ptr(refined), X * 8, C.byref(rcCam), C.byref(self.pyr[rc].desc), C.byref(rp), roiR)
            self.img_variance = var
        else:
            opt = refined.copy()
        self.optimized = opt
        return opt


def exr_lines_to_rgba(lines, line_stride, width, height, chan_offset, chan_type):
    """numpy restatement of avdm_image_decode_exr_lines (and of what image::readImage hands mvsUtils::loadImage for an .exr,
    mvsUtils/fileIO.cpp:386-446, through OpenImageIO): `lines` = bytes of the scan lines as OpenEXR stores them (per line the channels one after
    the other, `width` samples each; "OpenEXR File Layout", openexr.com), line y at y * line_stride; chan_offset / chan_type of R, G, B, A
    (type 0 UINT -> float(u), 1 HALF -> exact, 2 FLOAT; A offset -1 -> 1).  Test infrastructure."""
    buf = np.frombuffer(bytes(lines), np.uint8)
    out = np.ones((height, width, 4), np.float32)
    dt = {0: "<u4", 1: "<f2", 2: "<f4"}
    for k in range(4):
        if chan_offset[k] < 0:
            continue
        size = 2 if chan_type[k] == 1 else 4
        for y in range(height):
            o = y * line_stride + chan_offset[k]
            out[y, :, k] = np.frombuffer(buf[o:o + size * width].tobytes(), dt[chan_type[k]]).astype(np.float32)
    return out

