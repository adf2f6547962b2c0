"""ctypes view of include/avdm.h — the C ABI of the gfx950 library (alicevision_amd/csrc/libavdm.so).

PyTorch is used by callers only for device memory and streams; every compute call goes through these entry points.
The library is loaded lazily and LOUDLY: if libavdm.so is missing there is no fallback of any kind.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVDM_LIB", os.path.join(HERE, "csrc", "libavdm.so"))  # AVDM_LIB: A/B builds of the SAME sources while tuning

AVDM_MAX_LEVELS = 8
FILTER_EXACT = 0
FILTER_CUDA_FIXED8 = 1


class Camera(C.Structure):
    _fields_ = [("P", C.c_float * 12), ("iP", C.c_float * 9), ("R", C.c_float * 9), ("iR", C.c_float * 9), ("K", C.c_float * 9),
                ("iK", C.c_float * 9), ("C", C.c_float * 3), ("XVect", C.c_float * 3), ("YVect", C.c_float * 3), ("ZVect", C.c_float * 3)]


class Range(C.Structure):
    _fields_ = [("begin", C.c_uint), ("end", C.c_uint)]


class ROI(C.Structure):
    _fields_ = [("x", Range), ("y", Range)]

    @staticmethod
    def make(x0, x1, y0, y1):
        return ROI(Range(x0, x1), Range(y0, y1))

    @property
    def width(self):
        return self.x.end - self.x.begin

    @property
    def height(self):
        return self.y.end - self.y.begin


class Pyramid(C.Structure):
    _fields_ = [("base", C.c_void_p), ("levels", C.c_int), ("filter_mode", C.c_int), ("min_downscale", C.c_int), ("width0", C.c_int),
                ("height0", C.c_int), ("width", C.c_int * AVDM_MAX_LEVELS), ("height", C.c_int * AVDM_MAX_LEVELS),
                ("pitch", C.c_int * AVDM_MAX_LEVELS), ("offset", C.c_longlong * AVDM_MAX_LEVELS), ("bytes", C.c_longlong)]


class SgmParams(C.Structure):
    """SgmParams.hpp:21-55 defaults."""
    _fields_ = [("scale", C.c_int), ("stepXY", C.c_int), ("wsh", C.c_int), ("gammaC", C.c_double), ("gammaP", C.c_double), ("p1", C.c_double),
                ("p2Weighting", C.c_double), ("maxSimilarity", C.c_double), ("depthThicknessInflate", C.c_double), ("filteringAxes", C.c_char * 8),
                ("useConsistentScale", C.c_int), ("strictRoiQuirk", C.c_int), ("useCustomPatchPattern", C.c_int),
                ("referenceArithmetic", C.c_int)]

    @staticmethod
    def default(**kw):
        p = SgmParams(scale=2, stepXY=2, wsh=4, gammaC=5.5, gammaP=8.0, p1=10.0, p2Weighting=100.0, maxSimilarity=1.0, depthThicknessInflate=0.0,
                      filteringAxes=b"YX", useConsistentScale=0, strictRoiQuirk=1, useCustomPatchPattern=0, referenceArithmetic=0)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class RefineParams(C.Structure):
    """RefineParams.hpp:19-45 defaults."""
    _fields_ = [("scale", C.c_int), ("stepXY", C.c_int), ("wsh", C.c_int), ("halfNbDepths", C.c_int), ("nbSubsamples", C.c_int),
                ("optimizationNbIterations", C.c_int), ("sigma", C.c_double), ("gammaC", C.c_double), ("gammaP", C.c_double),
                ("interpolateMiddleDepth", C.c_int), ("useConsistentScale", C.c_int), ("useCustomPatchPattern", C.c_int),
                ("referenceArithmetic", C.c_int)]

    @staticmethod
    def default(**kw):
        p = RefineParams(scale=1, stepXY=1, wsh=3, halfNbDepths=15, nbSubsamples=10, optimizationNbIterations=100, sigma=15.0, gammaC=15.5,
                         gammaP=8.0, interpolateMiddleDepth=0, useConsistentScale=0, useCustomPatchPattern=0, referenceArithmetic=0)
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class Intrinsic(C.Structure):
    """avdm_intrinsic_t"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("scale_x", C.c_double), ("scale_y", C.c_double), ("offset_x", C.c_double),
                ("offset_y", C.c_double), ("distortion_model", C.c_int), ("k", C.c_double * 3)]


DISTORTION_NONE, DISTORTION_RADIALK1, DISTORTION_RADIALK3, DISTORTION_RADIALK3PT = 0, 1, 2, 3


class PatchSubpartParams(C.Structure):
    """CustomPatchPatternParams::SubpartParams"""
    _fields_ = [("isCircle", C.c_int), ("level", C.c_int), ("nbCoordinates", C.c_int), ("radius", C.c_float), ("weight", C.c_float)]


class PatchPatternSubpart(C.Structure):
    _fields_ = [("coordinates", (C.c_float * 2) * 24), ("nbCoordinates", C.c_int), ("level", C.c_float), ("downscale", C.c_float),
                ("weight", C.c_float), ("isCircle", C.c_int), ("wsh", C.c_int)]


class PatchPattern(C.Structure):
    _fields_ = [("subparts", PatchPatternSubpart * 4), ("nbSubparts", C.c_int)]


class JpegComponent(C.Structure):
    """avdm_jpeg_component_t"""
    _fields_ = [("coef", C.c_void_p), ("blocks_w", C.c_int), ("blocks_h", C.c_int), ("width", C.c_int), ("height", C.c_int), ("h_samp", C.c_int),
                ("v_samp", C.c_int), ("quant", C.c_uint16 * 64)]


class SgmTile(C.Structure):
    """avdm_sgm_tile_t"""
    _fields_ = [("out_vol", C.c_void_p), ("in_vol", C.c_void_p), ("pitch_y", C.c_longlong), ("pitch_x", C.c_int), ("last_depth_index", C.c_int),
                ("roi", ROI), ("rc_pyr", C.POINTER(Pyramid))]


P = C.POINTER
vp, i32, i64, f32, u8 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ubyte

# name -> (restype, argtypes); mirrors include/avdm.h one to one (tests check every symbol is exported)
SIGNATURES = {
    "avdm_last_error": (C.c_char_p, []),
    "avdm_version": (i32, []),
    "avdm_device_count": (i32, []),
    "avdm_device_info": (i32, [i32, C.c_char_p, C.c_size_t]),
    "avdm_stream_release": (i32, [vp]),
    "avdm_build_custom_patch_pattern": (i32, [i32, P(PatchSubpartParams), i32, P(PatchPattern)]),
    "avdm_pyramid_layout": (i32, [P(Pyramid), i32, i32, i32, i32, i32]),
    "avdm_image_rgba_f32_to_f16x255": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "avdm_rgb2lab": (i32, [vp, i32, i32, i32, vp]),
    "avdm_downscale_with_gaussian_blur": (i32, [vp, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "avdm_pyramid_build_levels": (i32, [P(Pyramid), vp]),
    "avdm_image_resize": (i32, [vp, i32, i32, i32, vp, i32, i32, i32, vp]),
    "avdm_image_decode_integer": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "avdm_image_decode_exr_lines": (i32, [vp, i32, vp, i64, i32, i32, vp, vp, vp]),
    "avdm_image_decode_jpeg_scratch_bytes": (C.c_size_t, [P(JpegComponent), i32]),
    "avdm_image_decode_jpeg": (i32, [vp, i32, i32, i32, P(JpegComponent), i32, i32, i32, i32, vp, vp]),
    "avdm_image_undistort": (i32, [vp, i32, vp, i32, P(Intrinsic), P(C.c_float * 4), vp]),
    "avdm_pyramid_fill": (i32, [P(Pyramid), vp, i32, vp, vp]),
    "avdm_tex2dlod": (i32, [vp, P(Pyramid), vp, i32, vp]),
    "avdm_volume_initialize_u8": (i32, [vp, i64, i32, i32, i32, i32, u8, vp]),
    "avdm_volume_initialize_f16": (i32, [vp, i64, i32, i32, i32, i32, f32, vp]),
    "avdm_volume_add_f16": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "avdm_volume_update_uninitialized": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "avdm_volume_compute_similarity": (i32, [vp, vp, i64, i32, vp, P(Camera), P(Camera), P(Pyramid), P(Pyramid), P(SgmParams), Range, ROI, vp]),
    "avdm_refine_similarity_scratch_bytes": (C.c_size_t, [C.c_size_t, i32]),
    "avdm_refine_outlier_refused": (i32, [P(C.c_uint)]),
    "avdm_volume_refine_similarity": (i32, [vp, i64, i32, i32, vp, i32, vp, i32, P(Camera), P(Camera), P(Pyramid), P(Pyramid), P(RefineParams),
                                            Range, ROI, vp]),
    "avdm_volume_optimize_scratch_bytes": (C.c_size_t, [i32, i32, i32]),
    "avdm_volume_optimize": (i32, [vp, vp, i64, i32, vp, P(Pyramid), P(SgmParams), i32, ROI, vp]),
    "avdm_volume_optimize_tiles": (i32, [i32, P(SgmTile), vp, P(SgmParams), vp]),
    "avdm_volume_optimize_prepare": (i32, [i32, P(SgmTile), vp, P(SgmParams), vp]),
    "avdm_volume_optimize_tiles_prepared": (i32, [i32, P(SgmTile), vp, P(SgmParams), vp]),
    "avdm_volume_retrieve_best_depth": (i32, [vp, i32, vp, i32, vp, vp, i64, i32, i32, P(Camera), P(SgmParams), Range, ROI, vp]),
    "avdm_volume_refine_best_depth": (i32, [vp, i32, vp, i32, vp, i64, i32, i32, P(RefineParams), ROI, vp]),
    "avdm_depth_sim_map_copy_depth_only": (i32, [vp, i32, vp, i32, i32, i32, f32, vp]),
    "avdm_normal_map_upscale": (i32, [vp, i32, vp, i32, f32, ROI, vp]),
    "avdm_depth_thickness_smooth_thickness": (i32, [vp, i32, P(SgmParams), P(RefineParams), ROI, vp]),
    "avdm_compute_sgm_upscaled_depth_pixsize_map": (i32, [vp, i32, vp, i32, P(Camera), P(Pyramid), P(RefineParams), f32, ROI, vp]),
    "avdm_depth_sim_map_compute_normal": (i32, [vp, i32, vp, i32, P(Camera), i32, ROI, vp]),
    "avdm_depth_sim_map_optimize_gradient_descent": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, i32, P(Camera), P(Pyramid),
                                                           P(RefineParams), ROI, vp]),
    "avdm_camera_fill": (None, [P(Camera), P(C.c_double * 9), P(C.c_double * 9), P(C.c_double * 3), i32]),
}



class FuseCamera(C.Structure):
    """avdm_fuse.h: camArr / iCamArr / CArr of a camera (row-major doubles) and its image size."""
    _fields_ = [("P", C.c_double * 12), ("iP", C.c_double * 9), ("C", C.c_double * 3), ("width", C.c_int), ("height", C.c_int)]


class FuseTc(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("depth_pitch", C.c_int), ("reserved", C.c_int), ("cam", FuseCamera)]


# mirrors include/avdm_fuse.h one to one
FUSE_SIGNATURES = {
    "avdm_fuse_filter_groups_scratch_bytes": (C.c_size_t, [i32, i32]),
    "avdm_fuse_filter_groups": (i32, [vp, i32, vp, i32, vp, i32, P(FuseCamera), i32, P(FuseTc), f32, i32, i32, vp, vp]),
    "avdm_fuse_filter_depth_maps": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
}

_lib = None


class AvdmError(RuntimeError):
    pass


def load(path=None):
    """Load libavdm.so and bind every declared symbol.  Raises if the library or a symbol is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise AvdmError(f"{p} not found: build it with `python -m alicevision_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in list(SIGNATURES.items()) + list(FUSE_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise AvdmError(f"{what} failed (status {rc}): {load().avdm_last_error().decode()}")


def patch_subparts(spec):
    """[("circle" | "full", radius, nbCoordinates, level, weight), ...] -> array of avdm_patch_subpart_params_t (the order of the
    reference's command-line tokens `type:radius:nbCoords:level:weight`, CustomPatchPatternParams.cpp:16-41)"""
    arr = (PatchSubpartParams * len(spec))()
    for i, (typ, radius, nb, level, weight) in enumerate(spec):
        arr[i] = PatchSubpartParams(isCircle=1 if str(typ).lower() == "circle" else 0, level=int(level), nbCoordinates=int(nb), radius=float(radius),
                                    weight=float(weight))
    return arr


def build_custom_patch_pattern(spec, group_per_level=False):
    """buildCustomPatchPattern through the library: the pattern becomes the one the similarity entry points use; returns a copy"""
    arr = patch_subparts(spec)
    out = PatchPattern()
    check(load().avdm_build_custom_patch_pattern(len(spec), arr, 1 if group_per_level else 0, C.byref(out)), "avdm_build_custom_patch_pattern")
    return out


def camera_fill(K, R, Cc, downscale):
    """fillHostCameraParameters (cuda/host/DeviceCache.cpp:41-134) through the library's host helper."""
    cam = Camera()
    Ka = (C.c_double * 9)(*[float(v) for v in K.reshape(-1)])
    Ra = (C.c_double * 9)(*[float(v) for v in R.reshape(-1)])
    Ca = (C.c_double * 3)(*[float(v) for v in Cc.reshape(-1)])
    load().avdm_camera_fill(C.byref(cam), C.byref(Ka), C.byref(Ra), C.byref(Ca), int(downscale))
    return cam
