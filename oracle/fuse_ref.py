"""Python face of oracle/_ref/libavdm_host_ref.so — the REFERENCE's own depth-map filtering functions (fuseCut::Fuser::filterGroupsRC /
filterDepthMapsRC / updateInSurr and the MultiViewParams / common.cpp / geometry.cpp helpers they call), compiled for this CPU from the
reference's text where it lies under /root/reference (oracle/ref/Makefile, gen_extract.py, fuse_standin.hpp, fuse_driver.cpp).

TEST INFRASTRUCTURE ONLY.  Same call shapes as oracle/fuse_oracle.py, so tests/test_fuse_ref.py can hold the C restatement
(oracle/avdm_fuse_oracle.c) against it call for call.  `available()` is False where the library did not travel."""
import ctypes as C
import os

import numpy as np

from .fuse_oracle import FuseCam

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libavdm_host_ref.so")
_lib_handle = None


def available():
    return os.path.exists(LIB_PATH)


def _lib():
    global _lib_handle
    if _lib_handle is None:
        lib = C.CDLL(LIB_PATH)
        fp, ucp = C.POINTER(C.c_float), C.POINTER(C.c_ubyte)
        lib.avref_fuse_filter_groups_rc.restype = C.c_int
        lib.avref_fuse_filter_groups_rc.argtypes = [ucp, fp, fp, C.POINTER(FuseCam), C.c_int, C.POINTER(FuseCam), C.POINTER(fp), C.c_float, C.c_int, C.c_int]
        lib.avref_fuse_filter_depth_maps_rc.restype = C.c_int
        lib.avref_fuse_filter_depth_maps_rc.argtypes = [fp, fp, ucp, C.c_size_t, C.c_int, C.c_int]
        lib.avref_fuse_pixel_size_plane_sweep_alpha.restype = C.c_double
        lib.avref_fuse_pixel_size_plane_sweep_alpha.argtypes = [C.POINTER(C.c_double), C.POINTER(FuseCam), C.POINTER(FuseCam)]
        lib.avref_fuse_camera_from_projection.restype = None
        lib.avref_fuse_camera_from_projection.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(FuseCam)]
        _lib_handle = lib
    return _lib_handle


def camera_from_projection(P, width, height):
    """FuseCam with iCamArr / CArr as the reference derives them from the projection matrix (decomposeProjectionMatrix + inverses,
    MultiViewParams.cpp:293-296) — the arrays both sides of the pin receive"""
    pp = (C.c_double * 12)(*[float(v) for v in np.asarray(P, np.float64).reshape(-1)])
    cam = FuseCam()
    _lib().avref_fuse_camera_from_projection(pp, int(width), int(height), C.byref(cam))
    return cam


def filter_groups_rc(depth, sim, rc, tcs, tc_depths, pix_tolerance_factor=2.0, pix_size_ball=0, pix_size_ball_wsp=0):
    """Fuser::filterGroupsRC of the reference: the uint8 modal-count map (h, w).  tc_depths[i] may be None."""
    lib = _lib()
    h, w = rc.height, rc.width
    depth = np.ascontiguousarray(depth, np.float32)
    sim = np.ascontiguousarray(sim, np.float32)
    assert depth.shape == (h, w) and sim.shape == (h, w)
    n = len(tcs)
    cams = (FuseCam * max(n, 1))(*tcs)
    keep = [None if d is None else np.ascontiguousarray(d, np.float32) for d in tc_depths]
    fp = C.POINTER(C.c_float)
    ptrs = (fp * max(n, 1))()
    for i, d in enumerate(keep):
        if d is not None:
            assert d.shape == (tcs[i].height, tcs[i].width)
            ptrs[i] = d.ctypes.data_as(fp)
    nmod = np.zeros((h, w), np.uint8)
    st = lib.avref_fuse_filter_groups_rc(nmod.ctypes.data_as(C.POINTER(C.c_ubyte)), depth.ctypes.data_as(fp), sim.ctypes.data_as(fp), C.byref(rc), n, cams,
                                         ptrs, float(pix_tolerance_factor), int(pix_size_ball), int(pix_size_ball_wsp))
    if st != 0:
        raise RuntimeError("avref_fuse_filter_groups_rc")
    return nmod


def filter_depth_maps_rc(depth, sim, nmod, min_num_of_modals=3, min_num_of_modals_wsp2ssp=4):
    """Fuser::filterDepthMapsRC of the reference: filtered copies (depth, sim)."""
    lib = _lib()
    d = np.array(depth, np.float32, copy=True, order="C")
    s = np.array(sim, np.float32, copy=True, order="C")
    m = np.ascontiguousarray(nmod, np.uint8)
    fp = C.POINTER(C.c_float)
    st = lib.avref_fuse_filter_depth_maps_rc(d.ctypes.data_as(fp), s.ctypes.data_as(fp), m.ctypes.data_as(C.POINTER(C.c_ubyte)), d.size, int(min_num_of_modals),
                                             int(min_num_of_modals_wsp2ssp))
    if st != 0:
        raise RuntimeError("avref_fuse_filter_depth_maps_rc")
    return d, s


def pixel_size_plane_sweep_alpha(p, rc, tc):
    lib = _lib()
    pp = (C.c_double * 3)(*[float(v) for v in p])
    return lib.avref_fuse_pixel_size_plane_sweep_alpha(pp, C.byref(rc), C.byref(tc))
