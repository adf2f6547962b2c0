"""ctypes / numpy front end of oracle/avdm_fuse_oracle.c — the CPU restatement of fuseCut::Fuser's depth-map filtering
(Fuser.cpp:66-304).  TEST INFRASTRUCTURE ONLY (tests/, smoke, cpu_baseline); pinned to the reference's own functions by
tests/test_fuse_ref.py (oracle/fuse_ref.py, see the C file's header)."""
import ctypes as C

import numpy as np

from . import oracle as _o


class FuseCam(C.Structure):
    _fields_ = [("P", C.c_double * 12), ("iP", C.c_double * 9), ("C", C.c_double * 3), ("width", C.c_int), ("height", C.c_int)]


def _lib():
    lib = _o.load()
    if not getattr(lib, "_fuse_bound", False):
        fp, ucp = C.POINTER(C.c_float), C.POINTER(C.c_ubyte)
        lib.avo_fuse_filter_groups_rc.restype = C.c_int
        lib.avo_fuse_filter_groups_rc.argtypes = [ucp, fp, fp, C.POINTER(FuseCam), C.c_int, C.POINTER(FuseCam), C.POINTER(fp), C.c_float, C.c_int, C.c_int]
        lib.avo_fuse_filter_depth_maps_rc.restype = None
        lib.avo_fuse_filter_depth_maps_rc.argtypes = [fp, fp, ucp, C.c_size_t, C.c_int, C.c_int]
        lib.avo_fuse_pixel_size_plane_sweep_alpha.restype = C.c_double
        lib.avo_fuse_pixel_size_plane_sweep_alpha.argtypes = [C.POINTER(C.c_double), C.POINTER(FuseCam), C.POINTER(FuseCam)]
        lib._fuse_bound = True
    return lib


def fuse_cam(P, iP, Cc, width, height, cls=FuseCam):
    """camArr (3x4), iCamArr (3x3), CArr as numpy float64 -> struct (row-major)"""
    c = cls()
    c.P[:] = [float(v) for v in np.asarray(P, np.float64).reshape(-1)]
    c.iP[:] = [float(v) for v in np.asarray(iP, np.float64).reshape(-1)]
    c.C[:] = [float(v) for v in np.asarray(Cc, np.float64).reshape(-1)]
    c.width, c.height = int(width), int(height)
    return c


def camera_arrays(K, R, Cc):
    """P = K [R | -R C], iP = (K R)^-1 in double (what decomposeProjectionMatrix + inverse give up to rounding; both sides of a
    parity test receive the SAME arrays, so how they were formed does not enter the comparison)"""
    K, R, Cc = np.asarray(K, np.float64), np.asarray(R, np.float64), np.asarray(Cc, np.float64)
    P = K @ np.hstack([R, (-R @ Cc).reshape(3, 1)])
    iP = np.linalg.inv(R) @ np.linalg.inv(K)
    return P, iP, Cc


def filter_groups_rc(depth, sim, rc, tcs, tc_depths, pix_tolerance_factor=2.0, pix_size_ball=0, pix_size_ball_wsp=0):
    """Fuser::filterGroupsRC: returns the uint8 modal-count map (h, w).  tc_depths[i] may be None (camera without a depth map)."""
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
    rc_ = rc
    st = lib.avo_fuse_filter_groups_rc(nmod.ctypes.data_as(C.POINTER(C.c_ubyte)), depth.ctypes.data_as(fp), sim.ctypes.data_as(fp), C.byref(rc_), n, cams,
                                       ptrs, float(pix_tolerance_factor), int(pix_size_ball), int(pix_size_ball_wsp))
    if st != 0:
        raise MemoryError("avo_fuse_filter_groups_rc")
    return nmod


def filter_depth_maps_rc(depth, sim, nmod, min_num_of_modals=3, min_num_of_modals_wsp2ssp=4):
    """Fuser::filterDepthMapsRC: returns filtered copies (depth, sim)."""
    lib = _lib()
    d = np.array(depth, np.float32, copy=True, order="C")
    s = np.array(sim, np.float32, copy=True, order="C")
    m = np.ascontiguousarray(nmod, np.uint8)
    fp = C.POINTER(C.c_float)
    lib.avo_fuse_filter_depth_maps_rc(d.ctypes.data_as(fp), s.ctypes.data_as(fp), m.ctypes.data_as(C.POINTER(C.c_ubyte)), d.size, int(min_num_of_modals),
                                      int(min_num_of_modals_wsp2ssp))
    return d, s


def pixel_size_plane_sweep_alpha(p, rc, tc):
    lib = _lib()
    pp = (C.c_double * 3)(*[float(v) for v in p])
    return lib.avo_fuse_pixel_size_plane_sweep_alpha(pp, C.byref(rc), C.byref(tc))
