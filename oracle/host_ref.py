"""Python face of the depth-plane-list entry point of oracle/_ref/libavdm_host_ref.so — the REFERENCE's own depthMap/SgmDepthList.cpp
compiled whole and unchanged for this CPU (oracle/ref/Makefile, host_driver.cpp), over the reference's mvsData and the stand-ins of
oracle/ref/shim_host/.  TEST INFRASTRUCTURE ONLY: tests/test_host_ref.py holds oracle/host_oracle.py against it (and
tests/test_host_cpu.py holds the C++ host against host_oracle.py).  `available()` is False where the library did not travel."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libavdm_host_ref.so")
_lib_handle = None


def available():
    return os.path.exists(LIB_PATH)


def _lib():
    global _lib_handle
    if _lib_handle is None:
        lib = C.CDLL(LIB_PATH)
        dp, ip, fp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_float)
        lib.avref_sgm_depth_list.restype = C.c_int
        lib.avref_sgm_depth_list.argtypes = [C.c_int, dp, ip, ip, C.c_int, C.c_float, C.c_float, C.c_int, dp, ip, ip, dp, C.c_int, C.c_int, ip, ip,
                                             C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, fp, C.c_int, ip, ip]
        lib.avref_nearest_cams.restype = C.c_int
        lib.avref_nearest_cams.argtypes = [C.c_int, dp, dp, C.c_int, C.c_float, C.c_float, C.c_int, ip, ip, dp, C.c_int, C.c_int, C.c_int, ip, ip, ip]
        lib.avref_image_undistort.restype = C.c_int
        lib.avref_image_undistort.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.avref_default_params.restype = C.c_int
        lib.avref_default_params.argtypes = [C.c_char_p, C.c_int]
        lib.avref_tile_roi_list.restype = C.c_int
        lib.avref_tile_roi_list.argtypes = [C.c_int] * 6 + [ip, C.c_int]
        lib.avref_tile_weight_map.restype = C.c_int
        lib.avref_tile_weight_map.argtypes = [C.c_int, C.c_int, ip, C.c_int, C.c_int, fp, fp]
        _lib_handle = lib
    return _lib_handle


def image_undistort(src, cam, fill):
    """camera::UndistortImage of the reference (cameraUndistortImage.hpp:81-139, with IntrinsicScaleOffsetDisto::getDistortedPixel, the
    radial distortion classes and image/Sampler.hpp's bilinear sampler — the reference's own code, oracle/ref/undistort_standin.hpp) on
    a float RGBA image (H, W, 4); cam: alicevision_amd.abi.Intrinsic; fill: 4 floats.  Returns the undistorted image."""
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape[:2]
    assert (w, h) == (cam.width, cam.height)
    dst = np.zeros_like(src)
    f = (C.c_float * 4)(*[float(v) for v in fill])
    st = _lib().avref_image_undistort(dst.ctypes.data_as(C.c_void_p), w * 16, src.ctypes.data_as(C.c_void_p), w * 16, C.cast(C.byref(cam), C.c_void_p),
                                      C.cast(f, C.c_void_p))
    if st != 0:
        raise RuntimeError("avref_image_undistort")
    return dst


def default_params():
    """{"group.name": "value"} of the reference's own parameter headers (SgmParams.hpp, RefineParams.hpp, DepthMapParams.hpp, TileParams.hpp)"""
    buf = C.create_string_buffer(1 << 14)
    n = _lib().avref_default_params(buf, len(buf))
    assert n > 0
    return dict(line.split("=", 1) for line in buf.value.decode().strip().splitlines())


def _landmark_arrays(landmarks):
    begin, views, xy = [0], [], []
    for _, obs in landmarks:
        for v in sorted(obs):
            views.append(v)
            xy.append(obs[v])
        begin.append(len(views))
    return np.asarray(begin, np.int32), np.asarray(views, np.int32), np.ascontiguousarray(np.asarray(xy, np.float64).reshape(-1, 2))


def nearest_cams(K, Rs, landmarks, rc, nb, tcams=None, roi=None, process_downscale=1, min_angle=2.0, max_angle=70.0):
    """MultiViewParams::findNearestCamsFromLandmarks of the reference (tcams None) or findTileNearestCams (tcams, roi at process
    resolution): the list of T camera indices.  K: full-resolution pinhole matrix shared by the views, Rs: world -> camera rotations."""
    lib = _lib()
    n = len(Rs)
    K = np.asarray(K, np.float64)
    k4 = np.ascontiguousarray(np.tile(np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], np.float64), (n, 1)))
    R = np.ascontiguousarray(np.stack([np.asarray(r, np.float64) for r in Rs]).reshape(n, 9))
    begin, views, xy = _landmark_arrays(landmarks)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    tc = np.asarray(list(tcams) if tcams is not None else [0], np.int32)
    r = np.asarray(roi if roi is not None else (0, 0, 0, 0), np.int32)
    out = np.zeros(max(n, 1), np.int32)
    k = lib.avref_nearest_cams(n, k4.ctypes.data_as(dp), R.ctypes.data_as(dp), int(process_downscale), float(min_angle), float(max_angle), len(landmarks),
                               begin.ctypes.data_as(ip), views.ctypes.data_as(ip), xy.ctypes.data_as(dp), int(rc), int(nb),
                               -1 if tcams is None else len(tc), tc.ctypes.data_as(ip), r.ctypes.data_as(ip), out.ctypes.data_as(ip))
    if k < 0:
        raise RuntimeError("avref_nearest_cams: the reference threw (see stderr)")
    return [int(v) for v in out[:k]]


def tile_roi_list(buffer_w, buffer_h, padding, image_w, image_h, max_downscale):
    """mvsUtils::getTileRoiList of the reference (TileParams.cpp compiled whole): [(x0, x1, y0, y1)]"""
    out = np.zeros((4096, 4), np.int32)
    n = _lib().avref_tile_roi_list(int(buffer_w), int(buffer_h), int(padding), int(image_w), int(image_h), int(max_downscale),
                                   out.ctypes.data_as(C.POINTER(C.c_int)), 4096)
    return [tuple(int(v) for v in out[i]) for i in range(n)]


def tile_weight_map(roi, image_w, image_h, padding, downscale):
    """addSingleTileMapWeighted of the reference (mapIO.cpp:206-311) applied to a tile of ones: (weights of the tile at roi / downscale,
    the full map at image / downscale after the addition)"""
    x0, x1, y0, y1 = roi
    tw = (x1 + downscale - 1) // downscale - x0 // downscale
    th = (y1 + downscale - 1) // downscale - y0 // downscale
    fw, fh = (image_w + downscale - 1) // downscale, (image_h + downscale - 1) // downscale
    w = np.zeros((th, tw), np.float32)
    full = np.zeros((fh, fw), np.float32)
    r = np.asarray(roi, np.int32)
    fp = C.POINTER(C.c_float)
    st = _lib().avref_tile_weight_map(int(image_w), int(image_h), r.ctypes.data_as(C.POINTER(C.c_int)), int(padding), int(downscale), w.ctypes.data_as(fp),
                                      full.ctypes.data_as(fp))
    if st != 0:
        raise RuntimeError("avref_tile_weight_map")
    return w, full


def depth_list(K, Rs, Cs, width, height, landmarks, rc, tcams, roi, process_downscale=1, min_angle=2.0, max_angle=70.0, sgm_scale=2, max_depths=1500,
               step_z=-1, seeds_range_inflate=0.2, use_sfm_seeds=True, depth_list_per_tile=False):
    """SgmDepthList::computeListRc of the reference for one tile: (depths float32 array, [(first, count)] per T camera), ([], []) when the
    reference produces no list.  Same scene description as host_oracle.Cameras / host_oracle.depth_list: full-resolution K, R_i, C_i and
    image size, landmarks = [(X, {view index: (u, v)})] with full-resolution observations, roi = (x0, x1, y0, y1) at process resolution."""
    lib = _lib()
    n = len(Rs)
    K = np.asarray(K, np.float64)
    P = np.stack([K @ np.concatenate([np.asarray(Rs[i], np.float64), (-np.asarray(Rs[i], np.float64) @ np.asarray(Cs[i], np.float64))[:, None]], axis=1)
                  for i in range(n)]).astype(np.float64)
    P = np.ascontiguousarray(P.reshape(n, 12))
    w = np.full(n, width // process_downscale, np.int32)
    h = np.full(n, height // process_downscale, np.int32)
    X = np.ascontiguousarray(np.array([l[0] for l in landmarks], np.float64).reshape(-1, 3))
    begin, views, xy = [0], [], []
    for _, obs in landmarks:
        for v in sorted(obs):
            views.append(v)
            xy.append(obs[v])
        begin.append(len(views))
    begin = np.asarray(begin, np.int32)
    views = np.asarray(views, np.int32)
    xy = np.ascontiguousarray(np.asarray(xy, np.float64).reshape(-1, 2))
    tc = np.asarray(list(tcams), np.int32)
    r = np.asarray(roi, np.int32)
    cap = 65536
    out = np.zeros(cap, np.float32)
    out_n = C.c_int(0)
    lim = np.zeros((max(len(tc), 1), 2), np.int32)
    dp, ip, fp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_float)
    st = lib.avref_sgm_depth_list(n, P.ctypes.data_as(dp), w.ctypes.data_as(ip), h.ctypes.data_as(ip), int(process_downscale), float(min_angle), float(max_angle),
                                  len(landmarks), X.ctypes.data_as(dp), begin.ctypes.data_as(ip), views.ctypes.data_as(ip), xy.ctypes.data_as(dp), int(rc),
                                  len(tc), tc.ctypes.data_as(ip), r.ctypes.data_as(ip), int(sgm_scale), int(max_depths), int(step_z), float(seeds_range_inflate),
                                  int(bool(use_sfm_seeds)), int(bool(depth_list_per_tile)), out.ctypes.data_as(fp), cap, C.byref(out_n), lim.ctypes.data_as(ip))
    if st != 0:
        raise RuntimeError("avref_sgm_depth_list: the reference threw (see stderr)")
    k = out_n.value
    if k == 0:
        return np.zeros(0, np.float32), []
    return out[:k].copy(), [(int(a), int(b)) for a, b in lim[:len(tc)]]
