"""Depth-map filtering on the GPU through include/avdm_fuse.h (fuseCut::Fuser::filterGroupsRC / filterDepthMapsRC,
fuseCut/Fuser.cpp:144-304).  torch is used for device memory and the stream only; there is no CPU path."""
import ctypes as C

import torch

from . import abi


def fuse_camera(P, iP, Cc, width, height):
    """camArr (3x4), iCamArr (3x3), CArr (row-major doubles) + image size -> avdm_fuse_camera_t"""
    c = abi.FuseCamera()
    c.P[:] = [float(v) for v in P.reshape(-1)]
    c.iP[:] = [float(v) for v in iP.reshape(-1)]
    c.C[:] = [float(v) for v in Cc.reshape(-1)]
    c.width, c.height = int(width), int(height)
    return c


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_map(t, w, h, dtype):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.shape[0] == h and t.shape[1] == w and t.stride(1) == 1):
        raise ValueError(f"expected a {h}x{w} {dtype} map on the GPU with unit column stride")


def filter_groups(rc_depth, rc_sim, rc_cam, tc_cams, tc_depths, pix_tolerance_factor=2.0, pix_size_ball=0, pix_size_ball_wsp=0, out=None):
    """Modal-count map (uint8, h x w) of the reference camera.  tc_depths[i] is a float32 CUDA tensor or None (camera without a map);
    row-pitched views (stride(0) > width) are accepted."""
    lib = abi.load()
    w, h = rc_cam.width, rc_cam.height
    _check_map(rc_depth, w, h, torch.float32)
    _check_map(rc_sim, w, h, torch.float32)
    n = len(tc_cams)
    tcs = (abi.FuseTc * max(n, 1))()
    for i, (cam, d) in enumerate(zip(tc_cams, tc_depths)):
        tcs[i].cam = cam
        if d is not None:
            _check_map(d, cam.width, cam.height, torch.float32)
            tcs[i].depth = d.data_ptr()
            tcs[i].depth_pitch = d.stride(0) * 4
    if out is None:
        out = torch.empty((h, w), dtype=torch.uint8, device=rc_depth.device)
    _check_map(out, w, h, torch.uint8)
    scratch = torch.empty(lib.avdm_fuse_filter_groups_scratch_bytes(w, h) // 4, dtype=torch.int32, device=rc_depth.device)
    abi.check(lib.avdm_fuse_filter_groups(out.data_ptr(), out.stride(0), rc_depth.data_ptr(), rc_depth.stride(0) * 4, rc_sim.data_ptr(),
                                          rc_sim.stride(0) * 4, C.byref(rc_cam), n, tcs, float(pix_tolerance_factor), int(pix_size_ball),
                                          int(pix_size_ball_wsp), scratch.data_ptr(), _stream()), "avdm_fuse_filter_groups")
    return out


def filter_depth_maps(depth, sim, nmod, min_num_of_modals=3, min_num_of_modals_wsp2ssp=4):
    """In place on depth / sim (float32 CUDA tensors h x w)."""
    lib = abi.load()
    h, w = depth.shape
    _check_map(depth, w, h, torch.float32)
    _check_map(sim, w, h, torch.float32)
    _check_map(nmod, w, h, torch.uint8)
    abi.check(lib.avdm_fuse_filter_depth_maps(depth.data_ptr(), depth.stride(0) * 4, sim.data_ptr(), sim.stride(0) * 4, nmod.data_ptr(), nmod.stride(0),
                                              w, h, int(min_num_of_modals), int(min_num_of_modals_wsp2ssp), _stream()), "avdm_fuse_filter_depth_maps")
    return depth, sim
