"""Inputs for the depth-map filtering tests: depth / similarity maps of every view of the analytic scene (alicevision_amd.synthetic),
with controlled defects (noise, outliers, masked and weakly supported pixels)."""
import os

import numpy as np
import torch

from alicevision_amd import exr_io, scene_io
from alicevision_amd.synthetic import _surface, look_at_rotation


def view_depth_map(sc, i, z0=4.0, amp=0.2, device="cpu"):
    """distance camera centre -> surface along the ray of every pixel of view i (float64 tensor H x W)"""
    dev = torch.device(device)
    v, u = torch.meshgrid(torch.arange(sc.height, device=dev, dtype=torch.float64), torch.arange(sc.width, device=dev, dtype=torch.float64), indexing="ij")
    M = sc.R[i].T @ np.linalg.inv(sc.K)
    dx = M[0, 0] * u + M[0, 1] * v + M[0, 2]
    dy = M[1, 0] * u + M[1, 1] * v + M[1, 2]
    dz = M[2, 0] * u + M[2, 1] * v + M[2, 2]
    cx, cy, cz = [float(t) for t in sc.C[i]]
    t = (z0 - cz) / dz
    for _ in range(14):
        t = (_surface(cx + t * dx, cy + t * dy, z0, amp) - cz) / dz
    return t * torch.sqrt(dx * dx + dy * dy + dz * dz)


class FuseScene:
    pass


def make_fuse_scene(n_views=5, width=160, height=120, seed=7, noise=0.0, outliers=0.0, masked=0.0, weak=0.0, device="cpu"):
    """K, R, C of n_views cameras and their depth / sim maps (float32 numpy, H x W).
    noise: relative Gaussian depth noise; outliers: fraction of pixels with a wrong depth; masked: fraction set to depth -2 (alpha mask);
    weak: fraction of pixels flagged weakly supported (sim += 2, i.e. >= 1)."""
    rng = np.random.RandomState(seed)
    fs = FuseScene()
    fs.width, fs.height, fs.n = width, height, n_views
    f = 600.0 * width / 640.0
    fs.K = np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])
    z0 = 4.0
    target = np.array([0.0, 0.0, z0])
    fs.C, fs.R = [], []
    for i in range(n_views):
        ang = 2.0 * np.pi * i / max(n_views, 1) + 0.3
        r = 0.0 if i == 0 else 0.25 + 0.1 * (i % 3)
        c = np.array([r * np.cos(ang), 0.8 * r * np.sin(ang), 0.03 * i])
        fs.C.append(c)
        fs.R.append(look_at_rotation(c, target))
    geo = FuseScene()
    geo.width, geo.height, geo.K, geo.R, geo.C = width, height, fs.K, fs.R, fs.C
    fs.depth, fs.sim, fs.exact = [], [], []
    for i in range(n_views):
        d = view_depth_map(geo, i, device=device).cpu().numpy()
        fs.exact.append(d.astype(np.float32))
        d = d * (1.0 + noise * rng.standard_normal(d.shape))
        m = rng.uniform(size=d.shape) < outliers
        d = np.where(m, d * rng.uniform(0.6, 1.5, size=d.shape), d)
        s = rng.uniform(-0.95, -0.2, size=d.shape)
        wk = rng.uniform(size=d.shape) < weak
        s = np.where(wk, s + 2.0, s)
        inval = rng.uniform(size=d.shape) < 0.02
        d = np.where(inval, -1.0, d)
        s = np.where(inval, 1.0, s)
        mk = rng.uniform(size=d.shape) < masked
        d = np.where(mk, -2.0, d)
        fs.depth.append(np.ascontiguousarray(d, np.float32))
        fs.sim.append(np.ascontiguousarray(s, np.float32))
    return fs


def camera_structs(fs, maker, cls=None):
    """one camera struct per view through `maker(P, iP, C, w, h)` (oracle.fuse_oracle.fuse_cam or alicevision_amd.fuse.fuse_camera)"""
    from oracle.fuse_oracle import camera_arrays
    out = []
    for i in range(fs.n):
        P, iP, Cc = camera_arrays(fs.K, fs.R[i], fs.C[i])
        out.append(maker(P, iP, Cc, fs.width, fs.height))
    return out


def write_depth_maps(folder, sc, depths, sims, downscale=1):
    """<viewId>_depthMap.exr / _simMap.exr as aliceVision_depthMapEstimation writes them (float depth, half sim, AliceVision:P +
    AliceVision:downscale metadata, mapIO.cpp:402-540)"""
    os.makedirs(folder, exist_ok=True)
    for i in range(len(depths)):
        P = sc.K @ np.concatenate([sc.R[i], (-sc.R[i] @ sc.C[i])[:, None]], axis=1)
        attrs = {"AliceVision:P": exr_io.m44d(list(P.flatten()) + [0, 0, 0, 1]), "AliceVision:downscale": int(downscale)}
        vid = scene_io.view_id(i)
        exr_io.write_exr(os.path.join(folder, "%d_depthMap.exr" % vid), {"Y": depths[i]}, attributes=attrs, compression=3)
        exr_io.write_exr(os.path.join(folder, "%d_simMap.exr" % vid), {"Y": sims[i]}, attributes=attrs, compression=3, half=True)
