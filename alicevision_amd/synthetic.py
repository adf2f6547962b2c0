"""Seeded synthetic multi-view scenes: pinhole cameras on an arc looking at a textured relief surface.

Everything is analytic (surface, texture, camera rays), so images of any size can be rendered on the CPU or on the GPU
with plain torch ops and the exact depth of every reference pixel is known.  Used by tests/, bench.py and smoke().
"""
import math

import numpy as np
import torch


class Scene:
    def __init__(self):
        self.images = None   # (N, H, W, 4) float32 linear RGBA in [0, 1], alpha = 1
        self.K = None        # (3, 3) float64 (numpy)
        self.R = None        # list of (3, 3) float64: world -> camera
        self.C = None        # list of (3,) float64 camera centres
        self.gt_depth = None  # (H, W) float32: distance camera centre -> surface along the pixel ray, view 0
        self.z_range = None  # (zmin, zmax) of the surface along the optical axis of view 0
        self.width = self.height = 0


def _surface(x, y, z0, amp):
    return z0 + amp * torch.sin(1.3 * x + 0.4) * torch.cos(1.1 * y - 0.3)


def _texture(x, y, waves):
    # waves: (n, 5) = kx, ky, phase, amplitude, channel-mix
    out = []
    for ch in range(3):
        acc = torch.full_like(x, 0.5)
        for kx, ky, ph, a, mix in waves:
            acc = acc + a * torch.sin(kx * x + ky * y + ph + 2.1 * ch * mix)
        out.append(acc.clamp(0.02, 0.98))
    return out


def look_at_rotation(C, target):
    """world -> camera rotation (rows = camera axes): +z towards target, +x right, +y down (world +y is 'down')."""
    z = target - C
    z = z / np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=0)


def make_scene(n_views=3, width=640, height=480, seed=1, device="cpu", z0=4.0, amp=0.2, baseline=0.3, focal=None, dtype=torch.float32, render=None):
    """Render `n_views` images of size width x height.  View 0 is the reference camera (centre of the arc).
    render = list of view indices: only those are rendered and `images` is a dict {view: (H, W, 4)} (a rank of a multi-GPU job renders
    the views it owns; the cameras of all views are always set)."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    sc.width, sc.height = width, height
    f = focal if focal is not None else 600.0 * width / 640.0
    sc.K = np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]])
    target = np.array([0.0, 0.0, z0])
    # cameras: view 0 in the middle, the others alternate left/right/up/down on an arc
    offs = [(0.0, 0.0)]
    k = 1
    while len(offs) < n_views:
        ring = (k + 3) // 4
        ang = (k % 4) * (math.pi / 2.0) + 0.35 * ring
        offs.append((baseline * ring * math.cos(ang), baseline * ring * 0.8 * math.sin(ang)))
        k += 1
    sc.C = [np.array([ox, oy, 0.02 * i]) for i, (ox, oy) in enumerate(offs)]
    sc.R = [look_at_rotation(c, target) for c in sc.C]

    # texture: 3 octaves of oriented sinusoids, wavelengths expressed in reference-image pixels at depth z0
    px = z0 / f
    waves = []
    for lam_px, a in ((7.0, 0.10), (13.0, 0.12), (29.0, 0.14), (61.0, 0.10)):
        for _ in range(3):
            th = rng.uniform(0, math.pi)
            kk = 2.0 * math.pi / (lam_px * px)
            waves.append((kk * math.cos(th), kk * math.sin(th), rng.uniform(0, 2 * math.pi), a / 1.7, rng.uniform(0.3, 1.0)))

    dev = torch.device(device)
    v, u = torch.meshgrid(torch.arange(height, device=dev, dtype=torch.float64), torch.arange(width, device=dev, dtype=torch.float64), indexing="ij")
    Kinv = np.linalg.inv(sc.K)
    imgs = []
    for i in (range(n_views) if render is None else render):
        M = sc.R[i].T @ Kinv  # pixel -> world ray direction
        dx = M[0, 0] * u + M[0, 1] * v + M[0, 2]
        dy = M[1, 0] * u + M[1, 1] * v + M[1, 2]
        dz = M[2, 0] * u + M[2, 1] * v + M[2, 2]
        cx, cy, cz = [float(t) for t in sc.C[i]]
        t = (z0 - cz) / dz
        for _ in range(12):
            x = cx + t * dx
            y = cy + t * dy
            t = (_surface(x, y, z0, amp) - cz) / dz
        x = cx + t * dx
        y = cy + t * dy
        r, g, b = _texture(x, y, waves)
        img = torch.stack([r, g, b, torch.ones_like(r)], dim=-1).to(dtype)
        imgs.append(img)
        if i == 0:
            sc.gt_depth = (t * torch.sqrt(dx * dx + dy * dy + dz * dz)).to(torch.float32)
    sc.images = torch.stack(imgs, dim=0).contiguous() if render is None else dict(zip(render, imgs))
    sc.z_range = (z0 - amp - 0.02, z0 + amp + 0.02)
    return sc


def plane_depths(scene, n_planes, margin=0.15):
    """Fronto-parallel plane list for view 0 (distances along its optical axis), ascending — stands in for SgmDepthList
    in kernel-level tests and benchmarks ('kernel micro-benchmarks bypass SgmDepthList')."""
    zmin, zmax = scene.z_range
    span = zmax - zmin
    lo, hi = zmin - margin * span, zmax + margin * span
    # uniform in inverse depth like the epipolar sampling of the reference produces
    inv = np.linspace(1.0 / lo, 1.0 / hi, n_planes)
    return (1.0 / inv).astype(np.float32)
