"""Per-tile driver of the gfx950 kernels through the C ABI (include/avdm.h), with torch tensors as device memory.

This is harness glue for tests/, bench.py and smoke(): it sequences the entry points exactly like the reference's
Sgm::sgmRc (Sgm.cpp:117-188,203-325), Sgm::smoothThicknessMap (:190-201) and Refine::refineRc (Refine.cpp:97-272).
The production host (tiling, depth lists, I/O, CLI) is the C++ code under alicevision_amd/host/.
There is NO fallback: every stage calls libavdm.so and raises AvdmError on a non-zero status.
"""
import ctypes as C
import os

import torch

from . import abi


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ceil_div(a, b):
    return (a + b - 1) // b


def optimize_scratch(lib, X, Y, Z, device="cuda"):
    """scratch of avdm_volume_optimize (the per-axis adaptive-P2 map)"""
    n = int(lib.avdm_volume_optimize_scratch_bytes(X, Y, Z))
    return torch.empty(max(n, 4), dtype=torch.uint8, device=device)


def optimize_tiles_batched(tiles, rc, timers=None):
    """avdm_volume_optimize_tiles over the pending volumes of several DepthMapTile objects (run_sgm(..., optimize="defer")): all tiles of a
    depth map aggregated by one launch per axis, as host/DepthMapEstimator.cpp does for every group of tiles; then each tile's WTA map."""
    lib = tiles[0].lib
    sp = tiles[0].sgm
    n = len(tiles)
    arr = (abi.SgmTile * n)()
    total = 0
    sizes = []
    for t in tiles:
        AX, AY = t.sgm_extent()
        Z = t._sgm_pending[1]
        sizes.append(int(lib.avdm_volume_optimize_scratch_bytes(AX, AY, Z)))
        total += sizes[-1]
    dev = tiles[0].best.device
    key = (n, total)
    if getattr(tiles[0], "_batch_scratch_key", None) != key:
        tiles[0]._batch_scratch = torch.empty(max(total, 4), dtype=torch.uint8, device=dev)
        tiles[0]._batch_scratch_key = key
    for i, t in enumerate(tiles):
        AX, _ = t.sgm_extent()
        Z = t._sgm_pending[1]
        arr[i] = abi.SgmTile(t.best.data_ptr(), t.second.data_ptr(), AX * t.Zp, t.Zp, Z, t.sgm_extent_roi(), C.pointer(t.pyr[rc].desc))
    # the adaptive-P2 maps of all tiles first (their own stage: they depend only on the R pyramid), then the path launches alone
    for name, fn in (("sgm_p2_map", lib.avdm_volume_optimize_prepare), ("sgm_optimize", lib.avdm_volume_optimize_tiles_prepared)):
        rng = timers.range(name) if timers is not None else None
        if rng is not None:
            rng.__enter__()
        abi.check(fn(n, arr, _ptr(tiles[0]._batch_scratch), C.byref(sp), _stream()), name)
        if rng is not None:
            rng.__exit__(None, None, None)
    for t in tiles:
        t.finish_sgm(rc, t._sgm_pending[1])


class StageTimers:
    """HIP-event timing of stage ranges on the current stream (torch events wrap hipEvent on the same stream)."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    class _Range:
        def __init__(self, owner, name):
            self.o, self.name = owner, name

        def __enter__(self):
            if self.o.enabled:
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()

        def __exit__(self, *exc):
            if self.o.enabled:
                self.b.record()
                self.o.events.setdefault(self.name, []).append((self.a, self.b))

    def range(self, name):
        return StageTimers._Range(self, name)

    def reset(self):
        self.events = {}

    def mean_ms(self, per=None):
        """mean ms per occurrence (or per `per` steps) of every stage; call after a device synchronize"""
        out = {}
        for k, lst in self.events.items():
            tot = sum(a.elapsed_time(b) for a, b in lst)
            out[k] = tot / (per if per else len(lst))
        return out


class DevicePyramid:
    """fp16 Lab mip pyramid in HBM (replaces DeviceMipmapImage)."""

    def __init__(self, rgba, min_downscale, max_downscale, filter_mode, device="cuda", storage=None):
        h, w = rgba.shape[:2]
        self._layout(w, h, min_downscale, max_downscale, filter_mode, device, storage)
        if rgba is not None:
            self.fill(rgba)

    def _layout(self, width, height, min_downscale, max_downscale, filter_mode, device, storage):
        """storage: a byte tensor of pyramid_bytes() the pyramid lives in instead of an allocation of its own (a slot of the multi-GPU exchange
        arena, sharding.ViewExchange.buffer)"""
        lib = abi.load()
        self.desc = abi.Pyramid()
        abi.check(lib.avdm_pyramid_layout(C.byref(self.desc), width, height, min_downscale, max_downscale, filter_mode), "avdm_pyramid_layout")
        if storage is None:
            storage = torch.zeros(self.desc.bytes, dtype=torch.uint8, device=device)
        if storage.dtype != torch.uint8 or storage.numel() != self.desc.bytes or not storage.is_contiguous():
            raise ValueError("pyramid storage must be a contiguous byte tensor of %d bytes" % self.desc.bytes)
        self.buf = storage
        self.desc.base = self.buf.data_ptr()

    @staticmethod
    def pyramid_bytes(width, height, min_downscale, max_downscale, filter_mode):
        d = abi.Pyramid()
        abi.check(abi.load().avdm_pyramid_layout(C.byref(d), width, height, min_downscale, max_downscale, filter_mode), "avdm_pyramid_layout")
        return int(d.bytes)

    @classmethod
    def allocate(cls, width, height, min_downscale, max_downscale, filter_mode, device="cuda", storage=None):
        """pyramid storage without content (to be filled later or received from another rank)"""
        self = cls.__new__(cls)
        self._layout(width, height, min_downscale, max_downscale, filter_mode, device, storage)
        return self

    def fill(self, rgba):
        lib = abi.load()
        rgba = rgba.to(device=self.buf.device, dtype=torch.float32).contiguous()
        h, w = rgba.shape[:2]
        scratch = None
        if self.desc.min_downscale > 1:
            scratch = torch.empty(h * w * 8, dtype=torch.uint8, device=self.buf.device)
        abi.check(lib.avdm_pyramid_fill(C.byref(self.desc), _ptr(rgba), w * 16, _ptr(scratch) if scratch is not None else None, _stream()),
                  "avdm_pyramid_fill")
        self._keep = (rgba, scratch)

    @classmethod
    def from_host_bytes(cls, desc_like, raw_bytes, device="cuda"):
        """Upload a pyramid built elsewhere (e.g. by the oracle) so later stages can be compared on identical inputs."""
        self = cls.__new__(cls)
        self.desc = abi.Pyramid()
        C.memmove(C.byref(self.desc), C.byref(desc_like), C.sizeof(abi.Pyramid))
        self.buf = torch.from_numpy(raw_bytes).to(device)
        self.desc.base = self.buf.data_ptr()
        return self

    def level(self, l):
        d = self.desc
        raw = self.buf[d.offset[l]:d.offset[l] + d.pitch[l] * d.height[l]].view(d.height[l], d.pitch[l])
        return raw[:, :d.width[l] * 8].contiguous().view(torch.float16).view(d.height[l], d.width[l], 4)


class DepthMapTile:
    """One tile of one R camera: SGM (similarity volume, path aggregation, WTA) then Refine (re-sweep, sub-sample arg-min,
    colour optimisation).  roi = (x0, x1, y0, y1) in process-resolution pixels.

    tile_buffer = (bufferWidth, bufferHeight) of the tile workflow (TileParams).  The reference allocates a tile's volumes and maps for the
    tile BUFFER (Sgm.cpp:37-72), sweeps the ROI into their corner, and its path aggregation walks the ALLOCATED extent
    (cuda_volumeAggregatePath takes X / Y from the volume, deviceSimilarityVolume.cu:278-283), so the reverse paths cross the 255-filled
    remainder of the buffer before they enter the ROI; the upscale ratio comes from the allocated map widths
    (deviceDepthSimilarityMap.cu:115-117).  None = a buffer the size of the ROI (single whole-image tile)."""

    def __init__(self, pyramids, K, Rs, Cs, sgm, refine, roi=None, device="cuda", tile_buffer=None):
        self.lib = abi.load()
        self.pyr = pyramids
        self.K, self.Rs, self.Cs = K, Rs, Cs
        self.sgm, self.refine = sgm, refine
        self.device = device
        W, H = pyramids[0].desc.width0, pyramids[0].desc.height0
        self.roi = roi if roi is not None else (0, W, 0, H)
        self.tile_buffer = tile_buffer
        self._alloc_for = None
        self._side = None  # side stream of the adaptive-P2 maps (run_sgm)
        self.timers = StageTimers()

    def enable_timers(self, on=True):
        self.timers.enabled = on

    def reset_timers(self):
        self.timers.reset()

    def stage_ms(self):
        return self.timers.mean_ms()

    def cam(self, i, scale):
        return abi.camera_fill(self.K, self.Rs[i], self.Cs[i], scale)

    def droi(self, ds):
        x0, x1, y0, y1 = self.roi
        return abi.ROI.make(x0 // ds, ceil_div(x1, ds), y0 // ds, ceil_div(y1, ds))

    def sgm_extent(self):
        """(X, Y) the SGM volumes are laid out for and aggregated over: the downscaled tile buffer (the ROI without one)"""
        ds = self.sgm.scale * self.sgm.stepXY
        roi = self.droi(ds)
        if self.tile_buffer is None:
            return roi.width, roi.height
        AX, AY = ceil_div(self.tile_buffer[0], ds), ceil_div(self.tile_buffer[1], ds)
        if AX < roi.width or AY < roi.height:
            raise ValueError("the tile does not fit its buffer")
        return AX, AY

    def sgm_extent_roi(self):
        """the ROI handed to the path aggregation: image coordinates start at the tile's ROI, the extent is the laid-out volume"""
        roi = self.droi(self.sgm.scale * self.sgm.stepXY)
        AX, AY = self.sgm_extent()
        return abi.ROI.make(roi.x.begin, roi.x.begin + AX, roi.y.begin, roi.y.begin + AY)

    def _alloc(self, Z):
        sp, rp = self.sgm, self.refine
        roiS, roiR = self.droi(sp.scale * sp.stepXY), self.droi(rp.scale * rp.stepXY)
        AX, AY = self.sgm_extent()
        key = (Z, roiS.width, roiS.height, roiR.width, roiR.height, AX, AY)
        if self._alloc_for == key:
            return
        dev = self.device
        X, Y = roiS.width, roiS.height
        Zp = ceil_div(Z, 4) * 4
        self.Zp = Zp
        self.best = torch.empty((AY, AX, Zp), dtype=torch.uint8, device=dev)
        self.second = torch.empty((AY, AX, Zp), dtype=torch.uint8, device=dev)
        self.depths_d = torch.empty(Z, dtype=torch.float32, device=dev)
        self.sgm_scratch = optimize_scratch(self.lib, AX, AY, Z, dev)
        self.sgm_depth_thickness = torch.empty((Y, X, 2), dtype=torch.float32, device=dev)
        self.sgm_depth_sim = torch.empty((Y, X, 2), dtype=torch.float32, device=dev)
        XR, YR = roiR.width, roiR.height
        Zr = rp.halfNbDepths * 2 + 1
        self.Zr, self.Zrp = Zr, ceil_div(Zr, 8) * 8
        self.sgm_upscaled = torch.empty((YR, XR, 2), dtype=torch.float32, device=dev)
        self.refined = torch.empty((YR, XR, 2), dtype=torch.float32, device=dev)
        self.optimized = torch.empty((YR, XR, 2), dtype=torch.float32, device=dev)
        self.refine_volume = torch.empty((YR, XR, self.Zrp), dtype=torch.float16, device=dev)
        self.img_variance = torch.empty((YR, XR), dtype=torch.float32, device=dev)
        self.tmp_depth = torch.empty((YR, XR), dtype=torch.float32, device=dev)
        self._alloc_for = key

    # ---- Sgm::sgmRc ----
    def run_sgm(self, rc, tcs, depths, tc_ranges=None, optimize=True, keep_raw=False):
        lib, sp = self.lib, self.sgm
        Z = len(depths)
        self._alloc(Z)
        roi = self.droi(sp.scale * sp.stepXY)
        AX, AY = self.sgm_extent()
        Zp = self.Zp
        py, pxx = AX * Zp, Zp
        st = _stream()
        self.depths_d.copy_(torch.as_tensor(depths, dtype=torch.float32), non_blocking=False)
        T = self.timers
        # The adaptive-P2 maps of the path aggregation depend on nothing but the R pyramid: they are evaluated on a side stream beside the
        # similarity sweep (avdm_volume_optimize_prepare), and the aggregation further down is the path launches alone.
        # (AVDM_SGM_PREPARE=0: the one-call form avdm_volume_optimize, the A/B reference; the bytes are the same.)
        prepared = None
        if optimize is True and os.environ.get("AVDM_SGM_PREPARE") != "0":
            tile1 = (abi.SgmTile * 1)(abi.SgmTile(self.best.data_ptr(), self.second.data_ptr(), py, pxx, Z, self.sgm_extent_roi(), C.pointer(self.pyr[rc].desc)))
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.best.device)
            ready = torch.cuda.Event()
            ready.record()  # the R pyramid is complete in the caller's stream order
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                with T.range("sgm_p2_map"):
                    abi.check(lib.avdm_volume_optimize_prepare(1, tile1, _ptr(self.sgm_scratch), C.byref(sp), _stream()), "volume_optimize_prepare")
                done = torch.cuda.Event()
                done.record()
            prepared = (tile1, done)
        with T.range("sgm_volume_init"):
            abi.check(lib.avdm_volume_initialize_u8(_ptr(self.best), py, pxx, AX, AY, Zp, 255, st), "volume_initialize")
            abi.check(lib.avdm_volume_initialize_u8(_ptr(self.second), py, pxx, AX, AY, Zp, 255, st), "volume_initialize")
        rcCam = self.cam(rc, sp.scale)
        with T.range("sgm_similarity"):
            for ti, tc in enumerate(tcs):
                tcCam = self.cam(tc, sp.scale)
                r = tc_ranges[ti] if tc_ranges else (0, Z)
                abi.check(lib.avdm_volume_compute_similarity(_ptr(self.best), _ptr(self.second), py, pxx, _ptr(self.depths_d), C.byref(rcCam),
                                                             C.byref(tcCam), C.byref(self.pyr[rc].desc), C.byref(self.pyr[tc].desc), C.byref(sp),
                                                             abi.Range(r[0], r[1]), roi, st), "volume_compute_similarity")
        with T.range("sgm_update_uninit"):
            abi.check(lib.avdm_volume_update_uninitialized(_ptr(self.best), _ptr(self.second), py, pxx, AX, AY, Z, st), "update_uninitialized")
        if keep_raw:
            self.best_raw = self.best.clone()
        if optimize == "defer":  # the caller aggregates the volumes of several tiles in one batched call (optimize_tiles_batched)
            self._sgm_pending = (rc, Z)
            return None
        if optimize and prepared is not None:
            torch.cuda.current_stream().wait_event(prepared[1])
            with T.range("sgm_optimize"):
                abi.check(lib.avdm_volume_optimize_tiles_prepared(1, prepared[0], _ptr(self.sgm_scratch), C.byref(sp), st), "volume_optimize_tiles_prepared")
        elif optimize:
            with T.range("sgm_optimize"):
                abi.check(lib.avdm_volume_optimize(_ptr(self.best), _ptr(self.second), py, pxx, _ptr(self.sgm_scratch), C.byref(self.pyr[rc].desc),
                                                   C.byref(sp), Z, self.sgm_extent_roi(), st), "volume_optimize")
        else:
            self.best.copy_(self.second)
        return self.finish_sgm(rc, Z)

    def finish_sgm(self, rc, Z):
        """winner-take-all depth / thickness from the aggregated volume (second half of Sgm::sgmRc)"""
        lib, sp = self.lib, self.sgm
        roi = self.droi(sp.scale * sp.stepXY)
        X, Zp = roi.width, self.Zp
        py, pxx = self.sgm_extent()[0] * Zp, Zp
        st = _stream()
        T = self.timers
        rc1 = self.cam(rc, 1)
        with T.range("sgm_retrieve_best_depth"):
            abi.check(lib.avdm_volume_retrieve_best_depth(_ptr(self.sgm_depth_thickness), X * 8, _ptr(self.sgm_depth_sim), X * 8,
                                                          _ptr(self.depths_d), _ptr(self.best), py, pxx, Z, C.byref(rc1), C.byref(sp),
                                                          abi.Range(0, Z), roi, st), "retrieve_best_depth")
        return self.sgm_depth_thickness, self.sgm_depth_sim

    # ---- Sgm::smoothThicknessMap + Refine::refineRc ----
    def run_refine(self, rc, tcs, refine_enabled=True, optimize_enabled=True):
        lib, sp, rp = self.lib, self.sgm, self.refine
        roiS, roiR = self.droi(sp.scale * sp.stepXY), self.droi(rp.scale * rp.stepXY)
        st = _stream()
        T = self.timers
        X, Y = roiR.width, roiR.height
        rcCam = self.cam(rc, rp.scale)
        if self.tile_buffer is None:
            ratio = float(roiS.width) / float(X)  # allocated widths == ROI widths
        else:  # the ratio of the ALLOCATED map widths (deviceDepthSimilarityMap.cu:115-117)
            ratio = float(ceil_div(self.tile_buffer[0], sp.scale * sp.stepXY)) / float(ceil_div(self.tile_buffer[0], rp.scale * rp.stepXY))
        with T.range("smooth_and_upscale"):
            abi.check(lib.avdm_depth_thickness_smooth_thickness(_ptr(self.sgm_depth_thickness), roiS.width * 8, C.byref(sp), C.byref(rp), roiS, st),
                      "smooth_thickness")
            abi.check(lib.avdm_compute_sgm_upscaled_depth_pixsize_map(_ptr(self.sgm_upscaled), X * 8, _ptr(self.sgm_depth_thickness),
                                                                      roiS.width * 8, C.byref(rcCam), C.byref(self.pyr[rc].desc), C.byref(rp), ratio,
                                                                      roiR, st), "sgm_upscale")
        Zr, Zrp = self.Zr, self.Zrp
        py, pxx = X * Zrp * 2, Zrp * 2
        if refine_enabled:
            with T.range("refine_volume_init"):
                abi.check(lib.avdm_volume_initialize_f16(_ptr(self.refine_volume), py, pxx, X, Y, Zrp, 0.0, st), "volume_initialize_f16")
            with T.range("refine_similarity"):
                for tc in tcs:
                    tcCam = self.cam(tc, rp.scale)
                    abi.check(lib.avdm_volume_refine_similarity(_ptr(self.refine_volume), py, pxx, Zr, _ptr(self.sgm_upscaled), X * 8, None, 0,
                                                                C.byref(rcCam), C.byref(tcCam), C.byref(self.pyr[rc].desc),
                                                                C.byref(self.pyr[tc].desc), C.byref(rp), abi.Range(0, Zr), roiR, st),
                              "refine_similarity")
            with T.range("refine_best_depth"):
                abi.check(lib.avdm_volume_refine_best_depth(_ptr(self.refined), X * 8, _ptr(self.sgm_upscaled), X * 8, _ptr(self.refine_volume), py,
                                                            pxx, Zr, C.byref(rp), roiR, st), "refine_best_depth")
        else:
            abi.check(lib.avdm_depth_sim_map_copy_depth_only(_ptr(self.refined), X * 8, _ptr(self.sgm_upscaled), X * 8, X, Y, 1.0, st),
                      "copy_depth_only")
        if optimize_enabled and rp.optimizationNbIterations > 0:
            with T.range("color_optimize"):
                abi.check(lib.avdm_depth_sim_map_optimize_gradient_descent(_ptr(self.optimized), X * 8, _ptr(self.img_variance), X * 4,
                                                                           _ptr(self.tmp_depth), X * 4, X, Y, _ptr(self.sgm_upscaled), X * 8,
                                                                           _ptr(self.refined), X * 8, C.byref(rcCam), C.byref(self.pyr[rc].desc),
                                                                           C.byref(rp), roiR, st), "optimize_gradient_descent")
        else:
            self.optimized.copy_(self.refined)
        return self.optimized
