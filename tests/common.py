"""Shared helpers of the test-suite: small seeded scenes and the oracle / HIP drivers on identical inputs."""
import numpy as np

from alicevision_amd import abi
from alicevision_amd.synthetic import make_scene, plane_depths


def small_case(width=256, height=192, n_views=3, n_planes=32, seed=7, **sgm_kw):
    sc = make_scene(n_views, width, height, seed=seed)
    sgm = abi.SgmParams.default(**sgm_kw)
    ref = abi.RefineParams.default()
    depths = plane_depths(sc, n_planes)
    return sc, sgm, ref, depths


def make_oracle(sc, sgm, ref, filter_mode=abi.FILTER_CUDA_FIXED8, roi=None):
    from oracle import oracle
    return oracle.OracleDepthMap(sc.images.numpy(), sc.K, sc.R, sc.C, sgm, ref, filter_mode=filter_mode, roi=roi)


def make_hip_from_oracle(o, sc, sgm, ref, roi=None, tile_buffer=None):
    """HIP tile driver fed with the ORACLE's pyramids (bit-identical inputs for the stage-level parity tests)."""
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    pyr = [DevicePyramid.from_host_bytes(p.desc, p.buf) for p in o.pyr]
    return DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=roi, tile_buffer=tile_buffer)


def level_mismatch(a, b):
    """fraction of differing entries and max abs difference of two uint8 arrays"""
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return float((d != 0).mean()), int(d.max())


def build_host():
    """the C++ host programs (alicevision_amd/host), built under the checker's make lock (several test processes may want them at once)"""
    import os
    from oracle.oracle import locked_make
    locked_make(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alicevision_amd", "host"), "-j8")
