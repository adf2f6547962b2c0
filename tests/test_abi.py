"""CPU tests of the drop-in boundary: include/avdm.h <-> libavdm.so <-> alicevision_amd/abi.py (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np

from alicevision_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = {"avdm.h": "SIGNATURES", "avdm_fuse.h": "FUSE_SIGNATURES"}


def _declared_functions(header="avdm.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avdm_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    assert len(_declared_functions()) >= 28
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == sorted(HEADERS)
    lib = abi.load()
    for header, table in HEADERS.items():
        names, bound = _declared_functions(header), getattr(abi, table)
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{header} but not exported by libavdm.so"
            assert n in bound, f"{n} has no ctypes signature in alicevision_amd/abi.py"
        for n in bound:
            assert n in names, f"{n} bound in abi.py but not declared in include/{header}"


def test_every_entry_point_cites_the_reference():
    """each declaration is preceded by a comment naming the reference interface it replaces (file:line)"""
    for header in HEADERS:
        src = open(os.path.join(ROOT, "include", header)).read()
        for n in _declared_functions(header):
            if n in ("avdm_last_error", "avdm_version"):
                continue
            i = src.index(n + "(")
            ctx = src[max(0, i - 700):i]
            last_comment = ctx[ctx.rfind("/*"):]
            assert re.search(r"\.(cu|cuh|cpp|hpp)|:\d+", last_comment), n  # file:line (the file may be named once per header section)


def test_struct_layouts_match_the_header():
    # sizes as a C compiler lays the header's structs out (checked against the oracle library, which includes the same header)
    assert C.sizeof(abi.Camera) == 69 * 4
    assert C.sizeof(abi.ROI) == 16
    assert C.sizeof(abi.Pyramid) == 8 + 4 * 4 + 4 + 3 * 8 * 4 + 8 * 8 + 8 or C.sizeof(abi.Pyramid) % 8 == 0
    assert C.sizeof(abi.SgmParams) % 8 == 0 and C.sizeof(abi.RefineParams) % 8 == 0
    assert C.sizeof(abi.FuseCamera) == 24 * 8 + 8 and C.sizeof(abi.FuseTc) == 16 + C.sizeof(abi.FuseCamera)


def test_camera_fill_matches_oracle_and_projects(oracle_lib):
    from oracle import oracle
    rng = np.random.RandomState(4)
    K = np.array([[1500.0, 0.0, 960.0], [0.0, 1490.0, 540.0], [0.0, 0.0, 1.0]])
    a = rng.uniform(-0.3, 0.3, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    R = Rx @ Ry
    Cc = rng.uniform(-1, 1, 3)
    for ds in (1, 2, 4):
        got = abi.camera_fill(K, R, Cc, ds)
        want = oracle.camera_fill(K, R, Cc, ds)
        assert bytes(got) == bytes(want)
        # column-major P = K' [R | -R C], K' = diag(1/ds, 1/ds, 1) K  (DeviceCache.cpp:41-134)
        P = np.array(got.P[:]).reshape(4, 3).T
        X = np.array([0.3, -0.2, 5.0])
        x = P @ np.append(X, 1.0)
        Ks = np.diag([1.0 / ds, 1.0 / ds, 1.0]) @ K
        xr = Ks @ (R @ (X - Cc))
        assert np.allclose(x[:2] / x[2], xr[:2] / xr[2], rtol=1e-5)
        assert np.allclose(np.array(got.C[:]), Cc, rtol=1e-6)
        assert np.allclose(np.array(got.ZVect[:]), R[2], atol=1e-6)  # optical axis = third row of R (world coordinates)


def test_pyramid_layout_matches_oracle(oracle_lib):
    for (w, h, mn, mx) in [(4000, 3000, 1, 128), (1920, 1080, 1, 128), (250, 186, 1, 128), (641, 479, 2, 16), (64, 48, 1, 8)]:
        a, b = abi.Pyramid(), abi.Pyramid()
        assert abi.load().avdm_pyramid_layout(C.byref(a), w, h, mn, mx, abi.FILTER_CUDA_FIXED8) == 0
        assert oracle_lib.avo_pyramid_layout(C.byref(b), w, h, mn, mx, abi.FILTER_CUDA_FIXED8) == 0
        assert a.levels == b.levels and a.bytes == b.bytes
        assert list(a.width) == list(b.width) and list(a.height) == list(b.height)
        assert list(a.pitch) == list(b.pitch) and list(a.offset) == list(b.offset)
        # DeviceMipmapImage.cpp:35: levels = log2(maxDs / minDs) + 1; level dims halve with floor (cudaMallocMipmappedArray)
        assert a.levels == int(np.log2(mx / mn)) + 1 or a.width[a.levels - 1] >= 1
        for l in range(1, a.levels):
            assert a.width[l] == max(a.width[l - 1] // 2, 1) and a.height[l] == max(a.height[l - 1] // 2, 1)


def test_errors_are_reported_not_thrown():
    lib = abi.load()
    p = abi.Pyramid()
    assert lib.avdm_pyramid_layout(C.byref(p), 0, 10, 1, 8, 0) != 0
    assert len(lib.avdm_last_error()) > 0
    assert lib.avdm_volume_optimize_scratch_bytes(1000, 750, 256) >= 1000 * 750 * 4


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(abi.AvdmError):
        abi.load(str(tmp_path / "nope.so"))


def test_custom_patch_pattern_builder_matches_oracle(oracle_lib):
    """avdm_build_custom_patch_pattern is host code (patchPattern.cpp:18-251): same structure as the oracle's restatement for grouped and
    ungrouped subparts, and the reference's refusals (no subpart, bad radius / count, two full subparts in a group, too many subparts
    or coordinates) are reported as errors"""
    lib = abi.load()
    specs = [
        ([("circle", 4, 16, 0, 0.5), ("full", 3, 0, 1, 0.3), ("circle", 7.5, 24, 2, 0.2)], False),
        ([("full", 2, 0, 1, 0.4), ("circle", 3.5, 12, 0, 0.35), ("circle", 6, 10, 1, 0.25)], True),
        ([("circle", 2, 8, 0, 0.5), ("circle", 4, 8, 0, 0.25), ("circle", 6, 8, 0, 0.25)], True),   # three circles grouped into one subpart
        ([("full", 4, 0, 0, 1.0)], False),
    ]
    for spec, group in specs:
        got, want = abi.PatchPattern(), abi.PatchPattern()
        arr = abi.patch_subparts(spec)
        assert lib.avdm_build_custom_patch_pattern(len(spec), arr, int(group), C.byref(got)) == 0, lib.avdm_last_error()
        assert oracle_lib.avo_build_custom_patch_pattern(len(spec), arr, int(group), C.byref(want)) == 0
        assert got.nbSubparts == want.nbSubparts
        for a, b in zip(got.subparts[:got.nbSubparts], want.subparts[:want.nbSubparts]):
            assert (a.nbCoordinates, a.level, a.downscale, a.weight, a.isCircle, a.wsh) == (b.nbCoordinates, b.level, b.downscale, b.weight, b.isCircle,
                                                                                           b.wsh)
            ca = np.array([[c[0], c[1]] for c in a.coordinates[:a.nbCoordinates]]).reshape(-1, 2)
            cb = np.array([[c[0], c[1]] for c in b.coordinates[:b.nbCoordinates]]).reshape(-1, 2)
            assert np.allclose(ca, cb, rtol=0, atol=2e-6)
            if a.isCircle:
                assert np.allclose(np.hypot(ca[:, 0], ca[:, 1]).reshape(-1, 1), np.hypot(cb[:, 0], cb[:, 1]).reshape(-1, 1), atol=1e-5)
    bad = [
        ([], False),
        ([("circle", 0.0, 8, 0, 1.0)], False),
        ([("circle", 3.0, 0, 0, 1.0)], False),
        ([("full", 2, 0, 1, 0.5), ("full", 3, 0, 1, 0.5)], True),
        ([("full", 1, 0, l, 0.2) for l in range(5)], False),
        ([("circle", 3, 25, 0, 1.0)], False),
        ([("circle", 3, 16, 0, 0.5), ("circle", 5, 16, 0, 0.5)], True),   # 32 coordinates in one group
    ]
    for spec, group in bad:
        arr = abi.patch_subparts(spec) if spec else None
        assert lib.avdm_build_custom_patch_pattern(len(spec), arr, int(group), None) != 0, spec
        assert b"custom patch pattern" in lib.avdm_last_error()
        assert oracle_lib.avo_build_custom_patch_pattern(len(spec), arr, int(group), None) != 0, spec


def test_counter_summaries_describe_the_kernels_that_ship():
    """bench.py quotes hardware-counter figures (roofline.traffic, similarity.valu_issue_frac) from the committed summaries under profiles/ only when
    they were taken with the kernel source that is being built — or, for the similarity kernels, with one whose kernels compile to the same
    instructions (scripts/isa_identity.py --certify).  A source change without a new counter session (or certificate) would silently drop those
    figures from the bench line: caught here, on the CPU."""
    import glob
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pattern, src in (("r*_sim_pmc.json", "avdm_similarity.hip"), ("r*_sgm_pmc.json", "avdm_sgm.hip")):
        newest = sorted(glob.glob(os.path.join(root, "profiles", pattern)))[-1]
        rec = json.load(open(newest))
        sha = hashlib.sha256(open(os.path.join(root, "alicevision_amd", "csrc", src), "rb").read()).hexdigest()
        ok = [rec.get("kernel_source_sha256")] + [e.get("sha256") for e in rec.get("isa_identical_sources", [])]
        assert sha in ok, "%s does not describe csrc/%s as it is now: re-measure, or scripts/isa_identity.py <measured rev> --certify" % (os.path.basename(newest), src)
        for e in rec.get("isa_identical_sources", []):
            assert os.path.exists(os.path.join(root, e["evidence"])), e


def test_refine_scratch_bytes_is_the_library_s_own_capacity_formula():
    """avdm_refine_similarity_scratch_bytes (what the scheduler prices a tile slot with, host/DepthMapEstimator.cpp::getNbSimultaneousTiles): the
    outlier list of a Refine sweep = a header + 8 bytes per unit, room for EVERY unit the sweep can append (two per (pixel, 8-plane chunk) pair:
    a full list cannot happen), at least 4096 units — a host
    function, callable without a GPU (one formula, in the library)"""
    lib = abi.load()
    f = lib.avdm_refine_similarity_scratch_bytes
    assert f(0, 31) == 0 and f(1000, 0) == 0
    assert f(1024 * 1024, 31) == 8 + (1024 * 1024 * 4 * 2) * 8       # a 1024 x 1024 tile, 31 planes = 4 chunks, every unit the sweep can append: 64 MB
    assert f(4000 * 3000, 31) == 8 + (4000 * 3000 * 4 * 2) * 8       # an undivided 12 MP frame: 768 MB
    assert f(100, 31) == 8 + 4096 * 8                                # the floor
    assert f(1024 * 1024, 32) == f(1024 * 1024, 31) < f(1024 * 1024, 33)
