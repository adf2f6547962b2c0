"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP stage against the CPU oracle through the C ABI.

Parity classes:
  bit-exact   : image pyramid (round 6), SGM path aggregation, WTA depth retrieval, thickness smoothing, SGM upscale, Refine sub-sample
                arg-min, volume init / update, and — in the product's reference-arithmetic mode — both similarity volumes: compared with ==;
  tolerance   : the DEFAULT similarity volumes (fast intrinsics, homogeneous patch projection, shifted sums), the default Refine volume,
                colour optimisation — tolerances stated in each test.
"""
import ctypes as C
import os

import numpy as np
import pytest

from alicevision_amd import abi

from alicevision_amd.synthetic import make_scene, plane_depths

from common import level_mismatch, make_hip_from_oracle, make_oracle, small_case

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _st():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module")
def case():
    """3 views 256x192, 32 planes: oracle run once, reused by the stage tests."""
    from oracle import oracle
    sc, sgm, ref, depths = small_case()
    o = make_oracle(sc, sgm, ref)
    with oracle.well_posed():  # well-conditioned NCC sums + exact R pixel (see test_similarity_volume_parity)
        o.run_sgm(0, [1, 2], depths)
        o.run_refine(0, [1, 2])
    return sc, sgm, ref, depths, o


def test_library_loaded_is_native():
    lib = abi.load()
    assert lib.avdm_device_count() >= 1
    buf = C.create_string_buffer(1024)
    abi.check(lib.avdm_device_info(0, buf, 1024))
    assert b"gfx950" in buf.value, buf.value


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
def test_pyramid_parity(mode):
    torch = _torch()
    from alicevision_amd.pipeline import DevicePyramid
    from oracle import oracle
    sc, sgm, ref, _ = small_case(width=250, height=186)  # non power-of-two sizes: floor-halved levels
    img = sc.images[0]
    hp = oracle.HostPyramid(img.numpy(), 1, 128, mode)
    dp = DevicePyramid(img.cuda(), 1, 128, mode)
    torch.cuda.synchronize()
    assert dp.desc.levels == hp.desc.levels
    for l in range(hp.desc.levels):
        a = hp.level(l).astype(np.float32)
        b = dp.level(l).cpu().numpy().astype(np.float32)
        assert a.shape == b.shape
        # BIT-EXACT since round 6 (rounds 1-5: one fp16 quantum on 1-5 % of the texels): the cube root of xyz2lab is the pinned build's C library
        # restated for the device (csrc/avdm_libm.h, held to glibc on the CPU by tests/test_libm.py), avdm_image.hip is compiled without FMA
        # contraction, and the wrapped taps of the level kernel keep a zero blend fraction like the reference's (session r06_a)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (l, float(np.abs(a - b).max()), float((a != b).mean()))


@pytest.mark.parametrize("w,h,s", [(64, 48, 2), (67, 45, 2), (101, 77, 3), (256, 192, 4), (33, 21, 1)])
def test_image_resize_bit_exact(w, h, s):
    """--downscale resize (imageAlgo::resizeImage -> OpenImageIO's default lanczos3): the device kernel against the oracle's restatement,
    even and odd sizes (non-integral ratios: per-column tap tables), clamp addressing at the borders; identical floats"""
    torch = _torch()
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    rng = np.random.default_rng(w * 1000 + h)
    src = rng.random((h, w, 4), dtype=np.float32)
    src[..., 3] = (rng.random((h, w)) > 0.1).astype(np.float32)
    dw, dh = w // s, h // s
    want = np.zeros((dh, dw, 4), np.float32)
    assert olib.avo_image_resize(oracle.ptr(want), dw * 16, dw, dh, oracle.ptr(src), w * 16, w, h, 4) == 0
    tsrc = torch.from_numpy(src).cuda()
    tdst = torch.full((dh, dw, 4), -1.0, dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_image_resize(_ptr(tdst), dw * 16, dw, dh, _ptr(tsrc), w * 16, w, h, _st()))
    torch.cuda.synchronize()
    got = tdst.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())
    # a constant image stays constant to rounding, and enlarging is refused
    one = torch.full((h, w, 4), 0.75, dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_image_resize(_ptr(tdst), dw * 16, dw, dh, _ptr(one), w * 16, w, h, _st()))
    torch.cuda.synchronize()
    assert float((tdst - 0.75).abs().max()) < 4e-6  # 13 x 13 normalised taps summed in fp32
    assert lib.avdm_image_resize(_ptr(one), w * 16, w, h, _ptr(tdst), dw * 16, dw, dh, _st()) != 0 or s == 1


@pytest.mark.parametrize("bits", [8, 16])
@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_image_decode_integer_bit_exact(bits, ch):
    """avdm_image_decode_integer (integer file samples -> linear float RGBA on the device: image::readImage(..., LINEAR) as
    mvsUtils::loadImage receives it) against the oracle: every sample value, every channel layout, pitched rows, with and without the sRGB
    decoding — identical floats (the transfer curve is a host-evaluated table)"""
    torch = _torch()
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    n = 1 << bits
    dt = np.uint8 if bits == 8 else np.uint16
    rng = np.random.default_rng(bits * 10 + ch)
    H, W = 9, n
    src = rng.integers(0, n, size=(H, W, ch)).astype(dt)
    src[0, :, 0] = np.arange(n).astype(dt)
    bpp = ch * (bits // 8)
    pad = 24  # bytes of row padding on the device side
    dev = np.zeros((H, W * bpp + pad), np.uint8)
    dev[:, :W * bpp] = src.reshape(H, -1).view(np.uint8)
    tsrc = torch.from_numpy(dev).cuda()
    for srgb in (1, 0):
        want = np.zeros((H, W, 4), np.float32)
        assert olib.avo_image_decode_integer(oracle.ptr(want), W * 16, oracle.ptr(src), W * bpp, W, H, ch, bits, srgb) == 0
        tdst = torch.full((H, W + 3, 4), -1.0, dtype=torch.float32, device="cuda")
        abi.check(lib.avdm_image_decode_integer(_ptr(tdst), (W + 3) * 16, _ptr(tsrc), W * bpp + pad, W, H, ch, bits, srgb, _st()))
        torch.cuda.synchronize()
        got = tdst.cpu().numpy()
        assert np.array_equal(got[:, :W].view(np.uint32), want.view(np.uint32)), float(np.abs(got[:, :W] - want).max())
        assert np.all(got[:, W:] == -1.0)
    assert lib.avdm_image_decode_integer(_ptr(tdst), (W + 3) * 16, _ptr(tsrc), W * bpp + pad, W, H, 5, bits, 1, _st()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["rgba_f32_headers", "rgb_f16_odd", "y_u32", "mixed_unaligned"])
def test_image_decode_exr_lines_bit_exact(layout):
    """avdm_image_decode_exr_lines (OpenEXR scan lines as stored -> linear float RGBA on the device: what image::readImage hands
    mvsUtils::loadImage for an .exr, mvsUtils/fileIO.cpp:386-446) against the oracle's restatement: FLOAT lines with the 8-byte chunk headers of
    an uncompressed file, HALF lines of odd width (alternate lines not 4-aligned), a UINT Y-only image, mixed types at odd offsets (the
    byte-by-byte path) — identical bits (FLOAT NaN payloads, subnormals, infinities included); the padding of the destination rows untouched"""
    import ctypes as C
    torch = _torch()
    from oracle import oracle
    lib = abi.load()
    rng = np.random.default_rng(11)
    H, W = 19, 131
    lead = 0
    if layout == "rgba_f32_headers":
        names, types = "ABGR", [2, 2, 2, 2]
        stride_extra = 8
    elif layout == "rgb_f16_odd":
        names, types = "BGR", [1, 1, 1]
        stride_extra = 0
    elif layout == "y_u32":
        names, types = "Y", [0]
        stride_extra = 8
    else:
        names, types = "ABGR", [1, 2, 0, 1]
        stride_extra, lead = 3, 1
    size = {0: 4, 1: 2, 2: 4}
    offs, o = {}, 0
    for n, t in zip(names, types):
        offs[n] = (o, t)
        o += W * size[t]
    bpl = o
    stride = bpl + stride_extra
    raw = rng.integers(0, 256, size=lead + H * stride, dtype=np.uint8)  # every bit pattern: NaNs, infinities, subnormals
    for n, (o, t) in offs.items():  # HALF NaNs become infinities (a signalling NaN is quieted by the hardware conversion, not by numpy's)
        if t == 1:
            for y in range(H):
                b0 = lead + y * stride + o
                hi = raw[b0 + 1:b0 + 2 * W:2]
                lo = raw[b0:b0 + 2 * W:2]
                nan = ((hi & 0x7c) == 0x7c) & (((hi & 0x03) != 0) | (lo != 0))
                hi[nan] &= 0xfc
                lo[nan] = 0
    pick = (lambda n: offs[n] if n in offs else offs["Y"])
    off = [pick("R")[0], pick("G")[0], pick("B")[0], offs["A"][0] if "A" in offs else -1]
    typ = [pick("R")[1], pick("G")[1], pick("B")[1], offs["A"][1] if "A" in offs else 2]
    want = oracle.exr_lines_to_rgba(raw[lead:].tobytes(), stride, W, H, off, typ)
    tsrc = torch.from_numpy(raw).cuda()
    tdst = torch.full((H, W + 3, 4), -1.0, dtype=torch.float32, device="cuda")
    coff, ctyp = (C.c_longlong * 4)(*off), (C.c_int * 4)(*typ)
    abi.check(lib.avdm_image_decode_exr_lines(_ptr(tdst), (W + 3) * 16, tsrc.data_ptr() + lead, stride, W, H, coff, ctyp, _st()))
    torch.cuda.synchronize()
    got = tdst.cpu().numpy()
    assert np.array_equal(got[:, :W].view(np.uint32), want.view(np.uint32))
    assert np.all(got[:, W:] == -1.0)
    bad = (C.c_int * 4)(3, 2, 2, 2)
    assert lib.avdm_image_decode_exr_lines(_ptr(tdst), (W + 3) * 16, tsrc.data_ptr() + lead, stride, W, H, coff, bad, _st()) != 0
    assert lib.avdm_image_decode_exr_lines(_ptr(tdst), (W + 3) * 16, tsrc.data_ptr() + lead, bpl - 4, W, H, coff, ctyp, _st()) != 0


def _jpeg_device_and_oracle(path):
    import ctypes as C
    torch = _torch()
    from alicevision_amd import jpeg_io
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    j = jpeg_io.read_coefficients(path)
    nc = len(j.components)
    host = j.descriptors([c["coef"].ctypes.data for c in j.components])
    want = np.zeros((j.height, j.width, 3), np.uint8)
    olib.avo_image_decode_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(abi.JpegComponent), C.c_int, C.c_int, C.c_int, C.c_int]
    assert olib.avo_image_decode_jpeg(want.ctypes.data, 3 * j.width, j.width, j.height, host, nc, j.hmax, j.vmax, 0 if j.stored_as_rgb else 1) == 0
    tcoef = [torch.from_numpy(c["coef"]).cuda() for c in j.components]
    dev = j.descriptors([t.data_ptr() for t in tcoef])
    scratch = torch.empty(int(lib.avdm_image_decode_jpeg_scratch_bytes(dev, nc)), dtype=torch.uint8, device="cuda")
    pitch = 3 * j.width + 5  # an unaligned pitch: the byte-wise store path
    for pitch in (3 * j.width + 5, (3 * j.width + 3) // 4 * 4 + 8):
        tdst = torch.full((j.height, pitch), 77, dtype=torch.uint8, device="cuda")
        abi.check(lib.avdm_image_decode_jpeg(_ptr(tdst), pitch, j.width, j.height, dev, nc, j.hmax, j.vmax, 0 if j.stored_as_rgb else 1, _ptr(scratch), _st()))
        torch.cuda.synchronize()
        got = tdst.cpu().numpy()
        assert np.array_equal(got[:, :3 * j.width].reshape(j.height, j.width, 3), want), (path, pitch)
        assert np.all(got[:, 3 * j.width:] == 77)
    return want, j


def test_jpeg_decode_bit_exact_on_the_golden_files():
    """avdm_image_decode_jpeg (dequantisation, libjpeg's integer inverse DCT, fancy up-sampling, YCbCr -> RGB on the device) == the oracle ==
    the pixels libjpeg-turbo decoded, for every golden file (all sampling modes, baseline / progressive, grey, RGB-stored, 1 x 1 ... )"""
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
    exp = np.load(os.path.join(golden, "expected.npz"))
    for name in exp.files:
        want, j = _jpeg_device_and_oracle(os.path.join(golden, name + ".jpg"))
        assert np.array_equal(want, exp[name]), name
    lib = abi.load()
    bad = (abi.JpegComponent * 3)()
    assert lib.avdm_image_decode_jpeg(None, 0, 8, 8, bad, 2, 1, 1, 1, None, _st()) != 0  # two components


def test_jpeg_decode_full_size(tmp_path):
    """a 12 MP 4:2:0 photograph-sized file (and 4:2:2 progressive at 3000 x 2000): device == oracle == libjpeg-turbo, and the decoded image
    goes through avdm_image_decode_integer to the linear float RGBA the pyramids are built from"""
    Image = pytest.importorskip("PIL.Image")
    torch = _torch()
    rng = np.random.default_rng(12)
    for (w, h, sub, prog) in ((4000, 3000, 2, False), (3001, 1999, 1, True)):
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        a = np.stack([128 + 100 * np.sin(x / 37.0) * np.cos(y / 25.0), 128 + 90 * np.cos(x / 13.0 + y / 19.0), 60 + x * (150.0 / w) + 20 * np.sin(y / 7.0)], -1)
        a = np.clip(a + rng.normal(0, 5, a.shape).astype(np.float32), 0, 255).astype(np.uint8)
        p = str(tmp_path / "big.jpg")
        Image.fromarray(a).save(p, quality=92, subsampling=sub, progressive=prog)
        want, j = _jpeg_device_and_oracle(p)
        assert np.array_equal(want, np.array(Image.open(p)))
    # RGB8 -> linear float RGBA (the existing path of 8-bit input)
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    lin = np.zeros((h, w, 4), np.float32)
    assert olib.avo_image_decode_integer(oracle.ptr(lin), w * 16, oracle.ptr(want), w * 3, w, h, 3, 8, 1) == 0
    trgb = torch.from_numpy(want).cuda()
    tlin = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_image_decode_integer(_ptr(tlin), w * 16, _ptr(trgb), w * 3, w, h, 3, 8, 1, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(tlin.cpu().numpy().view(np.uint32), lin.view(np.uint32))


@pytest.mark.parametrize("model,k", [(0, (0.0, 0.0, 0.0)), (1, (0.08, 0.0, 0.0)), (2, (0.1, -0.05, 0.01)), (2, (-0.3, 0.1, 0.0)), (3, (0.05, 0.02, -0.01))])
def test_image_undistort_bit_exact(model, k):
    """camera::UndistortImage (PrepareDenseScene): the device kernel against the oracle — double-precision geometry, float sample position,
    double accumulation in the sampler: identical floats, fill colour where the distorted position leaves the image, pitched rows"""
    torch = _torch()
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    rng = np.random.default_rng(11 + model)
    H, W = 123, 187
    src = rng.random((H, W, 4), dtype=np.float32)
    cam = abi.Intrinsic(width=W, height=H, scale_x=150.0, scale_y=148.5, offset_x=3.25, offset_y=-2.5, distortion_model=model, k=(C.c_double * 3)(*k))
    fill = (C.c_float * 4)(0.1, 0.2, 0.3, 0.0)
    want = np.zeros_like(src)
    assert olib.avo_image_undistort(oracle.ptr(want), W * 16, oracle.ptr(src), W * 16, C.byref(cam), C.byref(fill)) == 0
    pitch = (W + 5) * 16
    tsrc = torch.zeros((H, W + 5, 4), dtype=torch.float32, device="cuda")
    tsrc[:, :W] = torch.from_numpy(src).cuda()
    tdst = torch.full((H, W + 5, 4), -1.0, dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_image_undistort(_ptr(tdst), pitch, _ptr(tsrc), pitch, C.byref(cam), C.byref(fill), _st()))
    torch.cuda.synchronize()
    got = tdst.cpu().numpy()
    assert np.array_equal(got[:, :W].view(np.uint32), want.view(np.uint32)), float(np.abs(got[:, :W] - want).max())
    assert np.all(got[:, W:] == -1.0)  # padding untouched


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
def test_similarity_volume_parity(mode):
    """Weighted NCC is ill-conditioned in fp32 the way the reference accumulates it (DESIGN.md "NCC conditioning"): the
    HIP kernel (shifted sums) is compared with the oracle evaluating the same formula with double-precision sums, and the
    fp32-faithful oracle's own distance to that value is the noise floor the HIP-vs-fp32-oracle distance is held against."""
    torch = _torch()
    from oracle import oracle
    sc, sgm, ref, depths = small_case()
    Z = len(depths)
    rng_t = [(0, 32), (3, 29)]
    o = make_oracle(sc, sgm, ref, filter_mode=mode)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False)
        best64, second64 = o.best_raw[..., :Z].copy(), o.second[..., :Z].copy()
    o.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False)
    second32 = o.second[..., :Z].copy()
    h = make_hip_from_oracle(o, sc, sgm, ref)
    h.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    second_h = h.second.cpu().numpy()[..., :Z]
    best_h = h.best_raw.cpu().numpy()[..., :Z]
    # knife-edge rows / columns of the LITERAL restatement: the border test `rp < wsh + 2` (Patch.cuh:490-493) is decided by fp32
    # rounding noise where the pixel coordinate equals wsh + 2 exactly (stage pixel 3 at stepXY 2, wsh 4) — excluded from (2)
    inner = (slice(4, None), slice(4, None))
    second32_in, second64_in, second_h_in = second32[inner], second64[inner], second_h[inner]

    # (1) against the well-conditioned evaluation: uint8 truncation boundaries + (FIXED8) 1/256 weight-bucket flips only
    for got, want in ((best_h, best64), (second_h, second64)):
        frac, mx = level_mismatch(want, got)
        d = np.abs(want.astype(np.int16) - got.astype(np.int16))
        assert frac <= (0.03 if mode == abi.FILTER_CUDA_FIXED8 else 0.01), (frac, mx)
        assert (d > 1).mean() <= 2e-3, (d > 1).mean()
        assert ((want == 255) != (got == 255)).mean() <= 2e-3  # validity masks
    # (2) against the fp32-faithful restatement: no further than that restatement is from the exact value of its own formula
    floor, _ = level_mismatch(second64_in, second32_in)
    frac32, _ = level_mismatch(second32_in, second_h_in)
    assert floor > 0.05, floor  # the conditioning problem is real (otherwise tighten this test)
    assert frac32 <= 1.25 * floor + 0.03, (frac32, floor)


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
@pytest.mark.parametrize("wsh", [4, 2])
def test_reference_arithmetic_sgm_volume_equals_the_oracle_bit_for_bit(mode, wsh):
    """avdm_sgm_params_t::referenceArithmetic (strict_sgm_kernel, csrc/avdm_literal.hip): the reference's arithmetic as written, taps from LDS
    windows used as a cache, expf to the bits of the pinned build's C library — the best / second-best volumes over two T cameras with ragged
    plane ranges equal the oracle's literal evaluation (= the reference's own kernels compiled for the CPU, tests/test_oracle_ref.py) BYTE FOR BYTE,
    both filter modes, the default patch (unrolled sample loop) and another one, and the windows change no bit (AVDM_STRICT_WINDOWS=0: every tap
    from global memory)."""
    torch = _torch()
    sc, sgm, ref, depths = small_case()
    sgm.wsh = wsh
    Z = len(depths)
    rng_t = [(0, 32), (3, 29)]
    o = make_oracle(sc, sgm, ref, filter_mode=mode)
    o.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False)
    want_best, want_second = o.best_raw[..., :Z].copy(), o.second[..., :Z].copy()
    sgm_s = abi.SgmParams.default(wsh=wsh, referenceArithmetic=1)
    got = {}
    for win in ("1", "0"):
        os.environ["AVDM_STRICT_WINDOWS"] = win
        try:
            h = make_hip_from_oracle(o, sc, sgm_s, ref)
            h.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False, keep_raw=True)
            torch.cuda.synchronize()
            got[win] = (h.best_raw.cpu().numpy()[..., :Z].copy(), h.second.cpu().numpy()[..., :Z].copy())
        finally:
            os.environ.pop("AVDM_STRICT_WINDOWS", None)
    for win in ("1", "0"):
        assert np.array_equal(got[win][0], want_best), (win, float((got[win][0] != want_best).mean()))
        assert np.array_equal(got[win][1], want_second), (win, float((got[win][1] != want_second).mean()))
    assert (want_second != 255).mean() > 0.5  # (the comparison is not one of empty volumes)


def test_reference_arithmetic_refine_volume_equals_the_oracle_bit_for_bit(case):
    """avdm_refine_params_t::referenceArithmetic (strict_refine_kernel): the fp16 Refine volume — the sum over two T cameras of the sigmoid-filtered
    similarities, accumulated in the reference's order — equals the literal oracle's on the same SGM map, every half; with and without the windows"""
    torch = _torch()
    sc, sgm, ref, depths, o = case
    o2 = make_oracle(sc, sgm, ref)
    o2.sgm_depth_thickness = o.sgm_depth_thickness.copy()
    o2.run_refine(0, [1, 2], optimize_enabled=False)
    Zr = ref.halfNbDepths * 2 + 1
    want = o2.refine_volume[..., :Zr]
    ref_s = abi.RefineParams.default(referenceArithmetic=1)
    for win in ("1", "0"):
        os.environ["AVDM_STRICT_WINDOWS"] = win
        try:
            h = make_hip_from_oracle(o, sc, sgm, ref_s)
            h._alloc(len(depths))
            h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
            h.run_refine(0, [1, 2], optimize_enabled=False)
            torch.cuda.synchronize()
            got = h.refine_volume.cpu().numpy()[..., :Zr]
        finally:
            os.environ.pop("AVDM_STRICT_WINDOWS", None)
        assert np.array_equal(got.view(np.uint16), np.ascontiguousarray(want).view(np.uint16)), (win, float((got != want).mean()), float(np.abs(got.astype(np.float32) - want.astype(np.float32)).max()))
    assert (want != 0).mean() > 0.5


@pytest.mark.parametrize("X,Y,Z", [(37, 29, 20), (64, 48, 33), (5, 3, 256)])
def test_volume_init_update_add_bit_exact(X, Y, Z):
    """row a4 of SURVEY section 8 on its own: cuda_volumeInitialize (uint8 / fp16), cuda_volumeUpdateUninitializedSimilarity and cuda_volumeAdd
    (deviceSimilarityVolume.cu:25-153, kernels.cuh:48-107) on pitched z-fastest volumes with padded planes — identical bytes, padding
    planes and rows outside the volume untouched"""
    torch = _torch()
    from oracle import oracle
    lib, olib = abi.load(), oracle.load()
    rng = np.random.default_rng(X * Z)
    Zp = (Z + 3) // 4 * 4 + 4       # pitch_x with spare planes
    Xp = X + 3                      # pitch_y with spare pixels
    py, pxx = Xp * Zp, Zp
    # uint8 initialise
    host = rng.integers(0, 255, size=(Y + 2, Xp, Zp), dtype=np.uint8)
    want = host.copy()
    olib.avo_volume_initialize_u8(oracle.ptr(want), py, pxx, X, Y, Z, 255)
    t = torch.from_numpy(host.copy()).cuda()
    abi.check(lib.avdm_volume_initialize_u8(_ptr(t), py, pxx, X, Y, Z, 255, _st()))
    torch.cuda.synchronize()
    got = t.cpu().numpy()
    # (the device writes whole dwords: the planes up to the next multiple of four belong to the volume's padding and may be written)
    Z4 = (Z + 3) // 4 * 4
    assert np.array_equal(got[:Y, :X, :Z], want[:Y, :X, :Z]) and np.all(got[:Y, :X, :Z] == 255)
    assert np.array_equal(got[Y:], host[Y:]) and np.array_equal(got[:, X:], host[:, X:]) and np.array_equal(got[:, :, Z4:], host[:, :, Z4:])
    # update uninitialised: second >= 255 takes best
    best = rng.integers(0, 256, size=(Y, Xp, Zp), dtype=np.uint8)
    second = rng.integers(200, 256, size=(Y, Xp, Zp), dtype=np.uint8)
    want2 = second.copy()
    olib.avo_volume_update_uninitialized(oracle.ptr(best), oracle.ptr(want2), py, pxx, X, Y, Z)
    tb, ts = torch.from_numpy(best).cuda(), torch.from_numpy(second.copy()).cuda()
    abi.check(lib.avdm_volume_update_uninitialized(_ptr(tb), _ptr(ts), py, pxx, X, Y, Z, _st()))
    torch.cuda.synchronize()
    g2 = ts.cpu().numpy()
    assert np.array_equal(g2[:, :X, :Z], want2[:, :X, :Z]) and (want2 != second).any()
    assert np.array_equal(g2[:, X:], second[:, X:]) and np.array_equal(g2[:, :, Z4:], second[:, :, Z4:])
    # fp16 initialise + add
    Zh = (Z + 7) // 8 * 8
    pyh, pxh = Xp * Zh * 2, Zh * 2
    a = rng.standard_normal((Y, Xp, Zh)).astype(np.float16)
    b = rng.standard_normal((Y, Xp, Zh)).astype(np.float16)
    want3 = a.copy()
    olib.avo_volume_initialize_f16(oracle.ptr(want3), pyh, pxh, X, Y, Z, 0.25)
    ta = torch.from_numpy(a.copy()).cuda()
    abi.check(lib.avdm_volume_initialize_f16(_ptr(ta), pyh, pxh, X, Y, Z, 0.25, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(ta.cpu().numpy().view(np.uint16)[:, :X, :Z], want3.view(np.uint16)[:, :X, :Z])
    assert np.array_equal(ta.cpu().numpy().view(np.uint16)[:, X:], a.view(np.uint16)[:, X:])
    want4 = a.copy()
    olib.avo_volume_add_f16(oracle.ptr(want4), oracle.ptr(b), pyh, pxh, X, Y, Z)
    ta, tb2 = torch.from_numpy(a.copy()).cuda(), torch.from_numpy(b).cuda()
    abi.check(lib.avdm_volume_add_f16(_ptr(ta), _ptr(tb2), pyh, pxh, X, Y, Z, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(ta.cpu().numpy().view(np.uint16)[:, :X, :Z], want4.view(np.uint16)[:, :X, :Z])


def test_sgm_aggregation_bit_exact(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    Y, X, Zp = o.second.shape
    Z = len(depths)
    roi = o.droi(sgm.scale * sgm.stepXY)
    vin = torch.from_numpy(o.second).cuda()
    vout = torch.full_like(vin, 7)
    pyr = make_hip_from_oracle(o, sc, sgm, ref).pyr[0]
    from alicevision_amd.pipeline import optimize_scratch
    scratch = optimize_scratch(lib, X, Y, Z)
    abi.check(lib.avdm_volume_optimize(_ptr(vout), _ptr(vin), X * Zp, Zp, _ptr(scratch), C.byref(pyr.desc), C.byref(sgm), Z, roi, _st()))
    torch.cuda.synchronize()
    got = vout.cpu().numpy()[..., :Z]
    want = o.filtered[..., :Z]
    assert np.array_equal(got, want), level_mismatch(got, want)


@pytest.mark.parametrize("Z,axes,p2,p1", [(5, b"YX", 100.0, 10.0), (67, b"Y", 100.0, 10.0), (130, b"X", -40.0, 10.0), (256, b"YX", 100.0, 10.0),
                                          (300, b"XY", 100.0, 10.0), (512, b"YX", 100.0, 10.0), (1027, b"XY", 100.0, 10.0),
                                          (64, b"YX", -99.99995, 10.0),   # frac(P2) ~ 1: every step takes the fp32 step of the packed kernel
                                          (256, b"XY", -120.9999, 3.0),   # idem on the full-dword kernel
                                          (41, b"YX", 100.0, 10.5),       # non-integer P1: fp32 kernel
                                          (256, b"YX", 100.0, 0.25)])
def test_sgm_aggregation_shapes(Z, axes, p2, p1):
    """ragged depth counts (tail bytes, several dwords per lane), single axes, fixed P2, non-square ROI with an offset; the packed
    uint16 kernel, its fp32 fallback step and the pure fp32 kernel must all be bit-exact"""
    torch = _torch()
    from oracle import oracle
    rng = np.random.RandomState(Z)
    sc, sgm, ref, _ = small_case(width=128, height=96, filteringAxes=axes, p2Weighting=p2, p1=p1)
    o = make_oracle(sc, sgm, ref)
    lib, olib = abi.load(), oracle.load()
    X, Y = 23, 17
    Zp = (Z + 3) // 4 * 4
    roi = abi.ROI.make(3, 3 + X, 5, 5 + Y)
    vin = rng.randint(0, 256, size=(Y, X, Zp)).astype(np.uint8)
    want = np.full_like(vin, 9)
    olib.avo_volume_optimize(oracle.ptr(want), oracle.ptr(vin), X * Zp, Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
    pyr = make_hip_from_oracle(o, sc, sgm, ref).pyr[0]
    tin = torch.from_numpy(vin).cuda()
    tout = torch.full_like(tin, 9)
    from alicevision_amd.pipeline import optimize_scratch
    scratch = optimize_scratch(lib, X, Y, Z)
    abi.check(lib.avdm_volume_optimize(_ptr(tout), _ptr(tin), X * Zp, Zp, _ptr(scratch), C.byref(pyr.desc), C.byref(sgm), Z, roi, _st()))
    torch.cuda.synchronize()
    got = tout.cpu().numpy()
    assert np.array_equal(got, want), level_mismatch(got, want)  # including the untouched padding planes z >= Z


def test_sgm_aggregation_tiles_batch():
    """several tiles (different sizes, ROI offsets and depth counts, two R images) in ONE batched call == one call per tile"""
    torch = _torch()
    from alicevision_amd.pipeline import optimize_scratch
    rng = np.random.RandomState(11)
    sc, sgm, ref, _ = small_case(width=160, height=128)
    o = make_oracle(sc, sgm, ref)
    pyrs = make_hip_from_oracle(o, sc, sgm, ref).pyr
    lib = abi.load()
    specs = [(23, 17, 64, 3, 5, 0), (31, 9, 64, 0, 0, 1), (8, 40, 256, 7, 2, 0), (16, 16, 256, 1, 1, 1), (12, 5, 37, 2, 9, 0)]
    vin, single, tiles = [], [], (abi.SgmTile * len(specs))()
    batch_out, keep = [], []
    total = 0
    for i, (X, Y, Z, x0, y0, cam) in enumerate(specs):
        Zp = (Z + 3) // 4 * 4
        v = torch.from_numpy(rng.randint(0, 256, size=(Y, X, Zp)).astype(np.uint8)).cuda()
        roi = abi.ROI.make(x0, x0 + X, y0, y0 + Y)
        out1 = torch.full_like(v, 9)
        sc1 = optimize_scratch(lib, X, Y, Z)
        abi.check(lib.avdm_volume_optimize(_ptr(out1), _ptr(v), X * Zp, Zp, _ptr(sc1), C.byref(pyrs[cam].desc), C.byref(sgm), Z, roi, _st()))
        single.append(out1)
        out2 = torch.full_like(v, 9)
        batch_out.append(out2)
        vin.append(v)
        tiles[i] = abi.SgmTile(out2.data_ptr(), v.data_ptr(), X * Zp, Zp, Z, roi, C.pointer(pyrs[cam].desc))
        total += int(lib.avdm_volume_optimize_scratch_bytes(X, Y, Z))
    scratch = torch.empty(total, dtype=torch.uint8, device="cuda")
    abi.check(lib.avdm_volume_optimize_tiles(len(specs), tiles, _ptr(scratch), C.byref(sgm), _st()))
    torch.cuda.synchronize()
    for a, b in zip(single, batch_out):
        assert torch.equal(a, b)


@pytest.fixture(scope="module")
def big_pyramid():
    """one 1760x1320 R image (SGM stage 440x330 at scale 2 x stepXY 2): real colour steps for the adaptive P2 at the sizes below"""
    sc, sgm, ref, _ = small_case(width=1760, height=1320, n_views=1, seed=5)
    o = make_oracle(sc, sgm, ref)
    return sc, o


def _structured_volume(rng, Y, X, Zp):
    """uint8 cost volume with a smooth minimum valley + noise + invalid (255) voxels: exercises every branch of the recurrence
    (a uniformly random volume saturates at 255 after a few steps and hides errors of the steady state)"""
    yy, xx, zz = np.meshgrid(np.arange(Y), np.arange(X), np.arange(Zp), indexing="ij")
    valley = Zp * (0.5 + 0.3 * np.sin(xx / 37.0) * np.cos(yy / 23.0))
    v = np.minimum(np.abs(zz - valley) * 3.0 + rng.randint(0, 40, size=(Y, X, Zp)), 254).astype(np.uint8)
    v[rng.rand(Y, X, Zp) < 0.02] = 255
    v[rng.rand(Y, X) < 0.03] = 255  # whole invalid pixels
    return v


@pytest.mark.parametrize("X,Y,Z,axes,x0,y0,quirk,p2w", [(300, 401, 256, b"YX", 0, 0, 0, 100.0), (401, 300, 256, b"XY", 0, 0, 0, 12.0),
                                                      (401, 300, 300, b"YX", 0, 0, 0, 12.0), (300, 401, 300, b"XY", 0, 0, 0, 100.0),
                                                      (401, 300, 256, b"YX", 21, 9, 0, 30.0), (300, 311, 256, b"YX", 33, 5, 1, 12.0),
                                                      (401, 300, 300, b"YX", 7, 19, 1, 100.0)])
def test_sgm_aggregation_at_scale_bit_exact(big_pyramid, X, Y, Z, axes, x0, y0, quirk, p2w):
    """the register ring's steady state (whole spans of 4 slots x 8 slices, B >= 300) and the later 64-step chunks of the P2 map, both
    axis orders, ROI offsets (also with the reference's begin-x / begin-y swap, kernels.cuh:688-709), adaptive P2 from a real pyramid:
    HIP == avo_volume_optimize byte for byte.  Follows deviceSimilarityVolume.cu:262-425."""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.pipeline import DevicePyramid, optimize_scratch
    sc, o = big_pyramid
    sgm = abi.SgmParams.default(filteringAxes=axes, strictRoiQuirk=quirk, p2Weighting=p2w)
    lib, olib = abi.load(), oracle.load()
    rng = np.random.RandomState(X * 7 + Z)
    Zp = (Z + 3) // 4 * 4
    roi = abi.ROI.make(x0, x0 + X, y0, y0 + Y)
    vin = _structured_volume(rng, Y, X, Zp)
    want = np.full_like(vin, 9)
    olib.avo_volume_optimize(oracle.ptr(want), oracle.ptr(vin), X * Zp, Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
    pyr = DevicePyramid.from_host_bytes(o.pyr[0].desc, o.pyr[0].buf)
    tin = torch.from_numpy(vin).cuda()
    tout = torch.full_like(tin, 9)
    scratch = optimize_scratch(lib, X, Y, Z)
    abi.check(lib.avdm_volume_optimize(_ptr(tout), _ptr(tin), X * Zp, Zp, _ptr(scratch), C.byref(pyr.desc), C.byref(sgm), Z, roi, _st()))
    torch.cuda.synchronize()
    got = tout.cpu().numpy()
    assert np.array_equal(got, want), level_mismatch(got, want)
    # the adaptive P2 really varies here (otherwise this is the fixed-P2 test again)
    assert len(np.unique(want[2:-2, 2:-2, 1:Z - 1])) > 100


@pytest.mark.parametrize("BX,BY,X,Y,Z,axes,x0,y0", [(256, 256, 216, 204, 256, b"YX", 54, 51), (128, 96, 100, 71, 64, b"XY", 9, 30)])
def test_sgm_aggregation_over_buffer_extent_bit_exact(big_pyramid, BX, BY, X, Y, Z, axes, x0, y0):
    """the extent the REFERENCE aggregates over (last paragraph): the volume of the tile BUFFER (BX x BY), the tile's
    similarities (X x Y) in its corner and 255 everywhere else, ROI begin at the tile's offset.  The reverse paths cross the 255-filled
    remainder before they enter the tile.  HIP == avo_volume_optimize byte for byte over the whole buffer — what
    OracleDepthMap.run_sgm(tile_buffer=...) does on the CPU, which equals the reference's own Sgm.cpp (tests/test_oracle_ref.py)."""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.pipeline import DevicePyramid, optimize_scratch
    sc, o = big_pyramid
    sgm = abi.SgmParams.default(filteringAxes=axes, strictRoiQuirk=1)
    lib, olib = abi.load(), oracle.load()
    rng = np.random.RandomState(BX + Z)
    Zp = (Z + 3) // 4 * 4
    roi = abi.ROI.make(x0, x0 + BX, y0, y0 + BY)
    vin = np.full((BY, BX, Zp), 255, np.uint8)
    vin[:Y, :X] = _structured_volume(rng, Y, X, Zp)
    want = np.full_like(vin, 9)
    olib.avo_volume_optimize(oracle.ptr(want), oracle.ptr(vin), BX * Zp, Zp, BX, BY, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
    pyr = DevicePyramid.from_host_bytes(o.pyr[0].desc, o.pyr[0].buf)
    tin = torch.from_numpy(vin).cuda()
    tout = torch.full_like(tin, 9)
    scratch = optimize_scratch(lib, BX, BY, Z)
    abi.check(lib.avdm_volume_optimize(_ptr(tout), _ptr(tin), BX * Zp, Zp, _ptr(scratch), C.byref(pyr.desc), C.byref(sgm), Z, roi, _st()))
    torch.cuda.synchronize()
    got = tout.cpu().numpy()
    assert np.array_equal(got, want), level_mismatch(got, want)
    # and the extent matters: the same tile aggregated over its own ROI only gives other bytes inside the tile
    alone = np.full((Y, X, Zp), 9, np.uint8)
    tile_in = np.ascontiguousarray(vin[:Y, :X])
    olib.avo_volume_optimize(oracle.ptr(alone), oracle.ptr(tile_in), X * Zp, Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, abi.ROI.make(x0, x0 + X, y0, y0 + Y))
    assert (alone[..., :Z] != want[:Y, :X, :Z]).mean() > 0.01


@pytest.mark.parametrize("mode,axes,W,H,NP,roi,buf,tcr", [
    (abi.FILTER_CUDA_FIXED8, b"YX", 320, 240, 48, (80, 240, 40, 200), (192, 176), None),           # interior tile, buffer larger than the tile
    (abi.FILTER_EXACT, b"XY", 320, 240, 40, (128, 320, 96, 240), (256, 256), [(0, 40), (5, 33)]),  # tile at the far image corner, other axis order
    (abi.FILTER_CUDA_FIXED8, b"YX", 250, 186, 32, (60, 187, 0, 119), (160, 128), None),            # offset in x only, sizes not divisible by 4
])
def test_offset_tile_equals_oracle_over_buffer_extent(mode, axes, W, H, NP, roi, buf, tcr):
    """A tile that does not start at the image origin, laid out and aggregated over its tile BUFFER like the reference does
    (deviceSimilarityVolume.cu:278-283, Sgm.cpp:37-72) — the harness (pipeline.DepthMapTile(tile_buffer=...)) against
    OracleDepthMap.run_sgm / run_refine(tile_buffer=...), which tests/test_oracle_ref.py pins to the reference's own Sgm.cpp / Refine.cpp
    bit for bit.  Stage by stage on identical inputs, then end to end."""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.pipeline import optimize_scratch
    sc, sgm, ref, depths = small_case(width=W, height=H, n_planes=NP, seed=11, filteringAxes=axes)
    Z = len(depths)
    o = make_oracle(sc, sgm, ref, filter_mode=mode, roi=roi)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths, tc_ranges=tcr, tile_buffer=buf)
        want = o.run_refine(0, [1, 2], tile_buffer=buf)
    X, Y = o.second.shape[1], o.second.shape[0]
    AY, AX = o.second_buffer.shape[:2]
    assert (AX, AY) != (X, Y)
    # (a) sweep: the tile's similarities in the corner of the buffer volume (tolerance class), 255 in the remainder
    h = make_hip_from_oracle(o, sc, sgm, ref, roi=roi, tile_buffer=buf)
    assert h.sgm_extent() == (AX, AY)
    h.run_sgm(0, [1, 2], depths, tc_ranges=tcr, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    sec = h.second.cpu().numpy()
    frac, mx = level_mismatch(o.second[..., :Z], sec[:Y, :X, :Z])
    assert frac <= (0.03 if mode == abi.FILTER_CUDA_FIXED8 else 0.01), (frac, mx)
    rest = np.ones((AY, AX), bool)
    rest[:Y, :X] = False
    assert np.all(sec[rest][:, :Z] == 255) and np.all(h.best_raw.cpu().numpy()[rest][:, :Z] == 255)
    # (b) path aggregation over the buffer extent on IDENTICAL input bytes: bit-exact over the whole buffer, then WTA bit-exact
    h.second.copy_(torch.from_numpy(o.second_buffer))
    lib = abi.load()
    abi.check(lib.avdm_volume_optimize(_ptr(h.best), _ptr(h.second), AX * h.Zp, h.Zp, _ptr(h.sgm_scratch), C.byref(h.pyr[0].desc), C.byref(sgm), Z,
                                       h.sgm_extent_roi(), _st()))
    torch.cuda.synchronize()
    got = h.best.cpu().numpy()
    assert np.array_equal(got[..., :Z], o.filtered_buffer[..., :Z]), level_mismatch(got[..., :Z], o.filtered_buffer[..., :Z])
    dt, dsm = h.finish_sgm(0, Z)
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy(), o.sgm_depth_thickness)
    assert np.array_equal(dsm.cpu().numpy(), o.sgm_depth_sim)
    # ... and the extent is what makes it so: aggregated over its ROI alone the same tile gives other bytes
    alone = np.full((Y, X, h.Zp), 9, np.uint8)
    tile_in = np.ascontiguousarray(o.second_buffer[:Y, :X])
    oracle.load().avo_volume_optimize(oracle.ptr(alone), oracle.ptr(tile_in), X * h.Zp, h.Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, o.droi(sgm.scale * sgm.stepXY))
    assert (alone[..., :Z] != o.filtered[..., :Z]).mean() > 0.005
    # (c) end to end, everything on the GPU (from the same pyramids): BASELINE's bar
    h.run_sgm(0, [1, 2], depths, tc_ranges=tcr)
    out = h.run_refine(0, [1, 2]).cpu().numpy()
    both = (want[..., 0] > 0) & (out[..., 0] > 0)
    assert both.mean() > 0.5 and ((want[..., 0] > 0) != (out[..., 0] > 0)).mean() < 5e-3
    err = (out[..., 0] - want[..., 0])[both]
    rmse = np.sqrt(np.mean(np.sort(err ** 2)[: int(0.995 * err.size)]))
    untrimmed = float(np.sqrt(np.mean(err ** 2)))
    print("offset tile", roi, "rmse over the best 99.5 %%: %.3e, untrimmed: %.3e" % (rmse, untrimmed))
    assert rmse < 1e-3, (rmse, untrimmed)
    # untrimmed: these are 250-320-pixel images (a pixel is 1.3e-2 depth units, one plane step 1e-2): a handful of pixels whose winning plane
    # flips carry the excess — at 12 MP the same assertion holds at 1e-3 (test_parity_of_default_tiles_at_12mp)
    assert untrimmed < 3e-3, (rmse, untrimmed)


def test_sgm_aggregation_tiles_batch_at_scale_bit_exact(big_pyramid):
    """avdm_volume_optimize_tiles with three large tiles (different sizes / offsets / depth counts) in one launch per axis == the oracle
    tile by tile"""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.pipeline import DevicePyramid
    sc, o = big_pyramid
    sgm = abi.SgmParams.default()
    lib, olib = abi.load(), oracle.load()
    rng = np.random.RandomState(3)
    pyr = DevicePyramid.from_host_bytes(o.pyr[0].desc, o.pyr[0].buf)
    specs = [(401, 300, 256, 0, 0), (300, 320, 300, 130, 7), (333, 301, 256, 40, 25)]
    tiles = (abi.SgmTile * len(specs))()
    wants, outs, keep, total = [], [], [], 0
    for i, (X, Y, Z, x0, y0) in enumerate(specs):
        Zp = (Z + 3) // 4 * 4
        roi = abi.ROI.make(x0, x0 + X, y0, y0 + Y)
        vin = _structured_volume(rng, Y, X, Zp)
        want = np.full_like(vin, 9)
        olib.avo_volume_optimize(oracle.ptr(want), oracle.ptr(vin), X * Zp, Zp, X, Y, C.byref(o.pyr[0].desc), C.byref(sgm), Z, roi)
        wants.append(want)
        tin = torch.from_numpy(vin).cuda()
        tout = torch.full_like(tin, 9)
        keep.append(tin)
        outs.append(tout)
        tiles[i] = abi.SgmTile(tout.data_ptr(), tin.data_ptr(), X * Zp, Zp, Z, roi, C.pointer(pyr.desc))
        total += int(lib.avdm_volume_optimize_scratch_bytes(X, Y, Z))
    scratch = torch.empty(total, dtype=torch.uint8, device="cuda")
    abi.check(lib.avdm_volume_optimize_tiles(len(specs), tiles, _ptr(scratch), C.byref(sgm), _st()))
    torch.cuda.synchronize()
    for want, tout in zip(wants, outs):
        got = tout.cpu().numpy()
        assert np.array_equal(got, want), level_mismatch(got, want)


def test_retrieve_best_depth_bit_exact(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    Y, X, Zp = o.filtered.shape
    Z = len(depths)
    roi = o.droi(sgm.scale * sgm.stepXY)
    vol = torch.from_numpy(o.filtered).cuda()
    dt = torch.empty((Y, X, 2), dtype=torch.float32, device="cuda")
    ds = torch.empty((Y, X, 2), dtype=torch.float32, device="cuda")
    dd = torch.from_numpy(np.asarray(depths, np.float32)).cuda()
    rc1 = abi.camera_fill(sc.K, sc.R[0], sc.C[0], 1)
    abi.check(lib.avdm_volume_retrieve_best_depth(_ptr(dt), X * 8, _ptr(ds), X * 8, _ptr(dd), _ptr(vol), X * Zp, Zp, Z, C.byref(rc1),
                                                  C.byref(sgm), abi.Range(0, Z), roi, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy(), o.sgm_depth_thickness)
    assert np.array_equal(ds.cpu().numpy(), o.sgm_depth_sim)


def test_smooth_and_upscale_bit_exact(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    roiS, roiR = o.droi(sgm.scale * sgm.stepXY), o.droi(ref.scale * ref.stepXY)
    dt = torch.from_numpy(o.sgm_depth_thickness).cuda()
    abi.check(lib.avdm_depth_thickness_smooth_thickness(_ptr(dt), roiS.width * 8, C.byref(sgm), C.byref(ref), roiS, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy(), o.sgm_depth_thickness_smooth)
    h = make_hip_from_oracle(o, sc, sgm, ref)
    X, Y = roiR.width, roiR.height
    up = torch.empty((Y, X, 2), dtype=torch.float32, device="cuda")
    rc = abi.camera_fill(sc.K, sc.R[0], sc.C[0], ref.scale)
    for interp in (0, 1):
        rp = abi.RefineParams.default(interpolateMiddleDepth=interp)
        want = np.empty((Y, X, 2), np.float32)
        from oracle import oracle
        oracle.load().avo_compute_sgm_upscaled_depth_pixsize_map(oracle.ptr(want), X * 8, oracle.ptr(o.sgm_depth_thickness_smooth), roiS.width * 8,
                                                                 C.byref(rc), C.byref(o.pyr[0].desc), C.byref(rp), np.float32(roiS.width) / np.float32(X),
                                                                 roiR)
        abi.check(lib.avdm_compute_sgm_upscaled_depth_pixsize_map(_ptr(up), X * 8, _ptr(dt), roiS.width * 8, C.byref(rc), C.byref(h.pyr[0].desc),
                                                                  C.byref(rp), float(roiS.width) / float(X), roiR, _st()))
        torch.cuda.synchronize()
        assert np.array_equal(up.cpu().numpy(), want), interp


def test_refine_volume_parity(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    h = make_hip_from_oracle(o, sc, sgm, ref)
    h._alloc(len(depths))
    h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
    h.run_refine(0, [1, 2], optimize_enabled=False)
    torch.cuda.synchronize()
    Zr = ref.halfNbDepths * 2 + 1
    a = o.refine_volume[..., :Zr].astype(np.float32)
    b = h.refine_volume.cpu().numpy()[..., :Zr].astype(np.float32)
    # sums of <= 2 sigmoid-filtered similarities in [0, 1], fp16 storage (quantum 2^-10 below 1, 2^-9 below 2)
    diff = np.abs(a - b)
    assert (diff > 2e-3).mean() <= 2e-3, (diff > 2e-3).mean()
    # a T-side border / alpha test falling the other way drops one whole sigmoid term: must stay exceptional
    assert (diff > 0.02).mean() <= 1e-4, ((diff > 0.02).mean(), diff.max())
    # ... and against the LITERAL restatement (== the reference's kernels bit for bit, tests/test_oracle_ref.py) on the same SGM map: the
    # sigmoid filter absorbs most of what the conditioning of the NCC sums does to the SGM volume (there: ~half of the voxels a level apart;
    # here: the literal evaluation itself lies within two fp16 quanta of the well-posed one on all but ~2e-3 of the entries), so the default
    # kernel is held to the reference's arithmetic at twice the quantum of (1), no further than the two distances it is composed of
    o2 = make_oracle(sc, sgm, ref)
    o2.sgm_depth_thickness = o.sgm_depth_thickness.copy()
    o2.run_refine(0, [1, 2], optimize_enabled=False)
    lit = o2.refine_volume[..., :Zr].astype(np.float32)
    floor = float((np.abs(a - lit) > 2e-3).mean())
    d_lit = np.abs(b - lit)
    print("refine volume, default kernel vs literal oracle: %.5f of the entries differ by > 2e-3, %.5f by > 4e-3, %.6f by > 2e-2, max %.3e; the literal "
          "oracle vs the well-posed one: %.5f by > 2e-3; zero pattern differs on %.6f" % ((d_lit > 2e-3).mean(), (d_lit > 4e-3).mean(), (d_lit > 2e-2).mean(),
                                                                                     d_lit.max(), floor, ((lit == 0) != (b == 0)).mean()))
    assert floor < 5e-3, floor  # (measured 1.5e-3; a literal evaluation much further from the well-posed one would call for a tighter look)
    assert (d_lit > 4e-3).mean() <= (diff > 2e-3).mean() + floor + 1e-6, ((d_lit > 4e-3).mean(), floor)
    assert (d_lit > 0.05).mean() <= 1e-4, ((d_lit > 0.05).mean(), d_lit.max())


def test_refine_chunk_window_equals_per_plane_windows(case):
    """The Refine kernel stages ONE T window for the 8 planes of a chunk (the hull of the projected patch on the first and the last plane,
    the centre colour read from the staged records); AVDM_SIM_CHUNK_WINDOW=0 keeps one window per plane with the centre from global memory.
    Same texels, same weights, same sums: the two volumes must be identical except where the two forms of the centre fetch round
    differently (the alpha / colour of the centre enters every sample weight) — held to a handful of fp16 quanta on a vanishing
    fraction of the voxels — and with a depth discontinuity and masked T texels in the scene."""
    import os
    torch = _torch()
    sc, sgm, ref, depths, o = case
    vols = []
    for flag in ("1", "0"):
        os.environ["AVDM_SIM_CHUNK_WINDOW"] = flag
        os.environ["AVDM_SIM_PLANE_PAIRS"] = "0"  # one plane per pass in both runs: the plane pairs have their own test
        try:
            h = make_hip_from_oracle(o, sc, sgm, ref)
            h._alloc(len(depths))
            dt = o.sgm_depth_thickness.copy()
            Yh, Xh = dt.shape[:2]
            dt[Yh // 3:, Xh // 2:, 0] *= 1.04  # a depth step inside workgroups: windows of neighbouring lanes part
            dt[:5, :7, 0] = -1.0
            h.sgm_depth_thickness.copy_(torch.from_numpy(dt))
            h.run_refine(0, [1, 2], optimize_enabled=False)
            torch.cuda.synchronize()
            vols.append(h.refine_volume.cpu().numpy().astype(np.float32))
        finally:
            os.environ.pop("AVDM_SIM_CHUNK_WINDOW", None)
            os.environ.pop("AVDM_SIM_PLANE_PAIRS", None)
    a, b = vols
    diff = np.abs(a - b)
    assert (diff > 0).mean() < 2e-3, (diff > 0).mean()
    assert diff.max() <= 4e-3, diff.max()
    assert (a != 0).mean() > 0.3


def test_plane_pairs_equal_single_planes(capsys):
    """The similarity kernels run several adjacent planes per pass over the patch — eight by default since round 5
    (ncc_accumulate_lds_fixed8_multi<4>), four on leftover chunks (ncc_accumulate_lds_fixed8_quad): the R side of a sample is evaluated once,
    from one plane's patch, for all of them; AVDM_SIM_PLANE_PAIRS=0 keeps one plane per pass.  The R taps of the other planes move by
    ~1e-4 ... 1e-3 texel per depth step (the tilt of the patch's x axis) and the sums are associated differently: the volumes must agree to
    the storage quantum almost everywhere — SGM: uint8 levels, Refine: fp16 sums — and the statistics are printed.  The plane spacing is
    the production one (128 planes over the scene's depth range at 320 x 240 = 256 planes at 12 MP: a coarser list exaggerates the tilt)."""
    import os
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=128, seed=11)
    Z = len(depths)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(3)]
    out = {}
    sgm_map = None
    for flag in ("1", "0"):
        os.environ["AVDM_SIM_PLANE_PAIRS"] = flag
        try:
            h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
            h.run_sgm(0, [1, 2], depths, keep_raw=True)
            torch.cuda.synchronize()
            best, second = h.best_raw.cpu().numpy()[..., :Z].copy(), h.second.cpu().numpy()[..., :Z].copy()
            if sgm_map is None:
                sgm_map = h.sgm_depth_thickness.clone()  # ONE depth map under both Refine runs
            h.sgm_depth_thickness.copy_(sgm_map)
            h.run_refine(0, [1, 2], optimize_enabled=False)
            torch.cuda.synchronize()
            out[flag] = (best, second, h.refine_volume.cpu().numpy().astype(np.float32))
        finally:
            os.environ.pop("AVDM_SIM_PLANE_PAIRS", None)
    (b1, s1, r1), (b0, s0, r0) = out["1"], out["0"]
    fb, mb = level_mismatch(b0, b1)
    fs, ms = level_mismatch(s0, s1)
    gb = float((np.abs(b0.astype(np.int16) - b1.astype(np.int16)) > 1).mean())
    gs = float((np.abs(s0.astype(np.int16) - s1.astype(np.int16)) > 1).mean())
    d = np.abs(r1 - r0)
    with capsys.disabled():
        print("\nplanes per pass (4 / 2) vs single planes: SGM best %.5f of the voxels differ (%.6f by > 1 level, max %d), second %.5f (%.6f, max %d); "
              "Refine %.4f differ, %.5f by more than one fp16 quantum (2e-3), max %.2e" % (fb, gb, mb, fs, gs, ms, (d > 0).mean(), (d > 2e-3).mean(), d.max()))
    assert fb <= 0.02 and fs <= 0.02, (fb, fs)          # uint8 truncation boundaries only
    assert gb <= 2e-4 and gs <= 2e-4, (gb, gs)          # more than one level: a handful of voxels
    assert mb <= 16 and ms <= 16, (mb, ms)              # (the few: low-texture patches whose weighted variance is a rounding residue)
    assert ((b0 == 255) != (b1 == 255)).mean() == 0.0   # validity is decided before the samples: identical
    assert (d > 2e-3).mean() <= 1e-3, (d > 2e-3).mean()
    assert (d > 2e-2).mean() <= 1e-5 and d.max() <= 0.15, ((d > 2e-2).mean(), d.max())
    assert (r1 != 0).mean() > 0.3 and (b1 != 255).mean() > 0.3


def test_sgm_similarity_four_planes_per_pass_equal_the_default(capsys):
    """AVDM_SIM_PLANES8=0 — four planes per pass everywhere, the round-4 default — against the default since round 5 (eight planes per pass where
    two chunks of a workgroup lie in the T camera's range), on full and partial plane ranges and on a tile with an offset: the R side of a
    sample comes from ONE of the eight planes (from one of four with the switch): storage-quantum differences, as between four planes per pass
    and one (test_plane_pairs_equal_single_planes)."""
    import os
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sgm, ref, depths = small_case(width=330, height=250, n_planes=70, seed=3)
    Z = len(depths)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(3)]
    out = {}
    for flag in ("0", "1"):
        os.environ["AVDM_SIM_PLANES8"] = flag
        try:
            res = []
            for roi, tcr in ((None, [(0, Z), (5, 61)]), ((64, 330, 48, 250), None)):
                h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=roi)
                h.run_sgm(0, [1, 2], depths, tc_ranges=tcr, keep_raw=True)
                torch.cuda.synchronize()
                res.append((h.best_raw.cpu().numpy()[..., :Z].copy(), h.second.cpu().numpy()[..., :Z].copy()))
            out[flag] = res
        finally:
            os.environ.pop("AVDM_SIM_PLANES8", None)
    changed = 0.0
    for (b0, s0), (b1, s1) in zip(out["0"], out["1"]):
        assert (b0 != 255).mean() > 0.3
        assert ((b0 == 255) != (b1 == 255)).mean() == 0.0  # validity is decided before the samples
        for a, b in ((b0, b1), (s0, s1)):
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            with capsys.disabled():
                print("\nfour vs eight planes per pass: %.6f of the voxels differ, %.6f by more than one level, max %d" % ((d > 0).mean(), (d > 1).mean(), d.max()))
            assert (d > 0).mean() <= 0.02 and (d > 1).mean() <= 2e-4 and d.max() <= 16, ((d > 0).mean(), (d > 1).mean(), d.max())
            changed += float((d > 0).mean())
    assert changed > 0.0, "the switch did not take effect"


def test_refine_similarity_experiment_equals_the_default(capsys):
    """AVDM_REFINE_PLANES8=0 — two passes of four planes per Refine chunk, the round-4 default — against the default since round 5 (the eight
    planes of a chunk in one pass): the R side of a sample comes from one of eight planes instead of one of four — fp16-quantum differences in
    the accumulated volume, as between four planes per pass and one (test_plane_pairs_equal_single_planes)."""
    import os
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=128, seed=11)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(3)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, [1, 2], depths)
    sgm_map = h.sgm_depth_thickness.clone()
    out = {}
    for flag in ("0", "1"):
        os.environ["AVDM_REFINE_PLANES8"] = flag
        try:
            h.sgm_depth_thickness.copy_(sgm_map)
            h.run_refine(0, [1, 2], optimize_enabled=False)
            torch.cuda.synchronize()
            out[flag] = h.refine_volume.cpu().numpy().astype(np.float32)
        finally:
            os.environ.pop("AVDM_REFINE_PLANES8", None)
    d = np.abs(out["1"] - out["0"])
    with capsys.disabled():
        print("\nAVDM_REFINE_PLANES8 vs default: %.4f of the entries differ, %.5f by more than one fp16 quantum (2e-3), max %.2e" % ((d > 0).mean(), (d > 2e-3).mean(), d.max()))
    assert (out["0"] != 0).mean() > 0.3
    assert ((out["0"] == 0) != (out["1"] == 0)).mean() <= 1e-4
    assert (d > 0).mean() > 0.0, "the switch did not take effect"
    assert (d > 2e-3).mean() <= 1e-3 and (d > 2e-2).mean() <= 1e-5 and d.max() <= 0.15, ((d > 2e-3).mean(), (d > 2e-2).mean(), d.max())


def test_refine_outlier_list_equals_the_wave_fallback(capsys):
    """The outlier list of the default Refine kernel (round 5; AVDM_REFINE_OUTLIER_LIST=0 is the round-4 form): a pixel whose SGM depth is wrong
    projects its patch far from its neighbours' in T; its LANE leaves the eight-plane pass and is appended to a list that refine_outlier_kernel
    works off — every plane from scratch, taps from global memory, the arithmetic of the per-plane fall-back — instead of dragging its whole WAVE
    onto that fall-back.  On a map with wrong depths on 2 % of the SGM pixels (4 x 4 blocks of Refine pixels): the list is used and never full,
    the listed pixels get what the fall-back gave them, the other lanes of their waves the eight-plane pass instead of the one-plane global
    path — fp16-quantum differences, the class of test_plane_pairs_equal_single_planes."""
    import ctypes
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=128, seed=11)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(3)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, [1, 2], depths)
    bad = h.sgm_depth_thickness.clone()
    g = torch.Generator().manual_seed(5)
    density, factor = float(os.environ.get("AVDM_TEST_OUTLIER_DENSITY", "0.02")), float(os.environ.get("AVDM_TEST_OUTLIER_FACTOR", "0.2"))  # (probing aid)
    wrong = (torch.rand(bad.shape[:2], generator=g) < density).to(bad.device) & (bad[..., 0] > 0)
    # much too near: the patch lands ~ 80 texels from its neighbours' in T — farther than a window of the workgroup's hull holds, which is what sends
    # a lane to the list.  (Until the chunk windows could reach outside the T image — session r06_p — a factor of 0.35, ~ 40 texels, was enough: the list's
    # units then came from the border workgroups' anchored windows; sessions r06_s / r06_t: 13 units at 0.35, 6 427 at 0.2.)
    bad[..., 0] = torch.where(wrong, bad[..., 0] * factor, bad[..., 0])
    lib = abi.load()
    lib.avdm_debug_refine_outlier_units.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    out, units = {}, {}
    os.environ["AVDM_REFINE_OUTLIER_STATS"] = "1"
    try:
        for flag in ("0", "1"):
            os.environ["AVDM_REFINE_OUTLIER_LIST"] = flag
            h.sgm_depth_thickness.copy_(bad)
            u = (ctypes.c_uint * 2)()
            lib.avdm_debug_refine_outlier_units(u)  # reset
            h.run_refine(0, [1, 2], optimize_enabled=False)
            torch.cuda.synchronize()
            assert lib.avdm_debug_refine_outlier_units(u) == 0
            units[flag] = (int(u[0]), int(u[1]))
            out[flag] = h.refine_volume.cpu().numpy().astype(np.float32)
    finally:
        os.environ.pop("AVDM_REFINE_OUTLIER_LIST", None)
        os.environ.pop("AVDM_REFINE_OUTLIER_STATS", None)
    d = np.abs(out["1"] - out["0"])
    with capsys.disabled():
        print("\noutlier list: %d units worked off, %d refused; vs the wave fall-back: %.4f of the entries differ, %.5f by more than one fp16 quantum (2e-3), max %.2e"
              % (units["1"] + ((d > 0).mean(), (d > 2e-3).mean(), d.max())))
    assert units["0"] == (0, 0) and units["1"][0] > 1000 and units["1"][1] == 0, units
    assert (out["0"] != 0).mean() > 0.3
    assert ((out["0"] == 0) != (out["1"] == 0)).mean() <= 1e-4
    # (the inlier lanes of a wave with an outlier used to run one plane per pass with fp32 taps from global memory, now the packed eight-plane
    # pass from the window: the Refine volume's tolerance class)
    assert (d > 2e-3).mean() <= 2e-3 and (d > 2e-2).mean() <= 1e-4 and d.max() <= 0.15, ((d > 2e-3).mean(), (d > 2e-2).mean(), d.max())
    # ... AND AGAINST THE ORACLE on the same map with the wrong depths (the list path had only been compared HIP with HIP): the
    # pyramids are the oracle's texel for texel since round 6, so its Refine volume of THIS map is the reference for the listed pixels too —
    # the well-posed evaluation in the Refine volume's tolerance class, over the whole volume and over the pixels with a wrong depth alone
    from oracle import oracle
    o = make_oracle(sc, sgm, ref)
    o.sgm_depth_thickness = bad.cpu().numpy().copy()
    with oracle.well_posed():
        o.run_refine(0, [1, 2], optimize_enabled=False)
    Zr = ref.halfNbDepths * 2 + 1
    want = o.refine_volume[..., :Zr].astype(np.float32)
    got = out["1"][..., :Zr]
    assert got.shape == want.shape, (got.shape, want.shape)
    do = np.abs(got - want)
    step = got.shape[0] // wrong.shape[0]
    wrong_full = np.kron(wrong.cpu().numpy().astype(np.uint8), np.ones((step, step), np.uint8)).astype(bool)[: got.shape[0], : got.shape[1]]
    dw = do[wrong_full]
    with capsys.disabled():
        print("outlier list vs the ORACLE on the same map: %.5f of all entries differ by > 2e-3, %.6f by > 2e-2; on the %d pixels with a wrong depth: %.5f / %.6f, max %.3e"
              % ((do > 2e-3).mean(), (do > 2e-2).mean(), int(wrong_full.sum()), (dw > 2e-3).mean(), (dw > 2e-2).mean(), float(dw.max()) if dw.size else 0.0))
    assert wrong_full.sum() > 500 and (want[wrong_full] != 0).mean() > 0.05
    assert (do > 2e-3).mean() <= 3e-3 and (do > 2e-2).mean() <= 2e-4, ((do > 2e-3).mean(), (do > 2e-2).mean())
    assert (dw > 2e-3).mean() <= 2e-2 and (dw > 2e-2).mean() <= 2e-3, ((dw > 2e-3).mean(), (dw > 2e-2).mean())


def test_refine_best_depth_bit_exact(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    roiR = o.droi(ref.scale * ref.stepXY)
    X, Y = roiR.width, roiR.height
    Zr = ref.halfNbDepths * 2 + 1
    vol = torch.from_numpy(o.refine_volume).cuda()
    Zrp = vol.shape[2]
    up = torch.from_numpy(o.sgm_upscaled).cuda()
    out = torch.empty((Y, X, 2), dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_volume_refine_best_depth(_ptr(out), X * 8, _ptr(up), X * 8, _ptr(vol), X * Zrp * 2, Zrp * 2, Zr, C.byref(ref), roiR, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), o.refined)


def test_optimize_parity(case):
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    h = make_hip_from_oracle(o, sc, sgm, ref)
    roiR = o.droi(ref.scale * ref.stepXY)
    X, Y = roiR.width, roiR.height
    up = torch.from_numpy(o.sgm_upscaled).cuda()
    refined = torch.from_numpy(o.refined).cuda()
    opt = torch.empty((Y, X, 2), dtype=torch.float32, device="cuda")
    var = torch.empty((Y, X), dtype=torch.float32, device="cuda")
    tmp = torch.empty((Y, X), dtype=torch.float32, device="cuda")
    rc = abi.camera_fill(sc.K, sc.R[0], sc.C[0], ref.scale)
    abi.check(lib.avdm_depth_sim_map_optimize_gradient_descent(_ptr(opt), X * 8, _ptr(var), X * 4, _ptr(tmp), X * 4, X, Y, _ptr(up), X * 8,
                                                               _ptr(refined), X * 8, C.byref(rc), C.byref(h.pyr[0].desc), C.byref(ref), roiR, _st()))
    torch.cuda.synchronize()
    assert np.array_equal(var.cpu().numpy(), o.img_variance)
    got, want = opt.cpu().numpy(), o.optimized
    valid = want[..., 0] > 0
    assert np.array_equal(got[..., 0] > 0, valid)
    pix = o.sgm_upscaled[..., 1][valid]
    # 100 Jacobi iterations in fp32 with libm-vs-ocml exp/acos: depth error relative to the pixel size
    rel = np.abs(got[..., 0] - want[..., 0])[valid] / pix
    dsim = np.abs(got[..., 1] - want[..., 1])[valid]
    print("optimize parity: depth rmse / pixSize %.3e, |d sim| max %.3e, 99.9th percentile %.3e" % (np.sqrt(np.mean(rel ** 2)), dsim.max(), np.percentile(dsim, 99.9)))
    assert np.sqrt(np.mean(rel ** 2)) < 1e-3, np.sqrt(np.mean(rel ** 2))  # measured 1.0e-4
    # the similarity of the optimised map (written as a HALF EXR channel, quantum 5e-4 ... 1e-3): 100 iterations with sigmoids on both sides of
    # every step; measured max 1.2e-2, 99.9th percentile 6.4e-3 with the hardware rcp / rsq / exp2 of round 4 (max < 1e-2 with IEEE division)
    assert dsim.max() < 2e-2 and np.percentile(dsim, 99.9) < 1e-2, (dsim.max(), np.percentile(dsim, 99.9))


def test_optimize_point_map_form_is_bit_identical_to_depth_map_form(case):
    """The production form of the colour optimisation keeps {point, depth} per pixel (one ray evaluation per pixel and iteration, no copy of
    the depth map); the reference-shaped form (kernel 19 + kernel 20 per iteration, AVDM_OPT_DEPTH_MAP_FORM=1) must give the same bits —
    with holes in the SGM map (pixels that never move, neighbours without depth) and for 1, 2, 7 and 100 iterations."""
    import os
    torch = _torch()
    sc, sgm, ref, depths, o = case
    lib = abi.load()
    h = make_hip_from_oracle(o, sc, sgm, ref)
    roiR = o.droi(ref.scale * ref.stepXY)
    X, Y = roiR.width, roiR.height
    up_np = o.sgm_upscaled.copy()
    rng = np.random.default_rng(5)
    holes = rng.random((Y, X)) < 0.03
    up_np[holes, 0] = -1.0
    up_np[Y // 3: Y // 3 + 4, X // 4: X // 2, 0] = -2.0
    up = torch.from_numpy(up_np).cuda()
    refined = torch.from_numpy(o.refined).cuda()
    rc = abi.camera_fill(sc.K, sc.R[0], sc.C[0], ref.scale)

    def run(n_iter, legacy):
        rp = abi.RefineParams.default(optimizationNbIterations=n_iter)
        for f, _t in ref._fields_:
            if f != "optimizationNbIterations":
                setattr(rp, f, getattr(ref, f))
        opt = torch.full((Y, X, 2), 7.0, dtype=torch.float32, device="cuda")
        var = torch.empty((Y, X), dtype=torch.float32, device="cuda")
        tmp = torch.empty((Y, X), dtype=torch.float32, device="cuda")
        if legacy:
            os.environ["AVDM_OPT_DEPTH_MAP_FORM"] = "1"
        try:
            abi.check(lib.avdm_depth_sim_map_optimize_gradient_descent(_ptr(opt), X * 8, _ptr(var), X * 4, _ptr(tmp), X * 4, X, Y, _ptr(up), X * 8,
                                                                       _ptr(refined), X * 8, C.byref(rc), C.byref(h.pyr[0].desc), C.byref(rp), roiR, _st()))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("AVDM_OPT_DEPTH_MAP_FORM", None)
        return opt.cpu().numpy()

    for n_iter in (1, 2, 7, 100):
        a, b = run(n_iter, False), run(n_iter, True)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n_iter, int((a.view(np.uint32) != b.view(np.uint32)).sum()))
    assert (run(7, False)[..., 0] > 0).mean() > 0.5


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
def test_end_to_end_depth_rmse(mode):
    """cfg1-like plumbing case: everything on the GPU (own pyramids) vs everything in the oracle; BASELINE bar: depth RMSE < 1e-3."""
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=48, seed=11)
    from oracle import oracle
    o = make_oracle(sc, sgm, ref, filter_mode=mode)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths)
        want = o.run_refine(0, [1, 2])
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, mode) for i in range(3)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, [1, 2], depths)
    got = h.run_refine(0, [1, 2]).cpu().numpy()
    torch.cuda.synchronize()
    both = (want[..., 0] > 0) & (got[..., 0] > 0)
    assert ((want[..., 0] > 0) != (got[..., 0] > 0)).mean() < 5e-3
    gt = sc.gt_depth.numpy()
    err = (got[..., 0] - want[..., 0])[both]
    # a uint8 level flip before SGM can move the WTA plane of a pixel: robust statistic + RMSE over the 99.5 % best pixels
    rmse = np.sqrt(np.mean(np.sort(err ** 2)[: int(0.995 * err.size)]))
    assert rmse < 1e-3, rmse
    # and the GPU result is as close to the analytic ground truth as the oracle's
    assert np.median(np.abs(got[..., 0] - gt)[both]) < 1.05 * np.median(np.abs(want[..., 0] - gt)[both]) + 1e-5


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
def test_texture_unit_parity(mode):
    """software tex2DLod (bilinear + mip-linear + clamp) of the HIP library vs the oracle's, on identical pyramids"""
    torch = _torch()
    from alicevision_amd.pipeline import DevicePyramid
    from oracle import oracle
    sc, sgm, ref, _ = small_case(width=250, height=186)
    hp = oracle.HostPyramid(sc.images[0].numpy(), 1, 128, mode)
    dp = DevicePyramid.from_host_bytes(hp.desc, hp.buf)
    rng = np.random.RandomState(5)
    n = 20000
    uvl = np.stack([rng.uniform(-0.05, 1.05, n), rng.uniform(-0.05, 1.05, n), rng.uniform(-0.5, 8.0, n)], 1).astype(np.float32)
    uvl[: n // 2, 2] = np.floor(uvl[: n // 2, 2].clip(0, 7))  # half of the probes at integral levels (the default path)
    want = np.empty((n, 4), np.float32)
    olib = oracle.load()
    buf = (C.c_float * 4)()
    for i in range(n):
        olib.avo_tex2dlod(C.byref(hp.desc), float(uvl[i, 0]), float(uvl[i, 1]), float(uvl[i, 2]), C.byref(buf))
        want[i] = buf[:]
    d_uvl = torch.from_numpy(uvl).cuda()
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    abi.check(abi.load().avdm_tex2dlod(_ptr(out), C.byref(dp.desc), _ptr(d_uvl), n, _st()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    # FMA contraction only: a few ulp of a value <= 255
    d = np.abs(got - want).max(axis=1)
    assert d.max() < 0.2, d.max()  # FIXED8: a 1/256 weight-bucket flip moves a sample by <= (texel difference)/256
    assert (d > 1e-3).mean() < (0.01 if mode == abi.FILTER_CUDA_FIXED8 else 1e-4), (d > 1e-3).mean()


def test_normal_map_parity(case):
    """avdm_depth_sim_map_compute_normal (Jacobi in double) against the oracle (closed-form eigenvalues): tolerance class — the
    normals agree to 1e-4 rad wherever the smallest eigenvalue is separated, validity masks are identical."""
    torch = _torch()
    from oracle import oracle
    sc, sgm, ref, depths, o = case
    lib, olib = abi.load(), oracle.load()
    roi = o.droi(sgm.scale * sgm.stepXY)
    X, Y = roi.width, roi.height
    dsm = np.ascontiguousarray(o.sgm_depth_sim)
    cam = abi.camera_fill(sc.K, sc.R[0], sc.C[0], sgm.scale)
    want = np.zeros((Y, X, 3), np.float32)
    olib.avo_depth_sim_map_compute_normal(oracle.ptr(want), X * 12, oracle.ptr(dsm), X * 8, C.byref(cam), sgm.stepXY, roi)
    got_t = torch.zeros((Y, X, 3), dtype=torch.float32, device="cuda")
    abi.check(lib.avdm_depth_sim_map_compute_normal(_ptr(got_t), X * 12, _ptr(torch.from_numpy(dsm).cuda()), X * 8, C.byref(cam), sgm.stepXY, roi, _st()))
    torch.cuda.synchronize()
    got = got_t.cpu().numpy()
    inv_w, inv_g = np.all(want == -1.0, axis=-1), np.all(got == -1.0, axis=-1)
    assert np.array_equal(inv_w, inv_g)
    ok = ~inv_w
    assert ok.mean() > 0.5
    cosang = np.einsum("ij,ij->i", want[ok].astype(np.float64), got[ok].astype(np.float64))
    # float32 unit vectors: 1 - cos of two roundings of the same direction is ~1e-7.  The SGM depth map is piecewise constant (plane
    # indices): a few neighbourhoods are degenerate (collinear points, double smallest eigenvalue) and have no unique normal
    assert (cosang > 1.0 - 5e-7).mean() > 0.98, float((cosang > 1.0 - 5e-7).mean())
    assert np.allclose(np.linalg.norm(got[ok], axis=-1), 1.0, atol=1e-5)


def test_consistent_scale_parity():
    """useConsistentScale (Patch.cuh:250-308; off by default in the reference): fractional mip levels per voxel, trilinear taps — the
    similarity and the Refine volumes against the oracle (tolerance classes of the default path, slightly wider: every tap is a
    blend of two levels), with T cameras at other distances than R so that the levels really differ"""
    torch = _torch()
    from oracle import oracle
    sc, sgm, ref, depths = small_case()
    # pull one T camera towards the surface and push the other away: pixel footprints differ by ~ +-15 %
    for i, f in ((1, 0.85), (2, 1.18)):
        sc.C[i] = np.array([sc.C[i][0], sc.C[i][1], 4.0 * (1.0 - f)])
    sc = _rerender(sc, sc)  # the same surface and texture seen from the moved cameras
    sgm = abi.SgmParams.default(useConsistentScale=1)
    ref = abi.RefineParams.default(useConsistentScale=1)
    Z = len(depths)
    o = make_oracle(sc, sgm, ref)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths, optimize=False)
        best64, second64 = o.best_raw[..., :Z].copy(), o.second[..., :Z].copy()
        o.run_sgm(0, [1, 2], depths)
        o.run_refine(0, [1, 2], optimize_enabled=False)
    h = make_hip_from_oracle(o, sc, sgm, ref)
    h.run_sgm(0, [1, 2], depths, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    for got, want in ((h.best_raw.cpu().numpy()[..., :Z], best64), (h.second.cpu().numpy()[..., :Z], second64)):
        frac, mx = level_mismatch(want, got)
        d = np.abs(want.astype(np.int16) - got.astype(np.int16))
        assert (want != 255).mean() > 0.3
        assert frac <= 0.05, (frac, mx)
        assert (d > 1).mean() <= 4e-3, (d > 1).mean()
        assert ((want == 255) != (got == 255)).mean() <= 4e-3
    # the levels matter: the same volumes without consistent scale differ clearly from these
    o0 = make_oracle(sc, abi.SgmParams.default(), abi.RefineParams.default())
    with oracle.well_posed():
        o0.run_sgm(0, [1, 2], depths, optimize=False)
    assert level_mismatch(o0.best_raw[..., :Z], best64)[0] > 0.2

    h._alloc(Z)
    h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
    h.run_refine(0, [1, 2], optimize_enabled=False)
    torch.cuda.synchronize()
    Zr = ref.halfNbDepths * 2 + 1
    a = o.refine_volume[..., :Zr].astype(np.float32)
    b = h.refine_volume.cpu().numpy()[..., :Zr].astype(np.float32)
    diff = np.abs(a - b)
    assert a.max() > 0.5
    assert (diff > 4e-3).mean() <= 4e-3, (diff > 4e-3).mean()
    assert (diff > 0.02).mean() <= 3e-4, ((diff > 0.02).mean(), diff.max())


@pytest.mark.parametrize("sgm_kw,ref_kw", [
    (dict(scale=3, stepXY=1), dict(scale=1, stepXY=1)),   # SGM at level log2(3) = 1.585 of a pyramid that starts at the Refine scale
    (dict(scale=2, stepXY=3), dict(scale=3, stepXY=1)),   # Refine at level log2(3 / 2) = 0.585 of a pyramid that starts at the SGM scale
])
def test_fractional_mip_levels(sgm_kw, ref_kw):
    """sgmScale / refineScale that are not a power-of-two multiple of each other put the coarser stage on a FRACTIONAL mip level: the
    reference's texture unit blends two levels (DeviceMipmapImage.cpp:92-99, deviceMipmappedArray.cu:348).  The similarity volumes of that
    stage come from the plain trilinear kernel (tolerance class of the consistent-scale test: every tap is a blend of two levels), the map
    kernels blend two levels with the operations of tex2DLod — the adaptive-P2 aggregation, the upscale and the colour optimisation stay
    in their bit-exact / tolerance classes on identical inputs — and the tile runs end to end."""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.pipeline import optimize_scratch
    sc, sgm, ref, depths = small_case(width=300, height=228, n_planes=24, seed=5, **sgm_kw)
    for k, v in ref_kw.items():
        setattr(ref, k, v)
    ref.optimizationNbIterations = 8
    Z = len(depths)
    o = make_oracle(sc, sgm, ref)
    lvl_s = np.log2(sgm.scale / min(sgm.scale, ref.scale))
    lvl_r = np.log2(ref.scale / min(sgm.scale, ref.scale))
    assert (lvl_s != int(lvl_s)) or (lvl_r != int(lvl_r))
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths)
        want = o.run_refine(0, [1, 2])
    h = make_hip_from_oracle(o, sc, sgm, ref)
    # (a) sweep
    h.run_sgm(0, [1, 2], depths, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    for got, w in ((h.best_raw.cpu().numpy()[..., :Z], o.best_raw[..., :Z]), (h.second.cpu().numpy()[..., :Z], o.second[..., :Z])):
        d = np.abs(w.astype(np.int16) - got.astype(np.int16))
        assert (w != 255).mean() > 0.3
        assert (d > 0).mean() <= 0.05 and (d > 1).mean() <= 4e-3, ((d > 0).mean(), (d > 1).mean())
        assert ((w == 255) != (got == 255)).mean() <= 4e-3
    # (b) aggregation with the adaptive P2 from the (possibly fractional) SGM level: bit-exact on identical input bytes, then WTA
    lib = abi.load()
    X, Y = o.second.shape[1], o.second.shape[0]
    h.second.copy_(torch.from_numpy(np.ascontiguousarray(o.second)))
    abi.check(lib.avdm_volume_optimize(_ptr(h.best), _ptr(h.second), X * h.Zp, h.Zp, _ptr(h.sgm_scratch), C.byref(h.pyr[0].desc), C.byref(sgm), Z,
                                       o.droi(sgm.scale * sgm.stepXY), _st()))
    torch.cuda.synchronize()
    assert np.array_equal(h.best.cpu().numpy()[..., :Z], o.filtered[..., :Z])
    dt, _ = h.finish_sgm(0, Z)
    torch.cuda.synchronize()
    assert np.array_equal(dt.cpu().numpy(), o.sgm_depth_thickness)
    # (c) Refine on the oracle's SGM map: upscale bit-exact (alpha test on the Refine level), Refine volume in its tolerance class
    h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
    h.run_refine(0, [1, 2], optimize_enabled=False)
    torch.cuda.synchronize()
    assert np.array_equal(h.sgm_upscaled.cpu().numpy(), o.sgm_upscaled)
    Zr = ref.halfNbDepths * 2 + 1
    diff = np.abs(o.refine_volume[..., :Zr].astype(np.float32) - h.refine_volume.cpu().numpy()[..., :Zr].astype(np.float32))
    assert (diff > 4e-3).mean() <= 4e-3 and (diff > 0.02).mean() <= 3e-4, ((diff > 4e-3).mean(), (diff > 0.02).mean(), diff.max())
    # (d) end to end on the GPU
    h.run_sgm(0, [1, 2], depths)
    got = h.run_refine(0, [1, 2]).cpu().numpy()
    both = (want[..., 0] > 0) & (got[..., 0] > 0)
    assert both.mean() > 0.5 and ((want[..., 0] > 0) != (got[..., 0] > 0)).mean() < 1e-2
    err = np.sort((got[..., 0] - want[..., 0])[both] ** 2)
    assert np.sqrt(err[: int(0.99 * err.size)].mean()) < 2e-3


def _rerender(base, moved):
    """images of the analytic scene for the cameras of `moved` (K, R, C), same surface and texture as `base`"""
    import math
    import torch as T
    from alicevision_amd.synthetic import Scene, _surface, _texture, look_at_rotation
    rng = np.random.RandomState(7)
    sc = Scene()
    sc.width, sc.height, sc.K = base.width, base.height, base.K
    sc.C = [np.asarray(c, np.float64) for c in moved.C]
    sc.R = [look_at_rotation(c, np.array([0.0, 0.0, 4.0])) for c in sc.C]
    f = sc.K[0, 0]
    px = 4.0 / f
    waves = []
    for lam_px, a in ((7.0, 0.10), (13.0, 0.12), (29.0, 0.14), (61.0, 0.10)):
        for _ in range(3):
            th = rng.uniform(0, math.pi)
            kk = 2.0 * math.pi / (lam_px * px)
            waves.append((kk * math.cos(th), kk * math.sin(th), rng.uniform(0, 2 * math.pi), a / 1.7, rng.uniform(0.3, 1.0)))
    v, u = T.meshgrid(T.arange(sc.height, dtype=T.float64), T.arange(sc.width, dtype=T.float64), indexing="ij")
    Kinv = np.linalg.inv(sc.K)
    imgs = []
    for i in range(len(sc.C)):
        M = sc.R[i].T @ Kinv
        dx = M[0, 0] * u + M[0, 1] * v + M[0, 2]
        dy = M[1, 0] * u + M[1, 1] * v + M[1, 2]
        dz = M[2, 0] * u + M[2, 1] * v + M[2, 2]
        cx, cy, cz = [float(t) for t in sc.C[i]]
        t = (4.0 - cz) / dz
        for _ in range(12):
            t = (_surface(cx + t * dx, cy + t * dy, 4.0, 0.2) - cz) / dz
        r, g, b = _texture(cx + t * dx, cy + t * dy, waves)
        imgs.append(T.stack([r, g, b, T.ones_like(r)], dim=-1).to(T.float32))
        if i == 0:
            sc.gt_depth = (t * T.sqrt(dx * dx + dy * dy + dz * dz)).to(T.float32)
    sc.images = T.stack(imgs, dim=0).contiguous()
    sc.z_range = base.z_range
    return sc


@pytest.mark.parametrize("group,consistent", [(False, 0), (True, 1)])
def test_custom_patch_pattern_parity(group, consistent):
    """useCustomPatchPattern (Patch.cuh:598-773, patchPattern.cpp; off by default in the reference): circles of colour-weighted samples
    and a full square at a coarser level, one NCC per subpart combined by the subparts' weights — the pattern builder against the
    oracle's (same structure), then the similarity and Refine volumes in the tolerance classes of the default path"""
    torch = _torch()
    from oracle import oracle
    spec = [("full", 2, 0, 1, 0.4), ("circle", 3.5, 12, 0, 0.35), ("circle", 6, 10, 1, 0.25)] if group else \
        [("circle", 4, 16, 0, 0.5), ("full", 3, 0, 1, 0.3), ("circle", 7.5, 24, 2, 0.2)]
    got_pp = abi.build_custom_patch_pattern(spec, group)
    want_pp = abi.PatchPattern()
    assert oracle.load().avo_build_custom_patch_pattern(len(spec), abi.patch_subparts(spec), 1 if group else 0, C.byref(want_pp)) == 0
    assert got_pp.nbSubparts == want_pp.nbSubparts == (2 if group else 3)
    for a, b in zip(got_pp.subparts[:got_pp.nbSubparts], want_pp.subparts[:want_pp.nbSubparts]):
        assert (a.nbCoordinates, a.level, a.downscale, a.weight, a.isCircle, a.wsh) == (b.nbCoordinates, b.level, b.downscale, b.weight, b.isCircle, b.wsh)
        ca = np.array([[c[0], c[1]] for c in a.coordinates[:a.nbCoordinates]]).reshape(-1, 2)
        cb = np.array([[c[0], c[1]] for c in b.coordinates[:b.nbCoordinates]]).reshape(-1, 2)
        assert np.allclose(ca, cb, rtol=0, atol=2e-6)  # cosf / sinf: libm vs the host's

    sc, _, _, depths = small_case()
    sgm = abi.SgmParams.default(useCustomPatchPattern=1, useConsistentScale=consistent)
    ref = abi.RefineParams.default(useCustomPatchPattern=1, useConsistentScale=consistent)
    Z = len(depths)
    o = make_oracle(sc, sgm, ref)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths, optimize=False)
        best64, second64 = o.best_raw[..., :Z].copy(), o.second[..., :Z].copy()
        o.run_sgm(0, [1, 2], depths)
        o.run_refine(0, [1, 2], optimize_enabled=False)
    h = make_hip_from_oracle(o, sc, sgm, ref)
    h.run_sgm(0, [1, 2], depths, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    for got, want in ((h.best_raw.cpu().numpy()[..., :Z], best64), (h.second.cpu().numpy()[..., :Z], second64)):
        frac, mx = level_mismatch(want, got)
        d = np.abs(want.astype(np.int16) - got.astype(np.int16))
        assert (want != 255).mean() > 0.3
        assert frac <= 0.05, (frac, mx)
        assert (d > 1).mean() <= 4e-3, (d > 1).mean()
        assert ((want == 255) != (got == 255)).mean() <= 4e-3
    # the pattern matters: the default square patch gives clearly different volumes
    o0 = make_oracle(sc, abi.SgmParams.default(), abi.RefineParams.default())
    with oracle.well_posed():
        o0.run_sgm(0, [1, 2], depths, optimize=False)
    assert level_mismatch(o0.best_raw[..., :Z], best64)[0] > 0.2

    h._alloc(Z)
    h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
    h.run_refine(0, [1, 2], optimize_enabled=False)
    torch.cuda.synchronize()
    Zr = ref.halfNbDepths * 2 + 1
    a = o.refine_volume[..., :Zr].astype(np.float32)
    b = h.refine_volume.cpu().numpy()[..., :Zr].astype(np.float32)
    diff = np.abs(a - b)
    assert a.max() > 0.3
    assert (diff > 4e-3).mean() <= 4e-3, (diff > 4e-3).mean()
    assert (diff > 0.02).mean() <= 3e-4, ((diff > 0.02).mean(), diff.max())

    # without a pattern the entry points refuse (here: the library keeps the last one, so only the flag-less call is checked)
    sgm0 = abi.SgmParams.default()
    assert sgm0.useCustomPatchPattern == 0


@pytest.mark.parametrize("mode", [abi.FILTER_CUDA_FIXED8, abi.FILTER_EXACT])
def test_odd_sizes_and_offset_roi_parity(mode):
    """image sizes that are no multiple of anything, a tile ROI in the middle of the image that touches two image borders, ragged
    plane ranges: the LDS windows (paired / half-paired records) are clipped at image borders and the last staged column pairs with
    itself — the volumes must stay in the tolerance classes of the default case"""
    torch = _torch()
    from oracle import oracle
    from alicevision_amd.synthetic import make_scene, plane_depths
    sc = make_scene(3, 251, 189, seed=11)
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default()
    depths = plane_depths(sc, 27)
    Z = len(depths)
    roi = (100, 251, 0, 132)  # x to the right image border, y from the top border; multiples of scale * step at the open ends
    rng_t = [(0, 27), (5, 22)]
    o = make_oracle(sc, sgm, ref, filter_mode=mode, roi=roi)
    with oracle.well_posed():
        o.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False)
        best64, second64 = o.best_raw[..., :Z].copy(), o.second[..., :Z].copy()
        o.run_sgm(0, [1, 2], depths, tc_ranges=rng_t)
        o.run_refine(0, [1, 2], optimize_enabled=False)
    h = make_hip_from_oracle(o, sc, sgm, ref, roi=roi)
    h.run_sgm(0, [1, 2], depths, tc_ranges=rng_t, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    for got, want in ((h.best_raw.cpu().numpy()[..., :Z], best64), (h.second.cpu().numpy()[..., :Z], second64)):
        assert got.shape == want.shape
        frac, mx = level_mismatch(want, got)
        d = np.abs(want.astype(np.int16) - got.astype(np.int16))
        assert (want != 255).mean() > 0.2
        assert frac <= (0.035 if mode == abi.FILTER_CUDA_FIXED8 else 0.015), (frac, mx)
        assert (d > 1).mean() <= 3e-3, (d > 1).mean()
        assert ((want == 255) != (got == 255)).mean() <= 3e-3
    h._alloc(Z)
    h.sgm_depth_thickness.copy_(torch.from_numpy(o.sgm_depth_thickness))
    h.run_refine(0, [1, 2], optimize_enabled=False)
    torch.cuda.synchronize()
    Zr = ref.halfNbDepths * 2 + 1
    a = o.refine_volume[..., :Zr].astype(np.float32)
    b = h.refine_volume.cpu().numpy()[..., :Zr].astype(np.float32)
    assert a.shape == b.shape and a.max() > 0.5
    diff = np.abs(a - b)
    assert (diff > 2e-3).mean() <= 3e-3, (diff > 2e-3).mean()
    assert (diff > 0.02).mean() <= 2e-4, ((diff > 0.02).mean(), diff.max())


def test_full_size_cfg3_properties():
    """BASELINE cfg3 at full size (12 MP, 256 planes, 10 T cameras) — where the oracle would take hours — through size-independent
    properties: the depth map reproduces the analytic surface, two runs are bit-identical (no race in the aggregated / atomically
    published stages), and the SGM aggregation of the same volume through the batched-tiles entry point equals the single call."""
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid, optimize_scratch
    from alicevision_amd.synthetic import make_scene, plane_depths
    V, W, H, Z, T = 11, 4000, 3000, 256, 10
    sc = make_scene(V, W, H, seed=3, device="cuda:0")
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default(optimizationNbIterations=20)
    pyr = [DevicePyramid(sc.images[v], 1, 128, abi.FILTER_CUDA_FIXED8, device="cuda:0") for v in range(V)]
    depths = plane_depths(sc, Z)
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, device="cuda:0")
    tcs = list(range(1, T + 1))
    h.run_sgm(0, tcs, depths, keep_raw=True)
    out1 = h.run_refine(0, tcs).clone()
    vol_in = h.second.clone()  # SGM input volume (second best, after update-uninitialised)
    h.run_sgm(0, tcs, depths)
    out2 = h.run_refine(0, tcs)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    depth = out1[..., 0].cpu().numpy()
    gt = sc.gt_depth.cpu().numpy()
    m = depth > 0
    inner = np.zeros_like(m)
    inner[32:-32, 32:-32] = True
    assert m[inner].mean() > 0.97
    rel = np.abs(depth - gt)[m & inner] / gt[m & inner]
    # (baseline 0.3 at depth 4: one refine sub-sample is ~1e-4 of the depth)
    assert np.median(rel) < 3e-4 and np.percentile(rel, 90) < 1e-3, (float(np.median(rel)), float(np.percentile(rel, 90)))

    # aggregation: one call vs the batched entry point with the same single tile
    lib = abi.load()
    roi = h.droi(sgm.scale * sgm.stepXY)
    X, Y, Zp = roi.width, roi.height, vol_in.shape[-1]
    a, b = torch.empty_like(vol_in), torch.empty_like(vol_in)
    scratch = optimize_scratch(lib, X, Y, Z)
    abi.check(lib.avdm_volume_optimize(_ptr(a), _ptr(vol_in), X * Zp, Zp, _ptr(scratch), C.byref(pyr[0].desc), C.byref(sgm), Z, roi, _st()))
    tile = abi.SgmTile(out_vol=b.data_ptr(), in_vol=vol_in.data_ptr(), pitch_y=X * Zp, pitch_x=Zp, last_depth_index=Z, roi=roi,
                       rc_pyr=C.pointer(pyr[0].desc))
    abi.check(lib.avdm_volume_optimize_tiles(1, C.byref(tile), _ptr(scratch), C.byref(sgm), _st()))
    torch.cuda.synchronize()
    assert torch.equal(a[..., :Z], b[..., :Z])
    # the aggregated volume is not the input, and every column still has a minimum below "invalid"
    assert not torch.equal(a[..., :Z], vol_in[..., :Z])
    assert float((a[..., :Z].min(dim=-1).values < 255).float().mean().item()) > 0.97


def test_full_size_cfg2_sweep_only_properties():
    """BASELINE cfg2 at full size (1920x1080, 128 planes, 4 T cameras, plane sweep only — no SGM aggregation, no Refine): the
    winner-take-all depth of the raw similarity volume is within a plane spacing of the analytic surface on most pixels, reruns are
    bit-identical, and every valid depth is one of the plane-sweep depths of its pixel's ray (retrieve-best-depth invariant)."""
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from alicevision_amd.synthetic import make_scene, plane_depths
    V, W, H, Z, T = 5, 1920, 1080, 128, 4
    sc = make_scene(V, W, H, seed=4, device="cuda:0")
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default()
    pyr = [DevicePyramid(sc.images[v], 1, 128, abi.FILTER_CUDA_FIXED8, device="cuda:0") for v in range(V)]
    depths = plane_depths(sc, Z)
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, device="cuda:0")
    _, ds1 = h.run_sgm(0, list(range(1, T + 1)), depths, optimize=False)
    ds1 = ds1.clone()
    _, ds2 = h.run_sgm(0, list(range(1, T + 1)), depths, optimize=False)
    torch.cuda.synchronize()
    assert torch.equal(ds1, ds2)
    depth = ds1[..., 0].cpu().numpy()
    step = sgm.scale * sgm.stepXY
    gt = sc.gt_depth.cpu().numpy()[::step, ::step][:depth.shape[0], :depth.shape[1]]
    m = depth > 0
    m[:4] = m[-4:] = False
    m[:, :4] = m[:, -4:] = False
    assert m.mean() > 0.8
    spacing = float(np.max(np.abs(np.diff(depths))))
    err = np.abs(depth - gt)[m]  # no aggregation: the per-pixel arg-min is noisy, most pixels still land next to the surface
    assert np.median(err) < 1.5 * spacing and (err < 3.0 * spacing).mean() > 0.75, (float(np.median(err)), spacing)
    # depth along the ray of a fronto-parallel plane at distance z from the camera plane: z / cos(angle to the optical axis)
    f, cx, cy = sc.K[0, 0], sc.K[0, 2], sc.K[1, 2]
    ys, xs = np.mgrid[0:depth.shape[0], 0:depth.shape[1]]
    cosang = 1.0 / np.sqrt(1.0 + ((xs * step - cx) / f) ** 2 + ((ys * step - cy) / f) ** 2)
    zplane = depth * cosang
    nearest = np.abs(zplane[..., None][m][:200000] - depths[None, :]).min(axis=-1)
    assert np.percentile(nearest / zplane[m][:200000], 99) < 2e-5


_PARITY_CACHE = {}
# cases that also go through the reference's CUDA-like evaluation (its platform spread) and the literal kernel's per-deviation switches
_ATTRIBUTED = ("cfg1", "crop3")
# cases held against the LITERAL oracle (= the reference's own code) only
_LITERAL_ONLY = ("tile24mp_interior", "tile24mp_corner", "tile12mp_corner_10T", "tile12mp_corner", "crop3_corner", "crop3_far_corner")


def _parity_case(name):
    """scripts/parity_report.py's measurement of one case, once per session (several tests read it); AVDM_PARITY_DUMP=<dir> keeps the JSON"""
    _torch()
    import json
    import os
    import sys
    if name not in _PARITY_CACHE:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
        import parity_report
        attributed = name in _ATTRIBUTED
        # the reference's own platform spread (oracle/_ref's CUDA-like evaluation against its literal one): the attributed cases (the tiles':
        # scripts/platform_spread.py on the CPU, profiles/r05_platform_spread_tiles.json)
        # (the 24 MP tile: against the literal evaluation — the reference's own arithmetic — only; its 1.5 M pixels cost the oracle over a minute
        # per evaluation, and the driver's limit for this suite is 20 minutes.  scripts/parity_report.py takes both evaluations and the interior
        # tile in a session of its own: profiles/r05_*_parity_tile24mp_*.json)
        # (round 6: the corner crops and the ten-T-camera tile too — the suite's time is the oracle's, and the well-posed evaluation of the same
        # geometry is held by crop3 / tile12mp_*; every case also runs the product's reference-arithmetic mode, which costs no oracle time)
        _PARITY_CACHE[name] = parity_report.run_case(name, parity_report.CASES[name], abi.FILTER_CUDA_FIXED8, with_ref=False, gpu_literal=True, spread=attributed,
                                                     deviations=tuple(parity_report.DEVIATIONS) if attributed else (),
                                                     modes=("literal",) if name in _LITERAL_ONLY else ("well_posed", "literal"), strict=("sgm", "all"))
        d = os.environ.get("AVDM_PARITY_DUMP")
        if d:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "parity_%s.json" % name), "w") as f:
                json.dump(_PARITY_CACHE[name], f, indent=1)
    return _PARITY_CACHE[name]


def _assert_literal_on_gpu(gl):
    """AVDM_SIM_LITERAL=1 — the reference's arithmetic as written, on the GPU (csrc/avdm_literal.hip) — against the oracle's literal mode
    (= the reference's own code compiled for the CPU, bit for bit: tests/test_oracle_ref.py), on identical pyramids, NO trimming: with the
    conditioning of the sums out of the comparison what is left is the device library's expf against glibc's, amplified by the
    cancellation in the reference's variance — the bar holds untrimmed."""
    assert gl["final_depth"]["rmse_untrimmed"] < 1e-3, gl["final_depth"]
    assert gl["final_depth"]["validity_differs"] < 1e-3
    lv = gl["similarity_volume_levels"]
    assert lv["0"] > 0.97 and lv["2"] + lv["3+"] < 3e-3 and lv["validity_differs"] < 1e-3, lv


def _assert_reference_arithmetic(r, final_bar=1e-4):
    """THE PARITY MODE OF THE PRODUCT (round 6; avdm_sgm_params_t / avdm_refine_params_t::referenceArithmetic, the CLI's --sgmReferenceArithmetic /
    --refineReferenceArithmetic), everything on the GPU FROM THE GPU'S OWN PYRAMIDS against the literal oracle = the reference's own kernels and host
    classes compiled for the CPU (tests/test_oracle_ref.py), NO trimming:
      * the Lab pyramids equal the oracle's texel for texel (glibc's cbrtf restated for the device, no contraction);
      * the SGM sweep in the reference's arithmetic (strict_sgm_kernel: every operation of compNCCby3DptsYK in its order, expf to the bits of the pinned
        build's C library): the similarity volume, the aggregated volume and the winner-take-all depths are IDENTICAL to the reference's — every byte;
        with the default Refine kernels the final depth map is inside BASELINE's bar (< 1e-3) on every case, the tiles of DESIGN.md section 2 included;
      * both sweeps in the reference's arithmetic: the Refine volume and the refined map are identical too; what is left is the colour optimisation
        (tolerance class: acosf / expf of the device library), < 1e-4 untrimmed."""
    assert r["pyramid_texels_differing"]["max_over_levels"] == 0.0, r["pyramid_texels_differing"]
    for key in ("reference_arithmetic_sgm_vs_oracle_literal", "reference_arithmetic_all_vs_oracle_literal"):
        m = r[key]
        assert m["similarity_volume_levels"]["0"] == 1.0 and m["sgm_filtered_volume_levels"]["0"] == 1.0 and m["sgm_wta_depth_differs"] == 0.0, (key, m)
        assert m["final_depth"]["validity_differs"] == 0.0, (key, m["final_depth"])
    sgm_only, both = r["reference_arithmetic_sgm_vs_oracle_literal"], r["reference_arithmetic_all_vs_oracle_literal"]
    assert sgm_only["final_depth"]["rmse_untrimmed"] < 1e-3, sgm_only["final_depth"]  # BASELINE's bar, untrimmed, the default Refine kernels
    assert both["refine_volume_abs"]["identical"] == 1.0, both["refine_volume_abs"]
    assert both["refined_depth"]["max_abs"] == 0.0, both["refined_depth"]
    assert both["final_depth"]["rmse_untrimmed"] < final_bar, both["final_depth"]
    # The similarity channel as the program writes it (one half per pixel, mapIO.cpp:403-540).  The Refine stage's similarity is IDENTICAL in the
    # parity mode.  AFTER the colour optimisation the channel is (1 - closeToRough) (w sim + (1 - w) depthEnergy / 20) (mapKernels.cuh:604) with
    # depthEnergy the smoothness ANGLE in degrees between neighbouring depths: on a smooth surface that angle is made of the last bits of the
    # depths, so ANY two evaluations — the reference's own two included (profiles/r06_platform_spread_sim_cfg1.json: 1.5 % identical halfs, p99
    # |d| 7.0) — draw it afresh; the parity mode, whose optimisation starts from identical maps and differs by its tolerance class only, must
    # be at least as close to the reference as the default mode is (test_deviation_attribution holds the default to the reference's own spread)
    assert both["refined_sim"]["identical_halfs"] == 1.0, both["refined_sim"]
    if "final_sim" in r["literal"]:
        assert both["final_sim"]["rmse"] <= r["literal"]["final_sim"]["rmse"], (both["final_sim"], r["literal"]["final_sim"])


def test_parity_table_cfg1():
    """The measured parity table of DESIGN.md section 2, asserted (scripts/parity_report.py; SURVEY 8d.1's cfg1: 3 views 640 x 480, 64 planes,
    single tile): everything on the GPU against everything in the oracle, NO trimming of the depth error —
      * well-posed oracle (double-precision NCC sums, exact R pixel): untrimmed final depth RMSE < 1e-3 (BASELINE.json's bar), similarity
        volume within one uint8 level except on < 0.1 % of the voxels;
      * LITERAL oracle (the reference's fp32 arithmetic as written; equal, bit for bit, to the reference's own kernels compiled for the CPU,
        tests/test_oracle_ref.py): the fp32 NCC sums of the reference are themselves several levels away from their exact value on ~40 % of
        the voxels, so the volumes differ accordingly; the depth map still agrees to RMSE < 1e-3 over the best 99.5 % of the pixels and to
        < 5e-3 untrimmed, and the GPU result is as close to the analytic ground truth as the literal one;
      * the attribution: the SAME literal arithmetic run on the GPU (AVDM_SIM_LITERAL=1) agrees with the literal oracle to < 1e-3 UNTRIMMED
        — the distance of the default kernels to the literal evaluation is the conditioning of the reference's sums, not the device."""
    r = _parity_case("cfg1")
    wp, lit = r["well_posed"], r["literal"]
    assert wp["final_depth"]["rmse_untrimmed"] < 1e-3, wp["final_depth"]
    assert wp["final_depth"]["validity_differs"] < 1e-3
    lv = wp["similarity_volume_levels"]
    assert lv["2"] + lv["3+"] < 1e-3 and lv["1"] < 0.03 and lv["validity_differs"] < 1e-3, lv
    assert lit["final_depth"]["rmse_best_99.5pct"] < 1e-3, lit["final_depth"]
    assert lit["final_depth"]["rmse_untrimmed"] < 1e-3, lit["final_depth"]  # 6.7e-4 since the knife-edge rows run the reference's own border test (2.6e-3 in rounds 2-3)
    assert lit["final_depth"]["validity_differs"] < 2e-3
    # the whole image is one tile: knife-edge rows on all four borders, and the validity mask of the similarity volume equals the reference code's
    assert lit["similarity_volume_levels"]["validity_differs"] == 0.0, lit["similarity_volume_levels"]
    g = lit["median_abs_vs_ground_truth"]
    assert g["gpu"] <= 1.02 * g["oracle"] + 1e-6, g
    _assert_literal_on_gpu(r["gpu_literal_vs_oracle_literal"])
    _assert_reference_arithmetic(r)


@pytest.mark.parametrize("name", ["crop2", "crop3"])
def test_parity_table_crops_of_the_full_size_geometry(name):
    """crop2 / crop3 of DESIGN.md's table: 512 x 512 crops of BASELINE's configurations 2 and 3 — the same cameras, image size (1920 x 1080 /
    4000 x 3000) and plane count (128 / 256), 4 T cameras — everything on the GPU against everything in the oracle, NO trimming.  At this
    geometry the bar (final depth RMSE < 1e-3) holds untrimmed against BOTH evaluations of the oracle — the well-posed one and the literal
    one, i.e. the reference's own arithmetic — and for the literal arithmetic run on the GPU."""
    r = _parity_case(name)
    wp, lit = r["well_posed"], r["literal"]
    for m in (wp, lit):
        assert m["final_depth"]["rmse_untrimmed"] < 1e-3, m["final_depth"]
        assert m["final_depth"]["validity_differs"] < 1e-3
        g = m["median_abs_vs_ground_truth"]
        assert g["gpu"] <= 1.02 * g["oracle"] + 1e-6, g
    lv = wp["similarity_volume_levels"]
    assert lv["2"] + lv["3+"] < 2e-3 and lv["1"] < 0.05 and lv["validity_differs"] < 1e-3, lv
    _assert_literal_on_gpu(r["gpu_literal_vs_oracle_literal"])
    _assert_reference_arithmetic(r)


@pytest.mark.parametrize("name", ["crop3_corner", "crop3_far_corner"])
def test_parity_at_the_real_shape_of_cfg3(name):
    """BASELINE's configuration 3 at its REAL shape: the image corners of the 4000 x 3000 frame — border rejection
    (Patch.cuh:486-496) and clamp addressing inside the tile — and all TEN T cameras of the 11-view scene bench.py runs, the outer rings
    (2 x and 3 x the inner ring's baseline) included; 256 planes, everything on the GPU against everything in the oracle, NO trimming.
    The bar (final depth RMSE < 1e-3) holds against both evaluations of the oracle.  In the corner crops two of the tile's borders are image
    borders: the rows / columns where the patch centre lies exactly wsh + 2 pixels from the border are valid or not by the last bit of the
    reference's re-projection (literal mode) — the default kernels test the exact pixel (DESIGN.md "knife-edge rows") — so the validity masks
    may differ on those rows against the literal oracle (two of 128 SGM columns / rows), not against the well-posed one."""
    r = _parity_case(name)
    lit = r["literal"]  # (round 6: against the reference's own arithmetic only — the ten-T-camera crop of rounds 4-5 became the ten-T-camera TILE below)
    corner = "corner" in name
    for key, m in (("literal", lit),):
        assert m["final_depth"]["rmse_untrimmed"] < 1e-3, (key, m["final_depth"])
        assert m["final_depth"]["validity_differs"] < (0.03 if corner and key == "literal" else 1e-3), (key, m["final_depth"])
        g = m["median_abs_vs_ground_truth"]
        assert g["gpu"] <= 1.02 * g["oracle"] + 1e-6, (key, g)
    gl = r["gpu_literal_vs_oracle_literal"]
    assert gl["final_depth"]["rmse_untrimmed"] < 1e-3, gl["final_depth"]
    _assert_reference_arithmetic(r)
    # the knife-edge rows of the SGM stage (crop3_corner: stage column / row 3): the default kernels evaluate the reference's own border test
    # there (avdm_similarity.hip lit::) — the validity of EVERY voxel of the similarity volume equals the reference code's (1.2 % of the
    # voxels differed in rounds 1-3, when the kernels tested the exact pixel)
    assert lit["similarity_volume_levels"]["validity_differs"] == 0.0, lit["similarity_volume_levels"]


@pytest.mark.parametrize("name", ["tile12mp_interior", "tile12mp_corner"])
def test_parity_of_default_tiles_at_12mp(name):
    """Two tiles of the DEFAULT tiling of a 12 MP image (mvsUtils::getTileRoiList: buffer 1024, padding 64 -> 5 x 4 tiles of 864 x 816; tile
    (2, 1) in the interior, tile (4, 3) clipped at the far image corner), laid out and aggregated over the tile BUFFER like the reference
    (OracleDepthMap(tile_buffer=...) is pinned to the reference's own Sgm.cpp / Refine.cpp, tests/test_oracle_ref.py): 256 planes, 2 T
    cameras, everything on the GPU against everything in the oracle, NO trimming."""
    r = _parity_case(name)
    corner = "corner" in name
    # (the corner tile: against the reference's own arithmetic only since round 6 — its well-posed figure, 9.1e-5, is profiles/r06_d_parity_tile12mp_corner.json)
    for key, m in [(k, r[k]) for k in ("well_posed", "literal") if k in r]:
        # the corner tile against the LITERAL oracle: 1.24e-3 untrimmed, 1.3e-4 over the best 99.5 %, identical validity masks, the literal
        # arithmetic on the GPU at 6.6e-5.  Located oracle against oracle on the CPU: TWO SGM pixels of 43 000 whose winner-take-all plane flips
        # between near-equal minima 30 planes apart — 36 full-size pixels carry the excess, 2.0e-4 without them
        # (KNOWN GAP of the default mode on the corner tile, a regression guard and not the bar: BASELINE's bar is asserted for this tile in the
        # reference-arithmetic mode below)
        bar = 2e-3 if corner and key == "literal" else 1e-3
        assert m["final_depth"]["rmse_untrimmed"] < bar, (key, m["final_depth"])
        assert m["final_depth"]["rmse_best_99.5pct"] < 3e-4, (key, m["final_depth"])
        assert m["final_depth"]["validity_differs"] < (0.03 if corner and key == "literal" else 1e-3), (key, m["final_depth"])
    if "well_posed" in r:
        lv = r["well_posed"]["similarity_volume_levels"]
        assert lv["2"] + lv["3+"] < 3e-3 and lv["1"] < 0.06 and lv["validity_differs"] < 1e-3, lv
    gl = r["gpu_literal_vs_oracle_literal"]
    assert gl["final_depth"]["rmse_untrimmed"] < 1e-3 and gl["final_depth"]["validity_differs"] < 1e-3, gl["final_depth"]
    # ... and in the product's reference-arithmetic mode the corner tile is where every other is: identical volumes, 7.5e-5 / 1.9e-5 (session r06_a)
    _assert_reference_arithmetic(r)


def test_parity_of_the_ten_t_camera_corner_tile_at_12mp():
    """The clipped corner tile of the default 12 MP tiling with all TEN T cameras of the bench's 11-view scene: best / second-best
    merging over ten sweeps (deviceSimilarityVolumeKernels.cuh:221-232) at a clipped tile is the bench's real shape.  Against the literal oracle
    (= the reference's own code), NO trimming: the default kernels inside BASELINE's bar, the reference-arithmetic mode identical volume for volume."""
    r = _parity_case("tile12mp_corner_10T")
    lit = r["literal"]
    assert lit["final_depth"]["rmse_untrimmed"] < 1e-3, lit["final_depth"]
    assert lit["final_depth"]["rmse_best_99.5pct"] < 3e-4 and lit["final_depth"]["validity_differs"] < 0.03, lit["final_depth"]
    assert lit["similarity_volume_levels"]["validity_differs"] < 1e-3, lit["similarity_volume_levels"]
    _assert_reference_arithmetic(r)


@pytest.mark.parametrize("name", ["tile24mp_interior", "tile24mp_corner"])
def test_parity_of_the_cfg5_tiles_at_24mp(name):
    """BASELINE configuration 5 at its OWN shape: a 24 MP frame (6000 x 4000) cut by `--tileBufferWidth 1664 --tileBufferHeight 1152
    --tilePadding 64` into 4 x 4 tiles of 1564 x 1064 (mvsUtils/TileParams.cpp:15-61; the grid is pinned to the reference's own getTileRoiList by
    tests/test_host_ref.py) — tile (1, 1) in the interior and tile (3, 3), clipped at the far image corner (1500 x 1000) — laid out and aggregated
    over the NON-SQUARE tile buffer like the reference (deviceSimilarityVolume.cu:278-283: 416 x 288 SGM columns / rows): 256 planes, 2 T cameras,
    everything on the GPU against the oracle's LITERAL evaluation (= the reference's kernels and host classes compiled for the CPU, bit for bit), NO
    trimming.

    BASELINE's bar (< 1e-3) is asserted in the product's REFERENCE-ARITHMETIC mode, where these tiles equal the reference volume for volume
    (session r06_a: 4.2e-5 / 4.7e-5 with the SGM sweep alone in that mode, 8e-6 / 1.3e-5 with both sweeps).

    KNOWN GAP OF THE DEFAULT (fast) MODE, stated as what it is: against the literal evaluation these two tiles measure 1.06e-3 / 9.4e-4 (rounds 5-6;
    2.9e-5 / 3.8e-5 against the well-posed evaluation of the same formulas) — a handful of SGM pixels (0.1 %) whose winner-take-all plane flips
    between two near-equal minima, because the reference forms its weighted variance as a difference of fp32 sums of ~5e6 and the default kernels
    do not.  The default mode's assertion below is a regression guard at the measured value + margin, NOT the bar."""
    r = _parity_case(name)
    _assert_reference_arithmetic(r)
    lit = r["literal"]
    assert lit["final_depth"]["rmse_untrimmed"] < 1.5e-3, lit["final_depth"]            # known gap of the default mode: 1.06e-3 / 9.4e-4 measured
    assert lit["final_depth"]["rmse_best_99.5pct"] < 3e-4, lit["final_depth"]
    assert lit["final_depth"]["validity_differs"] < 0.03, lit["final_depth"]
    assert lit["similarity_volume_levels"]["validity_differs"] < 1e-3, lit["similarity_volume_levels"]
    gl = r["gpu_literal_vs_oracle_literal"]
    assert gl["final_depth"]["rmse_untrimmed"] < 1e-3 and gl["final_depth"]["validity_differs"] < 1e-3, gl["final_depth"]


@pytest.mark.parametrize("name", ["cfg1", "crop3"])
def test_deviation_attribution(name):
    """the distance between the default similarity kernels and the reference's arithmetic — at BASELINE's bar, measured
    against the reference's OWN platform spread, and attributed deviation by deviation (profiles/r04_deviation_table.json has all four scenes,
    the fast-path variant builds included).

    * The bar.  Final depth RMSE of the default kernels against the LITERAL oracle (= the reference's kernels compiled for the CPU, bit for bit),
      untrimmed: 6.7e-4 on cfg1 (2.6e-3 in rounds 2-3), 2.6e-4 on crop3 — below 1e-3.
    * The yardstick.  BASELINE's bar is depth RMSE vs "the reference CUDA path".  The reference's sources evaluated two equally faithful ways —
      every fp32 operation as written (the literal oracle) and the way an nvcc build evaluates them as far as this container can tell
      (oracle/_ref/libavdm_ref_cuda.so: FMA contraction, the documented error model of the fast intrinsics; tests/test_platform_spread.py) —
      differ from EACH OTHER by X = 1.8e-3 on cfg1 and 2.6e-4 on crop3, untrimmed: the reference forms its NCC variance as a difference of fp32
      sums of ~5e6 and decides the validity of the rows next to the image border by the last bit of a re-projection.  The default kernels are
      within 1.15 X of BOTH evaluations (0.38 X / 1.01 X on cfg1, 0.96 X / 1.01 X on crop3) and nearer to the well-posed value both approximate
      than either of them is.
    * The attribution — the literal arithmetic on the GPU with ONE deviation of the default kernels switched on (AVDM_SIM_LITERAL_DEV):
        crop3 (no image border in the tile): EVERY single deviation — shifted sums, one merged exp2, homogeneous projection + v_rcp, R side
        shared by four planes — moves the literal evaluation by 0.9-1.0 of the full distance: the ill-conditioned sums decorrelate under any
        perturbation, the distances saturate instead of adding; only the centre colour at the exact pixel changes nothing;
        cfg1 (the tile is the image): what rounds 1-3 carried there was the border test on the exact pixel ALONE (2.6e-3: 0.98 of their
        distance) — on the knife-edge rows the kernels now run the reference's own test (avdm_similarity.hip lit::) and are at 6.7e-4;
      all four remaining deviations together reproduce the default kernels (within 1e-3 of the well-posed oracle, volumes identical on > 95 %)."""
    r = _parity_case(name)
    if "platform_spread" not in r:
        pytest.skip("oracle/_ref/libavdm_ref_cuda.so did not travel")
    sp, dev = r["platform_spread"], r["literal_plus_deviation"]
    rm = lambda m: m["final_depth"]["rmse_untrimmed"]
    X = rm(sp["cuda_vs_literal"])
    d_lit, d_wp, d_cuda = rm(r["literal"]), rm(r["well_posed"]), rm(sp["default_vs_cuda"])
    floor = rm(r["gpu_literal_vs_oracle_literal"])
    info = {"X": X, "default_vs_literal": d_lit, "default_vs_cuda": d_cuda, "default_vs_well_posed": d_wp, "literal_gpu_vs_literal": floor,
            "literal_plus": {k: rm(v["vs_literal"]) for k, v in dev.items()}}
    print(name, info)
    # two faithful evaluations of the reference differ by far more than its literal arithmetic differs between the CPU and the GPU
    assert X > 2.0 * floor and floor < 1e-3, info
    # the default kernels: BASELINE's bar against the reference's arithmetic as written, untrimmed, and within the reference's own spread of both evaluations ...
    assert d_lit < 1e-3, info
    assert d_lit <= 1.15 * X and d_cuda <= 1.15 * X, info
    t_lit, t_X = r["literal"]["final_depth"]["rmse_best_99.5pct"], sp["cuda_vs_literal"]["final_depth"]["rmse_best_99.5pct"]
    assert t_lit <= 1.25 * t_X, (t_lit, t_X)
    # ... and nearer to the value both approximate than either of them
    assert d_wp < 1e-3 and d_wp < 0.55 * min(rm(sp["well_posed_vs_literal"]), rm(sp["cuda_vs_well_posed"])), info
    # volumes: the two evaluations of the reference agree on fewer voxels with each other than the default kernels do with the well-posed oracle
    assert sp["cuda_vs_literal"]["similarity_volume_levels"]["0"] < 0.85 < r["well_posed"]["similarity_volume_levels"]["0"]
    # attribution
    full = rm(dev["all"]["vs_literal"])
    assert abs(full - d_lit) < 0.15 * d_lit + 5e-5, info  # the default's four deviations switched on in the literal kernel: as far from the reference as the default kernels
    assert rm(dev["all"]["vs_well_posed"]) < 1e-3 and dev["all"]["vs_well_posed"]["similarity_volume_levels"]["0"] > 0.95, dev["all"]["vs_well_posed"]
    others = ("shifted_sums", "merged_exp", "homogeneous_v_rcp", "shared_R")
    if name == "crop3":
        # no image border in the tile: neither the exact-pixel border test of rounds 1-3 nor ... changes a bit
        assert abs(rm(dev["exact_border_r3"]["vs_literal"]) - floor) < 1e-9, info
        for k in others:
            assert rm(dev[k]["vs_literal"]) > 0.7 * full, (k, info)
    else:
        # the tile is the image: what rounds 1-3 carried on the knife-edge rows, and what is left of it
        assert rm(dev["exact_border_r3"]["vs_literal"]) > 2.0e-3, info
        assert d_lit < 0.6 * rm(dev["all_r3"]["vs_literal"]), info
    # the sums alone are what makes the VOLUMES differ: with shifted sums the literal kernel's volume is the well-posed oracle's
    assert dev["shifted_sums"]["vs_well_posed"]["similarity_volume_levels"]["0"] > 0.98
    assert dev["shifted_sums"]["vs_literal"]["similarity_volume_levels"]["0"] < 0.65
    # THE SIMILARITY MAP END TO END (BASELINE: "depth / sim maps"), as the program writes it — one half per pixel, mapIO.cpp:403-540.
    # Its yardstick is the reference against itself as well: after the colour optimisation the channel carries the smoothness angle between
    # neighbouring depths (mapKernels.cuh:604), which any two evaluations draw afresh (see _assert_reference_arithmetic).  The default kernels are
    # no further from the reference's arithmetic than its two evaluations are from each other; BEFORE the optimisation (the Refine stage's own
    # similarity) both distances are small and are printed for DESIGN.md's table.
    Xs, ds = sp["cuda_vs_literal"]["final_sim"], r["literal"]["final_sim"]
    print(name, "final_sim: reference vs itself", Xs, "default vs literal", ds, "| refined_sim: reference vs itself", sp["cuda_vs_literal"]["refined_sim"],
          "default vs literal", r["literal"]["refined_sim"])
    assert ds["rmse"] <= 1.15 * Xs["rmse"] and ds["p99_abs"] <= 1.15 * Xs["p99_abs"] + 1e-2, (ds, Xs)


def test_full_size_cfg2_sweep_against_the_oracle():
    """BASELINE configuration 2 at FULL size — 1920 x 1080, 128 planes, 4 T cameras, sweep only — against the oracle (which finishes this
    size in about a minute on the box's host cores): the similarity volumes of the whole frame (480 x 270 x 128 voxels, best and second
    best over the four T cameras) against the well-posed evaluation (tolerance class of test_similarity_volume_parity) and against the
    literal one (= the reference's own kernels), held to the literal evaluation's own distance from the well-posed one; and the
    winner-take-all plane of the un-aggregated volume (doSgmOptimizeVolume off)."""
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid
    from oracle import oracle
    sc = make_scene(5, 1920, 1080, seed=2, device="cuda")
    images_np = sc.images.cpu().numpy()
    sgm, ref = abi.SgmParams.default(), abi.RefineParams.default()
    depths = plane_depths(sc, 128)
    Z, tcs = len(depths), [1, 2, 3, 4]
    pyr = [DevicePyramid(sc.images[i], 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(5)]
    h = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref)
    h.run_sgm(0, tcs, depths, optimize=False, keep_raw=True)
    torch.cuda.synchronize()
    g_best, g_second = h.best_raw.cpu().numpy()[..., :Z], h.second.cpu().numpy()[..., :Z]
    g_wta = h.sgm_depth_sim.cpu().numpy()[..., 0]
    o = oracle.OracleDepthMap(images_np, sc.K, sc.R, sc.C, sgm, ref)
    with oracle.well_posed():
        o.run_sgm(0, tcs, depths, optimize=False)
    wp_second = o.second[..., :Z].copy()
    for got, want in ((g_best, o.best_raw[..., :Z]), (g_second, wp_second)):
        d = np.abs(want.astype(np.int16) - got.astype(np.int16))
        assert (d > 0).mean() <= 0.05, (d > 0).mean()
        assert (d > 1).mean() <= 2e-3, (d > 1).mean()
        assert ((want == 255) != (got == 255)).mean() <= 1e-3
    assert (o.sgm_depth_sim[..., 0] != g_wta).mean() < 0.05  # the raw volume's arg-min flips where two planes are within a level
    o.run_sgm(0, tcs, depths, optimize=False)  # literal: the reference's arithmetic
    inner = (slice(4, None), slice(4, None))  # knife-edge rows of the literal border test (test_similarity_volume_parity)
    floor, _ = level_mismatch(wp_second[inner], o.second[..., :Z][inner])
    frac, _ = level_mismatch(o.second[..., :Z][inner], g_second[inner])
    assert floor > 0.05 and frac <= 1.25 * floor + 0.03, (frac, floor)


def test_harness_batched_aggregation_equals_per_tile():
    """bench.py's multi-tile workloads sweep every tile, aggregate all their volumes with one avdm_volume_optimize_tiles call
    (pipeline.optimize_tiles_batched, what the C++ host does per group of tiles) and then finish each tile: same maps as tile by tile"""
    torch = _torch()
    from alicevision_amd.pipeline import DepthMapTile, DevicePyramid, optimize_tiles_batched
    sc, sgm, ref, depths = small_case(width=320, height=240, n_planes=40, seed=4)
    pyr = [DevicePyramid(sc.images[i].cuda(), 1, 128, abi.FILTER_CUDA_FIXED8) for i in range(3)]
    rois = [(0, 192, 0, 240), (128, 320, 0, 240)]
    seq = []
    for r in rois:
        t = DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=r)
        t.run_sgm(0, [1, 2], depths)
        seq.append(t.sgm_depth_sim.cpu().numpy().copy())
    tiles = [DepthMapTile(pyr, sc.K, sc.R, sc.C, sgm, ref, roi=r) for r in rois]
    for t in tiles:
        assert t.run_sgm(0, [1, 2], depths, optimize="defer") is None
    optimize_tiles_batched(tiles, 0)
    torch.cuda.synchronize()
    for t, want in zip(tiles, seq):
        assert np.array_equal(t.sgm_depth_sim.cpu().numpy(), want)


def test_bench_rccl_path_on_one_gpu():
    """The multi-GPU code path of bench.py — process group over RCCL (backend "nccl"), ViewExchange.setup()'s in-place all-gathers and, for the
    streaming job (--stream-views), the per-step all-gather of the freshly built R pyramids on a side stream and its commit — executed with ONE
    rank on the one GPU of the test box (`--force-dist`), so that the first 8-GPU run cannot die on API misuse.  By default the timed region
    runs NO collective (the pyramids are handed over once, at set-up); the lines must carry the same depth maps as the run without a process
    group, and the fixed-job figure next to the weak-scaling value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    outs = []
    for extra in (["--force-dist", "--stream-views"], ["--force-dist"], []):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "cfg1", "--steps", "3", "--warmup", "1",
                            "--no-cpu-baseline", "--cli-e2e", "0"] + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        outs.append(json.loads(lines[0]))
    streamed, forced, plain = outs
    assert streamed["config"]["process_group"].startswith("nccl") and forced["config"]["process_group"].startswith("nccl") and plain["config"]["process_group"] is None
    # cfg1 has 3 views: set-up = one all-gather per row of `world` = 1 views (3); the streaming job adds one per step (1 warm-up + 3 timed)
    assert streamed["config"]["pyramid_exchange_collectives"] == 3 + 4 and forced["config"]["pyramid_exchange_collectives"] == 3
    assert plain["config"]["pyramid_exchange_collectives"] == 0
    assert streamed["stages_ms"]["pyramid_exchange"] > 0.0 and "pyramid_commit" in streamed["stages_ms"]
    assert "pyramid_exchange" not in forced["stages_ms"] and "pyramid_exchange" not in plain["stages_ms"]
    assert streamed["valid_fraction"] == forced["valid_fraction"] == plain["valid_fraction"] > 0.5
    for o in outs:
        assert o["n_gpus"] == 1 and o["scaling"] == "weak"
        fj = o["fixed_job"]
        assert fj["cameras"] == 3 and fj["cameras_per_rank"] == [3] and abs(fj["makespan_s"] - 3 * fj["step_s_per_rank"][0]) < 1e-9


# ---- the switch matrix: every non-default AVDM_* code-path switch still passes its parity class ------------------------
# DESIGN.md section 4.5 leans on these switches as A/B references; a switch that rots silently would take its A/B with it.
_SIM_SWITCHES = [("AVDM_SIM_PLANE_PAIRS", "0"), ("AVDM_SIM_CHUNK_WINDOW", "0"), ("AVDM_SIM_PACKED", "0"), ("AVDM_SIM_PAIRED", "0"), ("AVDM_SIM_REC12", "0"),
                 ("AVDM_SIM_LDS", "0"), ("AVDM_SIM_STATS", "1"),
                 ("AVDM_SIM_PLANES8", "0"), ("AVDM_REFINE_PLANES8", "0"), ("AVDM_REFINE_OUTLIER_LIST", "0")]                                           # tolerance class (similarity arithmetic)
_EXACT_SWITCHES = [("AVDM_SGM_PAIR", "0"), ("AVDM_SGM_INT16", "0"), ("AVDM_SGM_PREPARE", "0"), ("AVDM_OPT_DEPTH_MAP_FORM", "1")]  # bit-exact class
_STATIC_SWITCHES = [("AVDM_SGM_P2_MAP", "legacy"), ("AVDM_SGM_TIMER", "record")]                                # read once per process: own process


def _switch_run(o, sc, sgm, ref, depths):
    """one tile through every stage on the ORACLE's pyramids; (second-best volume, aggregated volume, final map)"""
    torch = _torch()
    h = make_hip_from_oracle(o, sc, sgm, ref)
    h.run_sgm(0, [1, 2], depths, keep_raw=True)
    Z = len(depths)
    second, filtered = h.second.cpu().numpy()[..., :Z].copy(), h.best.cpu().numpy()[..., :Z].copy()
    final = h.run_refine(0, [1, 2]).cpu().numpy().copy()
    torch.cuda.synchronize()
    return second, filtered, final


@pytest.fixture(scope="module")
def switch_baseline(case):
    sc, sgm, ref, depths, o = case
    for var, _ in _SIM_SWITCHES + _EXACT_SWITCHES + _STATIC_SWITCHES:
        assert var not in os.environ, var + " is set in the test environment"
    return _switch_run(o, sc, sgm, ref, depths)


@pytest.mark.parametrize("var,val", _SIM_SWITCHES + _EXACT_SWITCHES)
def test_switch_matrix(case, switch_baseline, var, val):
    sc, sgm, ref, depths, o = case
    Z = len(depths)
    base_second, base_filtered, base_final = switch_baseline
    os.environ[var] = val
    try:
        second, filtered, final = _switch_run(o, sc, sgm, ref, depths)
    finally:
        os.environ.pop(var, None)
    if (var, val) in _EXACT_SWITCHES:
        # the switch selects another FORM of a bit-exact stage: the same bytes end to end
        assert np.array_equal(second, base_second) and np.array_equal(filtered, base_filtered), (var, level_mismatch(filtered, base_filtered))
        assert np.array_equal(final, base_final), var
        return
    # a switch of the similarity kernels: the tolerance class of test_similarity_volume_parity against the well-posed oracle ...
    want = o.second[..., :Z]
    frac, mx = level_mismatch(want, second)
    d = np.abs(want.astype(np.int16) - second.astype(np.int16))
    assert frac <= 0.03 and (d > 1).mean() <= 2e-3 and ((want == 255) != (second == 255)).mean() <= 2e-3, (var, frac, mx, float((d > 1).mean()))
    # ... the form actually changed something or is a pure instrumentation switch, and the depth map stays within the end-to-end class
    both = (final[..., 0] > 0) & (o.optimized[..., 0] > 0)
    assert both.mean() > 0.5 and ((final[..., 0] > 0) != (o.optimized[..., 0] > 0)).mean() < 5e-3
    err = np.sort((final[..., 0] - o.optimized[..., 0])[both] ** 2)
    assert np.sqrt(err[: int(0.995 * err.size)].mean()) < 1e-3, (var, float(np.sqrt(err.mean())))
    fracb, _ = level_mismatch(base_second, second)
    assert fracb < 0.02, (var, fracb)


@pytest.mark.parametrize("var,val", _STATIC_SWITCHES)
def test_switch_matrix_static(case, switch_baseline, var, val):
    """switches the library reads once per process: the same tile in a process of its own, compared by digest"""
    import hashlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, hashlib, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from alicevision_amd import abi\n"
            "from common import small_case, make_oracle\n"
            "from test_gpu_parity import _switch_run\n"
            "lib = abi.load(); lib.avdm_debug_sgm_kernel_timing.argtypes = [ctypes.c_int]; lib.avdm_debug_sgm_kernel_timing(1)\n"
            "sc, sgm, ref, depths = small_case(); o = make_oracle(sc, sgm, ref)\n"
            "print('DIGEST', ' '.join(hashlib.sha256(a.tobytes()).hexdigest() for a in _switch_run(o, sc, sgm, ref, depths)))\n") % (root, os.path.join(root, "tests"))
    env = dict(os.environ)
    env[var] = val
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    got = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0].split()[1:]
    assert got == [hashlib.sha256(a.tobytes()).hexdigest() for a in switch_baseline], var
