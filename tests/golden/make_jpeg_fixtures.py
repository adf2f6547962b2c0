"""Golden vectors of the JPEG path: small JPEG files written by Pillow (libjpeg-turbo) and the pixels the SAME library decodes from them.

Run anywhere Pillow is installed:  python tests/golden/make_jpeg_fixtures.py

The reference reads photographs through OpenImageIO, whose JPEG reader is libjpeg / libjpeg-turbo with its defaults (accurate integer
inverse DCT, "fancy" chroma up-sampling); the library is not part of /root/reference, so parity is anchored on vectors decoded by it:
Pillow's decoder is the same library with the same defaults.  tests/golden/jpeg/ holds, per case, <name>.jpg and the expected 8-bit
RGB (grey replicated) in expected.npz; the cases cover 4:4:4 / 4:2:2 / 4:2:0, baseline and progressive scans, restart markers,
optimised Huffman tables, grey, RGB-stored (Adobe transform 0), odd sizes down to 1 x 1 and quality 1 ... 100.
tests/test_oracle.py::test_jpeg_* decode the files with the host's entropy decoder + the oracle and compare; the -m gpu test compares the
device with the oracle on the same files and on a 12 MP image made at run time."""
import os

import numpy as np
import PIL
from PIL import Image, features

DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg")


def picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 5.0), 128 + 90 * np.cos(x / 3.0 + y / 9.0), 60 + x * 150.0 / max(w, 1) + 20 * np.sin(y / 2.0)], -1)
    a += rng.normal(0, 12, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


CASES = [
    ("444_q90", (67, 45), "RGB", dict(quality=90, subsampling=0)),
    ("422_q75", (67, 45), "RGB", dict(quality=75, subsampling=1)),
    ("420_q75", (67, 45), "RGB", dict(quality=75, subsampling=2)),
    ("420_q10_even", (64, 48), "RGB", dict(quality=10, subsampling=2)),
    ("420_progressive", (131, 77), "RGB", dict(quality=80, subsampling=2, progressive=True)),
    ("422_progressive_restart", (90, 41), "RGB", dict(quality=85, subsampling=1, progressive=True, restart_marker_blocks=4)),
    ("420_restart_rows", (100, 60), "RGB", dict(quality=80, subsampling=2, restart_marker_rows=1)),
    ("444_optimized", (50, 50), "RGB", dict(quality=60, subsampling=0, optimize=True)),
    ("grey_q80", (53, 31), "L", dict(quality=80)),
    ("grey_progressive", (40, 40), "L", dict(quality=50, progressive=True)),
    ("rgb_stored", (33, 35), "RGB", dict(quality=90, keep_rgb=True)),
    ("420_1x1", (1, 1), "RGB", dict(quality=90, subsampling=2)),
    ("420_2x2", (2, 2), "RGB", dict(quality=90, subsampling=2)),
    ("422_3x5", (3, 5), "RGB", dict(quality=90, subsampling=1)),
    ("420_17x8", (17, 8), "RGB", dict(quality=95, subsampling=2)),
    ("420_q100", (48, 32), "RGB", dict(quality=100, subsampling=2)),
    ("420_q1", (48, 32), "RGB", dict(quality=1, subsampling=2)),
]


def main():
    os.makedirs(DST, exist_ok=True)
    expected = {}
    for k, (name, (w, h), mode, kw) in enumerate(CASES):
        a = picture(w, h, 100 + k)
        path = os.path.join(DST, name + ".jpg")
        Image.fromarray(a if mode == "RGB" else a[..., 0]).save(path, **kw)
        ref = np.array(Image.open(path))
        expected[name] = ref if ref.ndim == 3 else np.stack([ref] * 3, -1)
    np.savez_compressed(os.path.join(DST, "expected.npz"), **expected)
    with open(os.path.join(DST, "README.txt"), "w") as f:
        f.write("written and decoded by Pillow %s (libjpeg API %s, libjpeg-turbo: %s) through tests/golden/make_jpeg_fixtures.py\n" %
                (PIL.__version__, features.version("jpg"), features.check_feature("libjpeg_turbo")))
    print("wrote", len(CASES), "cases,", sum(os.path.getsize(os.path.join(DST, f)) for f in os.listdir(DST)), "bytes")


if __name__ == "__main__":
    main()
