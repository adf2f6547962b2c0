"""Reading the dump of the host's JPEG entropy decoder (`avdm_host_tool jpeg-dump`, host/jpeg.cpp) for the tests and the ctypes harness:
geometry, quantisation tables and the quantised DCT coefficients of every component."""
import os
import struct
import subprocess
import tempfile

import numpy as np

from . import abi

TOOL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "avdm_host_tool")


class JpegCoefficients:
    def __init__(self, blob):
        (self.width, self.height, nc, self.hmax, self.vmax, rgb, prog, self.exif_orientation) = struct.unpack("<8i", blob[:32])
        self.stored_as_rgb, self.progressive = bool(rgb), bool(prog)
        self.components = []
        o = 32
        for _ in range(nc):
            h, v, bw, bh, w, hh = struct.unpack("<6i", blob[o:o + 24])
            o += 24
            quant = np.frombuffer(blob[o:o + 128], np.uint16).copy()
            o += 128
            n = bw * bh * 64
            coef = np.frombuffer(blob[o:o + 2 * n], np.int16).copy().reshape(bh, bw, 64)
            o += 2 * n
            self.components.append(dict(h=h, v=v, blocks_w=bw, blocks_h=bh, width=w, height=hh, quant=quant, coef=coef))

    def descriptors(self, pointers):
        """avdm_jpeg_component_t[3] over the given coefficient addresses (host arrays for the oracle, device tensors for the library)"""
        comps = (abi.JpegComponent * 3)()
        for i, (c, ptr) in enumerate(zip(self.components, pointers)):
            comps[i].coef = ptr
            comps[i].blocks_w, comps[i].blocks_h, comps[i].width, comps[i].height = c["blocks_w"], c["blocks_h"], c["width"], c["height"]
            comps[i].h_samp, comps[i].v_samp = c["h"], c["v"]
            for k in range(64):
                comps[i].quant[k] = int(c["quant"][k])
        return comps


def read_coefficients(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "coefs.bin")
        r = subprocess.run([TOOL, "jpeg-dump", path, out], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr.strip())
        with open(out, "rb") as f:
            return JpegCoefficients(f.read())
