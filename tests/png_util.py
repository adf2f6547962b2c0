"""Minimal PNG reader / writer (zlib + struct) for the tests of the host PNG codec: 8-bit, non-interlaced, any scan-line filter."""
import struct
import zlib

import numpy as np


def _chunk(t, payload):
    return struct.pack(">I", len(payload)) + t + payload + struct.pack(">I", zlib.crc32(t + payload) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def write_png(path, img, filters=None, idat_split=None):
    """img: (h, w) uint8 greyscale or (h, w, 3) RGB.  filters: per-row filter type list (default 0); idat_split: bytes per IDAT chunk."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = 0 if filters is None else filters[y % len(filters)]
        cur = rows[y]
        out = np.zeros_like(cur)
        for i in range(w * ch):
            a = cur[i - ch] if i >= ch else 0
            b = prev[i]
            c = prev[i - ch] if i >= ch else 0
            pred = [0, a, b, (a + b) // 2, _paeth(int(a), int(b), int(c))][ft]
            out[i] = (cur[i] - pred) & 255
        raw.append(ft)
        raw.extend(out.astype(np.uint8).tobytes())
        prev = cur
    z = zlib.compress(bytes(raw), 6)
    parts = [z] if not idat_split else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 2: 4, 3: 2, 4: 6}[ch], 0, 0, 0)))
        f.write(_chunk(b"tEXt", b"Comment\x00test"))
        for p in parts:
            f.write(_chunk(b"IDAT", p))
        f.write(_chunk(b"IEND", b""))


def read_png_gray8(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, z, hdr = 8, b"", None
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        t = data[pos + 4:pos + 8]
        payload = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(t + payload) & 0xFFFFFFFF
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", payload)
        elif t == b"IDAT":
            z += payload
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and ctype == 0 and interlace == 0
    raw = zlib.decompress(z)
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.int32)
    for y in range(h):
        ft = raw[y * (w + 1)]
        line = np.frombuffer(raw[y * (w + 1) + 1:(y + 1) * (w + 1)], np.uint8).astype(np.int32)
        cur = np.zeros(w, np.int32)
        for i in range(w):
            a = cur[i - 1] if i else 0
            b = prev[i]
            c = prev[i - 1] if i else 0
            pred = [0, a, b, (a + b) // 2, _paeth(int(a), int(b), int(c))][ft]
            cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out
