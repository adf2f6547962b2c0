"""OpenEXR scan-line files in numpy + zlib (no OpenEXR / OpenImageIO in this image).

An independent implementation of the same published file layout as alicevision_amd/host/exr.cpp: the synthetic-scene tools
write the input images with it and the tests read the C++ host's depth / similarity maps back with it, so each side checks
the other.  Supported: single-part scan-line files, HALF / FLOAT / UINT channels, compression NONE / ZIPS / ZIP.
"""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PT_DTYPE = {0: np.uint32, 1: np.float16, 2: np.float32}
_LINES = {0: 1, 2: 1, 3: 16}


def _attr(name, typ, data):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _zip_block(raw):
    a = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int16)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    comp = zlib.compress((d & 0xff).astype(np.uint8).tobytes(), 4)
    return comp if len(comp) < len(raw) else raw


def _unzip_block(comp, raw_size):
    if len(comp) == raw_size:
        return comp
    d = np.frombuffer(zlib.decompress(comp), dtype=np.uint8).astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) & 0xff).astype(np.uint8)
    half = (raw_size + 1) // 2
    out = np.empty(raw_size, dtype=np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def write_exr(path, channels, attributes=None, half=False, data_origin=(0, 0), display_size=None, compression=3):
    """channels: dict name -> 2-D array (all the same shape).  attributes: dict name -> (exr type string, raw bytes) or
    python values (int -> "int", float -> "float", str -> "string")."""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    dt = np.float16 if half else np.float32
    planes = [np.ascontiguousarray(channels[n], dtype=dt) for n in names]
    x0, y0 = data_origin
    dw, dh = display_size if display_size else (w, h)
    head = struct.pack("<ii", _MAGIC, 2)
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1 if half else 2, 0, 1, 1) for n in names) + b"\0"
    head += _attr("channels", "chlist", chl)
    head += _attr("compression", "compression", bytes([compression]))
    head += _attr("dataWindow", "box2i", struct.pack("<4i", x0, y0, x0 + w - 1, y0 + h - 1))
    head += _attr("displayWindow", "box2i", struct.pack("<4i", 0, 0, dw - 1, dh - 1))
    head += _attr("lineOrder", "lineOrder", b"\0")
    head += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    for k, v in (attributes or {}).items():
        if isinstance(v, tuple):
            head += _attr(k, v[0], v[1])
        elif isinstance(v, (int, np.integer)):
            head += _attr(k, "int", struct.pack("<i", int(v)))
        elif isinstance(v, float):
            head += _attr(k, "float", struct.pack("<f", v))
        else:
            head += _attr(k, "string", str(v).encode())
    head += b"\0"
    lines = _LINES[compression]
    blocks = []
    for b0 in range(0, h, lines):
        b1 = min(b0 + lines, h)
        raw = b"".join(p[y].tobytes() for y in range(b0, b1) for p in planes)
        blocks.append(raw if compression == 0 else _zip_block(raw))
    off = len(head) + 8 * len(blocks)
    table = b""
    for blk in blocks:
        table += struct.pack("<Q", off)
        off += 8 + len(blk)
    with open(path, "wb") as f:
        f.write(head)
        f.write(table)
        for i, blk in enumerate(blocks):
            f.write(struct.pack("<ii", y0 + i * lines, len(blk)))
            f.write(blk)


def m44d(values):
    return ("m44d", struct.pack("<16d", *[float(v) for v in values]))


def read_exr(path, header_only=False):
    """returns (channels dict name -> float32 array, info dict with 'attributes' (name -> (type, bytes)), 'data_window',
    'display_window', 'channel_types')"""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    assert magic == _MAGIC, "not an OpenEXR file"
    assert (version & 0xff) == 2 and not (version & 0x1a00), "only single-part scan-line files"
    p = 8
    attrs = {}

    def cstr(p):
        e = buf.index(b"\0", p)
        return buf[p:e].decode(), e + 1

    while buf[p] != 0:
        name, p = cstr(p)
        typ, p = cstr(p)
        (size,) = struct.unpack_from("<i", buf, p)
        p += 4
        attrs[name] = (typ, buf[p:p + size])
        p += size
    p += 1
    chans = []
    c = attrs["channels"][1]
    q = 0
    while c[q] != 0:
        e = c.index(b"\0", q)
        name = c[q:e].decode()
        pt, _, xs, ys = struct.unpack_from("<iB3xii", c, e + 1)
        assert xs == 1 and ys == 1
        chans.append((name, pt))
        q = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    info = {"attributes": {k: v for k, v in attrs.items()}, "data_window": (x0, y0, x1, y1),
            "display_window": struct.unpack("<4i", attrs["displayWindow"][1]), "channel_types": dict(chans)}
    if header_only:
        return None, info
    w, h = x1 - x0 + 1, y1 - y0 + 1
    comp = attrs["compression"][1][0]
    assert comp in _LINES, "unsupported compression %d" % comp
    lines = _LINES[comp]
    nb = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % nb, buf, p)
    bpl = sum(w * np.dtype(_PT_DTYPE[pt]).itemsize for _, pt in chans)
    out = {n: np.empty((h, w), dtype=np.float32) for n, _ in chans}
    for off in offsets:
        y, sz = struct.unpack_from("<ii", buf, off)
        l0 = y - y0
        nl = min(lines, h - l0)
        raw = _unzip_block(buf[off + 8:off + 8 + sz], bpl * nl) if comp else buf[off + 8:off + 8 + sz]
        q = 0
        for l in range(nl):
            for n, pt in chans:
                dt = np.dtype(_PT_DTYPE[pt])
                out[n][l0 + l] = np.frombuffer(raw, dtype=dt, count=w, offset=q).astype(np.float32)
                q += w * dt.itemsize
    return out, info


def attr_value(info, name):
    """decode int / float / string / m44d / m33d / v3d attributes"""
    typ, data = info["attributes"][name]
    if typ == "int":
        return struct.unpack("<i", data)[0]
    if typ == "float":
        return struct.unpack("<f", data)[0]
    if typ == "string":
        return data.decode()
    if typ in ("m44d", "m33d", "v3d"):
        return np.frombuffer(data, dtype=np.float64).copy()
    return data
