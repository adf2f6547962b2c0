"""A second, independent reading of an Alembic (Ogawa) archive for the tests: flattens every object / property into a dict
path -> (kind, pod, extent, metadata, [sample blobs with their 16-byte digests]) so that two archives can be compared entry by entry.
Written separately from alicevision_amd/host/alembic.cpp (Python, no shared code): test infrastructure only."""
import struct

ISDATA = 1 << 63


class Ogawa:
    def __init__(self, data):
        self.b = data
        assert data[:5] == b"Ogawa" and data[5] == 0xff, "not a closed Ogawa archive"
        self.root = struct.unpack("<Q", data[8:16])[0]

    def group(self, pos):
        if pos == 0:
            return []
        n = struct.unpack("<Q", self.b[pos:pos + 8])[0]
        return [((c & ISDATA) != 0, c & ~ISDATA) for c in struct.unpack("<%dQ" % n, self.b[pos + 8:pos + 8 + 8 * n])]

    def data(self, pos):
        if pos == 0:
            return b""
        n = struct.unpack("<Q", self.b[pos:pos + 8])[0]
        return self.b[pos + 8:pos + 8 + n]


class Archive:
    def __init__(self, data):
        self.o = Ogawa(data)
        k = self.o.group(self.o.root)
        self.library_version = struct.unpack("<i", self.o.data(k[1][1]))[0]
        self.top = k[2][1]
        self.archive_metadata = self.o.data(k[3][1]).decode()
        self.time_samplings = self.o.data(k[4][1])
        md = self.o.data(k[5][1])
        self.imeta = [""]
        i = 0
        while i < len(md):
            self.imeta.append(md[i + 1:i + 1 + md[i]].decode())
            i += 1 + md[i]

    def children(self, pos):
        k = self.o.group(pos)
        if not k:
            return 0, []
        hdr = self.o.data(k[-1][1])[:-32]
        out, i, ci = [], 0, 1
        while i < len(hdr):
            n = struct.unpack("<I", hdr[i:i + 4])[0]
            name = hdr[i + 4:i + 4 + n].decode()
            i += 4 + n
            mi = hdr[i]
            i += 1
            if mi == 0xff:
                m = struct.unpack("<I", hdr[i:i + 4])[0]
                meta = hdr[i + 4:i + 4 + m].decode()
                i += 4 + m
            else:
                meta = self.imeta[mi]
            out.append((name, meta, k[ci][1]))
            ci += 1
        return k[0][1], out

    def properties(self, pos):
        k = self.o.group(pos)
        if not k:
            return []
        hdr = self.o.data(k[-1][1])
        out, i, ci = [], 0, 0

        def rd(sz):
            nonlocal i
            fmt, n = (("<B", 1), ("<H", 2), ("<I", 4))[sz]
            v = struct.unpack(fmt, hdr[i:i + n])[0]
            i += n
            return v

        while i < len(hdr):
            info = struct.unpack("<I", hdr[i:i + 4])[0]
            i += 4
            kind, sh = info & 3, (info >> 2) & 3
            h = {"kind": kind, "info": info}
            if kind:
                h["pod"], h["extent"], h["next"] = (info >> 4) & 0xf, (info >> 12) & 0xff, rd(sh)
                if info & 0x200:
                    rd(sh), rd(sh)
                if info & 0x100:
                    rd(sh)
            n = rd(sh)
            h["name"] = hdr[i:i + n].decode()
            i += n
            if (info >> 20) & 0xff == 0xff:
                m = rd(sh)
                h["meta"] = hdr[i:i + m].decode()
                i += m
            else:
                h["meta"] = self.imeta[(info >> 20) & 0xff]
            h["pos"] = k[ci][1]
            ci += 1
            out.append(h)
        return out


def flatten(data):
    a = Archive(data)
    out = {}

    def props(pos, path):
        for h in a.properties(pos):
            if h["kind"] == 0:
                out[path + "/" + h["name"]] = ("compound", h["meta"])
                props(h["pos"], path + "/" + h["name"])
            else:
                blobs = [a.o.data(c[1]) if c[1] else b"" for c in a.o.group(h["pos"])]
                # the header bit field without the INDEX of the metadata text (a position in each archive's own table; the text follows)
                out[path + "/" + h["name"]] = (h["kind"], h["pod"], h["extent"], h["info"] & ~(0xff << 20), h["meta"], blobs)

    def obj(pos, path):
        p, ch = a.children(pos)
        if p:
            props(p, path + ":")
        for name, meta, cpos in ch:
            out[path + "/" + name] = ("object", meta)
            obj(cpos, path + "/" + name)

    obj(a.top, "")
    out["<library version>"] = a.library_version
    out["<time samplings>"] = a.time_samplings
    return out
