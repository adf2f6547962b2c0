"""oracle/_ref build step: make a reference .cu file compilable by g++.

The ONLY thing g++ cannot parse in the reference's kernel-launch layer is the execution-configuration syntax
`kernel<<<grid, block[, shmem, stream]>>>(args...)`.  This script reads a reference source where it lies under /root/reference and
writes a build intermediate (into oracle/_ref/gen/, git-ignored) in which every such launch reads
`::shim::launch(grid, block, kernel, args...)` (oracle/ref/shim/cuda_runtime.h) — nothing else is touched.  Reference sources are
never copied into the repository.

    python gen_launches.py <in.cu> <out.cpp>
"""
import re
import sys

LAUNCH = re.compile(r"([A-Za-z_][A-Za-z_0-9]*(?:<[^<>;]*>)?)\s*<<<\s*([^,<>]+?)\s*,\s*([^,<>]+?)\s*(?:,[^<>]*)?>>>\s*\(")


def main(src, dst):
    text = open(src, encoding="utf-8-sig").read()
    out, n = LAUNCH.subn(lambda m: "::shim::launch(%s, %s, %s, " % (m.group(2), m.group(3), m.group(1)), text)
    if "<<<" in out:
        raise SystemExit("%s: a kernel launch was not rewritten" % src)
    with open(dst, "w") as f:
        f.write('#line 1 "%s"\n' % src)
        f.write(out)
    print("%s: %d launches rewritten -> %s" % (src, n, dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
