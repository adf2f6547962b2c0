"""Is the machine code of the similarity kernels the same as at another revision of csrc/avdm_similarity.hip?

The PMC summaries bench.py quotes (profiles/r*_sim_pmc.json) are stamped with the sha256 of the kernel SOURCE they were measured with.  A change
that only ADDS instantiations or switches (default off) changes that hash and nothing the counters describe.  This script compiles the file at
<rev> and in the working tree to gfx950 assembly (`hipcc -S --cuda-device-only`, the flags of alicevision_amd/build.py), and compares every
kernel that exists at <rev> instruction by instruction (labels renumbered; a template parameter appended with its default value is ignored in the
name).  Exit status 0 = every one of them is identical.

    python scripts/isa_identity.py <rev> [--certify profiles/r04_sim_pmc.json] [--out profiles/r04_isa_identity.txt]

--certify appends the working tree's sha256 to the summary's "isa_identical_sources" (bench.py accepts those) when — and only when — the
comparison holds.  No GPU needed.
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "alicevision_amd/csrc/avdm_similarity.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def assemble(src_dir, out):
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(src_dir, "avdm_similarity.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("hipcc failed:\n" + r.stderr[-2000:])


def kernels(path):
    """{mangled name: [instruction lines, labels renumbered]}"""
    lines = open(path).read().split("\n")
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(_ZN4avdm\w+):", lines[i])
        if m:
            body = []
            i += 1
            while i < len(lines) and "s_endpgm" not in lines[i]:
                ln = lines[i]
                if (ln.startswith("\t") and not ln.strip().startswith((";", "."))) or re.match(r"^\.LBB", ln):
                    body.append(re.sub(r"\.LBB\d+_", ".LBB_", ln.split(";")[0].rstrip()))
                i += 1
            out[m.group(1)] = body
        i += 1
    return out


def demangle(name):
    """readable kernel name without its argument list (c++filt when there is one)"""
    exe = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if exe is None:
        return name[:100]
    return subprocess.run([exe, name], capture_output=True, text=True).stdout.strip().split("(")[0]


def strip_default_tail(name):
    # hook for a template parameter appended with a default value between the two revisions (round 4: `bool DEINT = false`, removed again in
    # round 5 together with MODE): none at present, the names compare as they are
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("rev")
    ap.add_argument("--certify")
    ap.add_argument("--out")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="isa_identity_")
    try:
        # the sources include "../../include/avdm.h": keep the depth of the tree
        old_dir = os.path.join(tmp, "old", "alicevision_amd", "csrc")
        os.makedirs(old_dir)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "old", "include"))
        listing = subprocess.run(["git", "-C", ROOT, "ls-tree", "--name-only", a.rev, "alicevision_amd/csrc/"], capture_output=True, text=True, check=True).stdout.split()
        for f in listing:
            if f.endswith((".h", ".hip")):
                blob = subprocess.run(["git", "-C", ROOT, "show", "%s:%s" % (a.rev, f)], capture_output=True, check=True).stdout
                open(os.path.join(old_dir, os.path.basename(f)), "wb").write(blob)
        old_sha = hashlib.sha256(open(os.path.join(old_dir, "avdm_similarity.hip"), "rb").read()).hexdigest()
        new_sha = hashlib.sha256(open(os.path.join(ROOT, SRC), "rb").read()).hexdigest()
        assemble(old_dir, os.path.join(tmp, "old.s"))
        assemble(os.path.join(ROOT, "alicevision_amd", "csrc"), os.path.join(tmp, "new.s"))
        old, new = kernels(os.path.join(tmp, "old.s")), {strip_default_tail(k): v for k, v in kernels(os.path.join(tmp, "new.s")).items()}
        rows, bad = [], 0
        for name, body in sorted(old.items()):
            got = new.get(name)
            same = got == body
            bad += 0 if same else 1
            short = demangle(name)
            rows.append("%-9s %6d instructions  %s" % ("identical" if same else ("MISSING" if got is None else "DIFFERENT"), len(body), short))
        added = sorted(set(new) - set(old))
        rev_full = subprocess.run(["git", "-C", ROOT, "rev-parse", a.rev], capture_output=True, text=True, check=True).stdout.strip()
        text = ["# scripts/isa_identity.py %s: csrc/avdm_similarity.hip at that revision against the working tree, gfx950 assembly of every kernel of the revision" % a.rev,
                "# revision %s  source sha256 %s" % (rev_full, old_sha), "# working tree%s source sha256 %s" % (" " * 29, new_sha),
                "# flags: %s" % " ".join(FLAGS), ""] + rows + ["", "%d of %d kernels identical; %d kernels only in the working tree:" % (len(old) - bad, len(old), len(added))]
        for k in added:
            text.append("    " + demangle(k))
        text = "\n".join(text) + "\n"
        print(text)
        if a.out:
            open(os.path.join(ROOT, a.out), "w").write(text)
        if bad == 0 and a.certify:
            path = os.path.join(ROOT, a.certify)
            rec = json.load(open(path))
            if rec.get("kernel_source_sha256") != old_sha:
                sys.exit("%s was not measured with the source at %s" % (a.certify, a.rev))
            lst = [e for e in rec.get("isa_identical_sources", []) if e.get("sha256") != new_sha]
            lst.append({"sha256": new_sha, "evidence": a.out or "scripts/isa_identity.py " + a.rev,
                        "note": "every kernel of the measured source compiles to the same gfx950 instructions from this one (labels renumbered)"})
            rec["isa_identical_sources"] = lst
            json.dump(rec, open(path, "w"), indent=1)
            print("certified", new_sha, "in", a.certify)
        sys.exit(0 if bad == 0 else 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
