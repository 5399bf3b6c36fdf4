#!/usr/bin/env python3
"""Did an edit change the code of kernels it was not meant to touch?  Compiles one translation unit of cosdata_amd/csrc at a git revision
and in the working tree to gfx950 assembly (device side only) and compares every kernel instruction by instruction (labels normalised,
comments dropped).  Used at the end of round 4, when no device was at hand, to keep the measured walk kernels byte for byte while their
source moved into walk_kernel.inc and a candidate variant was added next to them.

usage: compare_kernel_isa.py <git revision> <file under cosdata_amd/csrc, e.g. kernels_walk.hip>     exit status 1 if a kernel changed"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cosdata_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fhip-fp32-correctly-rounded-divide-sqrt", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-inline-asm", "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include")]


def kernels(asm_path):
    out, name, buf = {}, None, []
    for line in open(asm_path):
        m = re.match(r"^(_Z\S+):\s", line)
        if m:
            if name:
                out[name] = buf
            name, buf = m.group(1), []
            continue
        if name is not None:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                out[name], name, buf = buf, None, []
                continue
            ins = line.split(";")[0].rstrip()
            if ins.strip():
                buf.append(re.sub(r"\.LBB\d+_", ".LBB_", ins))
    return out


def main():
    rev, unit = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        old_dir = os.path.join(tmp, "old")
        os.makedirs(old_dir)
        # the whole source directory of the revision (and the public header some units reach by a relative path): a unit includes its neighbours
        tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "cosdata_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", old_dir], input=tar, check=True)
        asm = {}
        for tag, src_dir in (("old", os.path.join(old_dir, "cosdata_amd", "csrc")), ("new", CSRC)):
            asm[tag] = os.path.join(tmp, tag + ".s")
            extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if unit == "kernels_scan.hip" else []
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-I", src_dir, os.path.join(src_dir, unit), "-o", asm[tag]], check=True, stderr=subprocess.DEVNULL)
        a, b = kernels(asm["old"]), kernels(asm["new"])
    changed = [k for k in a if k in b and a[k] != b[k]]
    gone, added = [k for k in a if k not in b], [k for k in b if k not in a]
    demangle = lambda k: subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    print(f"{unit}: {len(a)} kernels at {rev}, {len(b)} now; identical {len(a) - len(changed) - len(gone)}, changed {len(changed)}, gone {len(gone)}, new {len(added)}")
    for k in changed:
        print("  changed:", demangle(k)[:140], f"({len(a[k])} -> {len(b[k])} instructions)")
    for k in gone:
        print("  gone:   ", demangle(k)[:140])
    for k in added:
        print("  new:    ", demangle(k)[:140])
    return 1 if changed or gone else 0


if __name__ == "__main__":
    sys.exit(main())
