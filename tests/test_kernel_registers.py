"""Register budgets of the kernels whose occupancy the measured rates depend on, read from the metadata of the BUILT objects
(llvm-objdump --offloading + llvm-readelf --notes): a source edit that pushes one of them over its step of the VGPR ladder (or into
scratch) costs waves per SIMD without failing any parity test — this one fails instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(obj, tmp_path):
    src = os.path.join(ROOT, "cosdata_amd", "csrc", obj)
    if not os.path.exists(src) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("built objects / llvm tools not present")
    work = str(tmp_path)
    shutil.copy(src, work)                                    # the bundles are extracted next to the object: never into the tree
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(work) if "amdgcn" in f and "gfx950" in f]
    assert dev, os.listdir(work)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev[0]], cwd=work, check=True, capture_output=True, text=True).stdout
    out, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s+\.name:\s+(\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            out[name] = {}
        m = re.match(r"\s+\.(vgpr_count|private_segment_fixed_size|sgpr_count):\s+(\d+)", line)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


def _find(ks, *needles):
    hits = [k for k in ks if all(n in k for n in needles)]
    assert len(hits) == 1, (needles, hits)
    return ks[hits[0]]


def test_walk_and_finalize_kernels_keep_their_occupancy(tmp_path):
    ks = _kernels("kernels_walk.o", tmp_path)
    head = _find(ks, "walk_kernel<0, 1, 1, true, false, 8>")           # c2 headline: u8, 513..1024 dims, ef <= 64, reference filter
    assert head["vgpr_count"] <= 72 and head["private_segment_fixed_size"] == 0      # 7 waves per SIMD, nothing in scratch
    ef256 = _find(ks, "walk_kernel<0, 1, 4, true, false, 8>")
    assert ef256["vgpr_count"] <= 96                                                 # 5 waves per SIMD
    ef512 = _find(ks, "walk_kernel<0, 1, 8, true, false, 4>")          # ef > 256 takes four row buffers (walk_pb_policy)
    assert ef512["vgpr_count"] <= 96 and ef512["private_segment_fixed_size"] == 0    # 5 waves per SIMD (eight buffers: 104 VGPRs = 4 waves)
    ef1024 = _find(ks, "walk_kernel<0, 1, 16, true, false, 4>")        # round 5: ef up to 1024, a pool of 64 x 16 keys per wave
    assert ef1024["vgpr_count"] <= 128 and ef1024["private_segment_fixed_size"] == 0  # 4 waves per SIMD, nothing in scratch
    upper256 = _find(ks, "walk_kernel<0, 1, 4, true, false, 4>")       # the upper range of a split walk above ef 64 (four row buffers)
    assert upper256["vgpr_count"] <= 96 and upper256["private_segment_fixed_size"] == 0
    exact = _find(ks, "walk_kernel<0, 1, 1, true, true, 8>")
    assert exact["vgpr_count"] <= 80
    fin = _find(ks, "finalize_fast_kernel<1>")                          # big launches: one wave per query
    assert fin["vgpr_count"] <= 72 and fin["private_segment_fixed_size"] == 0        # 7 waves per SIMD (the general kernel: 121 VGPRs = 4)
    wide = _find(ks, "finalize_fast_kernel<8>")                         # small launches: eight waves per query, one workgroup per CU
    assert wide["vgpr_count"] <= 128 and wide["private_segment_fixed_size"] == 0


def _disassembly(obj, tmp_path, symbol_part):
    src = os.path.join(ROOT, "cosdata_amd", "csrc", obj)
    if not os.path.exists(src) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("built objects / llvm tools not present")
    work = str(tmp_path)
    shutil.copy(src, work)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dev = [f for f in os.listdir(work) if "amdgcn" in f and "gfx950" in f]
    text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", dev[0]], cwd=work, check=True, capture_output=True, text=True).stdout
    body, on = [], False
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            on = symbol_part in m.group(1)
            continue
        if on:
            ins = line.split("//")[0].strip()
            if ins:
                body.append(ins)
    assert body, symbol_part
    return body


def test_packed_sparse_kernel_instruction_budget(tmp_path):
    """the packed posting layout exists to cut the instructions a posting costs: 60 VGPRs, nothing in scratch, and a
    full step's eight postings apply with at most six vector instructions between two LDS adds (subtract, two 24-bit multiplies,
    min, or — plus a compare and a select in the variant with a first visited key), none of them the quarter-rate 32-bit multiply"""
    ks = _kernels("kernels_sparse.o", tmp_path)
    pk = _find(ks, "sparse_packed_kernel<8>")
    assert pk["vgpr_count"] <= 64 and pk["private_segment_fixed_size"] == 0
    body = _disassembly("kernels_sparse.o", tmp_path, "sparse_packed_kernelILi8E")
    adds = [i for i, ins in enumerate(body) if ins.startswith("ds_add_u32")]
    best = None
    for a in range(len(adds) - 7):                               # eight consecutive adds with no branch in between = one full step
        seg = body[adds[a]:adds[a + 7] + 1]
        if any(x.startswith(("s_cbranch", "s_branch")) for x in seg):
            continue
        valu = sum(1 for x in seg if x.startswith("v_"))
        best = valu if best is None else min(best, valu)
        assert not any(x.startswith("v_mul_lo_u32") for x in seg)
    assert best is not None and best <= 7 * 6, best               # seven gaps between eight adds
