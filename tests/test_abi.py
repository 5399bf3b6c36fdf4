"""The C-ABI library loads, exports every symbol include/cosdata_hip.h declares, and keeps its error
contract without a GPU (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cosdata_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cos_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from cosdata_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.ABI_SYMBOLS) == declared  # the ctypes table tracks the header


def test_params_struct_layout_matches_header(tmp_path):
    from cosdata_amd._lib import CosParams, CosSearchStats
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "cosdata_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu\\n", sizeof(cos_params), offsetof(cos_params, seed), '
                   'offsetof(cos_params, device), sizeof(cos_search_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sz, off_seed, off_dev, sz_stats = map(int, subprocess.check_output([str(exe)]).split())
    assert C.sizeof(CosParams) == sz == 80
    assert CosParams.seed.offset == off_seed and CosParams.device.offset == off_dev
    assert C.sizeof(CosSearchStats) == sz_stats


def test_cxx_host_links_and_error_contract(tmp_path):
    exe = tmp_path / "abi_smoke"
    so_dir = os.path.join(ROOT, "cosdata_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "abi_smoke.cpp"),
                           "-L", so_dir, "-lcosdata_hip", f"-Wl,-rpath,{so_dir}", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")


def test_no_cpu_fallback_without_device(gpu_available):
    import numpy as np
    import cosdata_amd as ca
    if gpu_available:
        pytest.skip("a GPU is present: the no-device contract is checked on CPU-only boxes")
    with pytest.raises(ca.CosdataError) as ei:
        ca.HNSWIndex(64)
    assert ei.value.status == 7  # NoDevice
    with pytest.raises(ca.CosdataError):
        ca.ScalarQuantization.quantize(np.zeros((2, 64), np.float32), ca.StorageType.UnsignedByte(), (-1.0, 1.0))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cosdata_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                code = "\n".join(l for l in src.splitlines() if not l.strip().startswith(("//", "#", "*", "/*")))
                assert "import oracle" not in code and "from oracle" not in code and "libcosdata_oracle" not in code, f


def test_scan_kernel_is_built_in_vgpr_form():
    """kernels_scan.hip relies on -mllvm -amdgpu-mfma-vgpr-form=1 (accumulators in VGPRs, query fragments read straight from
    AccVGPRs): without it the kernel is still correct but ~2x slower, so losing the flag in a Makefile edit must fail loudly."""
    mk = open(os.path.join(ROOT, "cosdata_amd", "csrc", "Makefile")).read()
    assert "kernels_scan.o: CXXFLAGS += -mllvm -amdgpu-mfma-vgpr-form=1" in mk
    assert "kernels_scan.hip" in mk


def test_python_mirror_of_header_constants():
    """constants the Python side repeats must equal the header's"""
    import cosdata_amd as ca
    hdr = open(os.path.join(ROOT, "include", "cosdata_hip.h")).read()
    m = re.search(r"#define\s+COS_LATENCY_MODE_DEFAULT_MAX_B\s+(\d+)u", hdr)
    assert m and int(m.group(1)) == ca.HNSWIndex.LATENCY_MODE_DEFAULT_MAX_B
    m = re.search(r"#define\s+COS_LATENCY_WAVES_DEFAULT_MAX_B\s+(\d+)u", hdr)
    assert m and int(m.group(1)) == ca.HNSWIndex.LATENCY_WAVES_DEFAULT_MAX_B
    m = re.search(r"#define\s+COS_WALK_ORDER_DEFAULT_MIN_B\s+(\d+)u", hdr)
    assert m and int(m.group(1)) == ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B


def test_tuning_registry_set_get_clear_and_the_one_environment_variable():
    """cos_tuning_* (cosdata_amd/csrc/tuning.h): the library's experiment knobs live in ONE registry; the only environment variable it
    reads is COS_TUNING="name=value,...", parsed once (unknown names and malformed items are skipped).  No device needed."""
    import subprocess
    import sys
    from cosdata_amd import _lib
    names = re.findall(r'"([a-z0-9_]+)"', re.search(r"NAMES\[TUNE_COUNT\] = \{(.*?)\};", open(os.path.join(ROOT, "cosdata_amd", "csrc", "tuning.hip")).read(), re.S).group(1))
    saved = {n: _lib.tuning_get(n) for n in names}         # a run steered through COS_TUNING keeps its knobs for the tests after this one
    try:
        _tuning_registry_checks(_lib)
    finally:
        _lib.tuning_clear(None)
        for n, v in saved.items():
            if v is not None:
                _lib.tuning_set(n, v)


def _tuning_registry_checks(_lib):
    import subprocess
    import sys
    _lib.tuning_clear(None)
    assert _lib.tuning_get("walk_pb") is None
    _lib.tuning_set("walk_pb", 4)
    assert _lib.tuning_get("walk_pb") == 4
    with _lib.tuning(flat_tile_kernel=1, walk_pb=8):
        assert _lib.tuning_get("flat_tile_kernel") == 1 and _lib.tuning_get("walk_pb") == 8
    assert _lib.tuning_get("flat_tile_kernel") is None and _lib.tuning_get("walk_pb") == 4
    _lib.tuning_clear("walk_pb")
    assert _lib.tuning_get("walk_pb") is None
    with pytest.raises(_lib.CosdataError):
        _lib.tuning_set("no_such_knob", 1)
    with pytest.raises(_lib.CosdataError):                  # a block with a rejected name leaves nothing of itself behind
        with _lib.tuning(walk_pb=8, no_such_knob=1):
            pass
    assert _lib.tuning_get("walk_pb") is None
    with pytest.raises(_lib.CosdataError):
        _lib.tuning_set("walk_pb", -(2 ** 63))               # INT64_MIN is the "unset" marker
    code = ("from cosdata_amd import _lib; print(_lib.tuning_get('walk_pb_upper'), _lib.tuning_get('flat_pf'), _lib.tuning_get('walk_pb'), "
            "_lib.tuning_get('sparse_layout'))")
    env = dict(os.environ, COS_TUNING="walk_pb_upper=4,bogus=1,flat_pf=3,sparse_layout=,=5")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.split() == ["4", "3", "None", "None"]


def test_the_library_reads_one_environment_variable():
    """`getenv` appears once in the library's sources (tuning.hip: COS_TUNING) — the launch policies of a library the Rust host links
    must not depend on whatever else is in the process environment"""
    import glob
    hits = []
    for f in sorted(glob.glob(os.path.join(ROOT, "cosdata_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".inc")):
            for i, line in enumerate(open(f, errors="replace"), 1):
                if "getenv(" in line:
                    hits.append((os.path.basename(f), i))
    assert [h[0] for h in hits] == ["tuning.hip"], hits


def test_tuning_names_follow_the_enum_and_are_documented():
    """tuning.hip's NAMES[] is indexed by tuning.h's TuneKey: a name out of order would make cos_tuning_set steer a different policy
    than the one it names.  Every knob is listed in INTEGRATION.md §16."""
    csrc = os.path.join(ROOT, "cosdata_amd", "csrc")
    enum = re.findall(r"^\s+TUNE_([A-Z0-9_]+),", open(os.path.join(csrc, "tuning.h")).read(), re.M)
    body = re.search(r"NAMES\[TUNE_COUNT\] = \{(.*?)\};", open(os.path.join(csrc, "tuning.hip")).read(), re.S).group(1)
    names = re.findall(r'"([a-z0-9_]+)"', body)
    assert names == [e.lower() for e in enum] and len(names) == len(set(names)) >= 20
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 16."):]
    assert re.findall(r"^\| `([a-z0-9_]+)` \|", sec, re.M) == names
    from cosdata_amd import _lib
    for n in names:                                   # every documented name is accepted by the library
        _lib.tuning_get(n)
