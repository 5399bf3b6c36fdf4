"""Device check of the walk's evaluation-block reduction (cosdev::wave_reduce_rows, device_common.h): the butterfly over
v_permlane32_swap / v_permlane16_swap / DPP must leave, in every lane of group g, the exact u32 sum of row g."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _compile(out, extra=()):
    src = os.path.join(ROOT, "tests", "cxx", "wave_reduce_check.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "cosdata_amd", "csrc"),
                           *extra, src, "-o", str(out)])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_wave_reduce_check_compiles_for_gfx950(tmp_path):
    _compile(tmp_path / "wave_reduce_check.o", extra=("-c",))


@pytest.mark.gpu
def test_wave_reduce_rows_matches_host_sums(tmp_path):
    exe = tmp_path / "wave_reduce_check"
    _compile(exe)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK"), r.stdout
