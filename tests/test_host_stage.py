"""StagePool (cosdata_amd/csrc/host_stage.h): the host-buffer API's parallel staging copy is plain host C++ — built with g++ and run here."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_pool_copies_exactly_the_range(tmp_path):
    exe = str(tmp_path / "host_stage_test")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "cosdata_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cxx", "host_stage_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host_stage ok" in out.stdout, out.stdout + out.stderr
