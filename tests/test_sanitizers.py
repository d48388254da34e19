"""CPU sanitizer runs (GPU ASan is not available on the pool): the oracle's C restatement and the
product library's host-only plan / layout arithmetic, both under AddressSanitizer + UBSan (incl. float-cast-overflow)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined,float-cast-overflow,float-divide-by-zero", "-fno-sanitize-recover=all",
       "-fno-omit-frame-pointer", "-O1", "-g"]

needs_gcc = pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="no host toolchain")


@needs_gcc
def test_oracle_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "no sanitizer report" in r.stdout


@needs_gcc
def test_plan_and_layout_arithmetic_under_asan_ubsan(tmp_path):
    """fresnel_amd/csrc/fgs_plan.cpp is the translation unit libfgs_hip.so links (fresnel_amd/build.py); here g++
    builds the same file with tests/native/plan_sanitize.cpp and sweeps valid, invalid and hostile FgsDims."""
    exe = str(tmp_path / "plan_sanitize")
    cmd = ["g++", "-std=c++17"] + SAN + ["-o", exe, os.path.join(ROOT, "tests", "native", "plan_sanitize.cpp"),
                                         os.path.join(ROOT, "fresnel_amd", "csrc", "fgs_plan.cpp")]
    subprocess.check_call(cmd, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " 0 failures" in r.stdout


def test_plan_source_is_the_one_the_library_builds():
    from fresnel_amd import build as fb
    assert "fgs_plan.cpp" in fb.SOURCES
    src = open(os.path.join(ROOT, "fresnel_amd", "csrc", "fgs_plan.cpp")).read()
    assert "hip/hip_runtime" not in src and "fgs_internal.h" not in src  # host-only: g++ must be able to build it
