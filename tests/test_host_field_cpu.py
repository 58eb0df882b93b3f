"""Host-side field arithmetic of the product library (gemini_amd/csrc/host_field.hpp: the tail of every MSM -- window Horner,
normalisation, partial-point sums): the x86-64 BMI2 + ADX Montgomery product (host_fq_adx.hpp) must agree with the portable loop on
random and boundary operands and through the group law, with either compiler that builds the library.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_field_check.cpp")
COMPILERS = [c for c in ("g++", "/opt/rocm/lib/llvm/bin/clang++") if shutil.which(c) or os.path.exists(c)]


@pytest.mark.parametrize("cxx", COMPILERS)
def test_adx_product_equals_the_portable_one(tmp_path, cxx):
    exe = tmp_path / "host_field_check"
    subprocess.check_call([cxx, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "gemini_amd", "csrc"), SRC, "-o", str(exe)])
    out = subprocess.run([str(exe), "200000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
    # the portable path alone (what a host without ADX runs) passes the same group-law checks
    env = dict(os.environ, GM_HOST_ADX="0")
    out0 = subprocess.run([str(exe), "20000"], capture_output=True, text=True, timeout=300, env=env)
    assert out0.returncode == 0 and "adx usable: 0" in out0.stdout and "ok" in out0.stdout, out0.stdout + out0.stderr
