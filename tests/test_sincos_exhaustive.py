"""The descriptor kernel's sin/cos (csrc/od_sincos.h: Cody-Waite + fdlibm kernels in double) against the oracle's definition
(float)cos((double)angle) on EVERY float in [0, 2 pi] -- 1.09e9 values, a few seconds on 8 threads."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_od_sincos_equals_libm_on_every_float(tmp_path):
    exe = str(tmp_path / "sincos_exhaustive")
    subprocess.check_call(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fno-fast-math", "-o", exe,
                           os.path.join(ROOT, "tests", "sincos_exhaustive.cpp"), "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout[-500:]
