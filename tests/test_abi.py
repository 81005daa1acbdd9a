"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the headers
declare; without a GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orb[xm]_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    pkg.build()
    L = C.CDLL(pkg.LIB_PATH)
    names = _declared("orbx.h") + _declared("orbm.h")
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing export " + n
    assert sorted(names) == sorted(pkg.EXPORTS)


def test_no_cpu_fallback_without_device(pkg):
    L = pkg.lib()
    if L.orbx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.OrbError, match="no CPU fallback"):
        pkg.ORBextractor(1000)
    with pytest.raises(pkg.OrbError, match="no CPU fallback"):
        pkg.ORBmatcher(0.7)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orb-slam3_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orbref" not in txt and "oracle/" not in txt.replace("# oracle/", ""), os.path.join(dirpath, f)


def test_scalar_helpers_on_cpu(pkg, oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
    assert pkg.ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    c = rng.integers(0, 9, 30).astype(np.int32)
    assert np.array_equal(pkg.ORBmatcher.ComputeThreeMaxima(c), oracle.three_maxima(c))


def test_cpp_facade_compiles_against_the_c_abi(pkg, tmp_path):
    """The drop-in C++ classes (facade/ORBextractor.h, ORBmatcher.h) build with plain g++ and link the library."""
    import subprocess
    pkg.build()
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "facade_smoke.cpp"),
                           "-L", os.path.join(ROOT, "orb-slam3_amd"), "-lorbslam3_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "orb-slam3_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
