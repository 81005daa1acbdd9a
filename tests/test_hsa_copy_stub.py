"""The copy thread's DMA path (csrc/hsa_copy.h) on stubbed HSA entry points: refusal before anything is issued, a later
piece refused while earlier ones complete (the completion signal must still reach zero), and a DMA that never completes
(finite deadline -> error instead of a hung copy thread).  Host-only: no GPU, no HSA runtime."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hsa_copy_failure_paths(tmp_path):
    exe = str(tmp_path / "hsa_copy_stub")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I/opt/rocm/include", "-o", exe,
                           os.path.join(ROOT, "tests", "hsa_copy_stub.cpp"), "-ldl"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "hsa_copy_stub ok" in r.stdout, r.stdout + r.stderr
