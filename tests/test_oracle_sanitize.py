"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers do not exist on this pool, so the memory-safety check
runs on the CPU restatement only): the oracle's own golden / known-answer / property tests are re-run in a child process
against oracle/_san/liborbref_san.so.  A heap overflow, use-after-free or UB report aborts the child -> this test fails."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_clean_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_san/liborbref_san.so"])
    env = dict(os.environ)
    env["ORBREF_LIB"] = os.path.join(ROOT, "oracle", "_san", "liborbref_san.so")
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"       # the interpreter itself 'leaks' by design
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_golden.py", "tests/test_oracle_kat.py", "tests/test_properties.py",
                        "tests/test_oracle_searches_cpu.py"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail
