"""The A/B reference kernels and test knobs (tests/ab) run against liborbslam3_amd_ab.so -- the product ABI built with -DORBX_AB -- in a
child process, so that every other test of this suite runs on the product library, which carries none of them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AB_LIB = os.path.join(ROOT, "orb-slam3_amd", "liborbslam3_amd_ab.so")


@pytest.mark.gpu
def test_ab_variants_in_a_child_process():
    assert os.path.exists(AB_LIB), "build it: make -C orb-slam3_amd/csrc"
    env = dict(os.environ, ORB_LIB=AB_LIB, ORB_AB_CHILD="1")
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "ab"), "-m", "gpu", "-x", "-q"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-1000:])
    assert " passed" in p.stdout and "failed" not in p.stdout


def test_product_library_has_no_ab_switches():
    """Not even the names: the switches are compiled out of liborbslam3_amd.so (csrc: ab_env / AB_LAUNCH), present in the _ab build."""
    names = [b"ORBX_FAST_V1", b"ORBX_QT_V1", b"ORBX_OD_V1", b"ORBX_BLUR_V2", b"ORBX_FAST_V3", b"ORBX_FAST_QCAP", b"ORBX_SERIAL",
             b"ORBM_KNN2_VALU", b"ORBM_WINDOW_CAP", b"ORBX_DL_KERNEL", b"ORBM_TOPK_WAVE", b"ORBM_CLAIM_V1", b"ORBX_BLUR_LATE",
             b"ORBX_FAST_WGS"]
    prod = open(os.path.join(ROOT, "orb-slam3_amd", "liborbslam3_amd.so"), "rb").read()
    ab = open(AB_LIB, "rb").read()
    for n in names:
        assert n not in prod, n
        assert n in ab, n
    for k in (b"k_fast3", b"k_quadtreeE", b"k_blur2", b"k_orient_descE", b"k_copy_out", b"k_track_topkE", b"k_track_claimE"):       # mangled-name fragments of the A/B kernels
        assert k not in prod and k in ab, k
