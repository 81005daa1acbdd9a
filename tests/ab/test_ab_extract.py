"""A/B reference kernels, scheduling variants and test knobs: compiled only into liborbslam3_amd_ab.so (-DORBX_AB) and read from the
environment there.  tests/test_ab_child.py runs this directory in a child process with ORB_LIB pointing at that build; the product
library (liborbslam3_amd.so) carries none of these switches (tests/test_abi.py checks that it does not even contain their names)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
from test_gpu_extract import _check, CASES


@pytest.mark.parametrize("switch", ["ORBX_BLUR_V2", "ORBX_BLUR_LATE"])
@pytest.mark.parametrize("w,h,nf,seed,lap,kind", [CASES[0], CASES[6], (1920, 1080, 2000, 77, (0, 0), "textured")])
def test_blur_variants_bit_exact(pkg, oracle, synth, monkeypatch, switch, w, h, nf, seed, lap, kind):
    # default: the 7x7 blur as two int8 Toeplitz products on the matrix cores (k_blur3) scheduled beside FAST -- covered by
    # every other test here.  ORBX_BLUR_V2 = the VALU kernel (k_blur2) beside the quadtree, ORBX_BLUR_LATE = k_blur3 beside the
    # quadtree: every blurred level and the descriptors must equal the oracle's either way (odd sizes: ragged strips and folds)
    monkeypatch.setenv(switch, "1")
    ex = pkg.ORBextractor(100, max_size=(w, h), max_batch=1)
    assert not ex.blur_in_pass()
    ex.close()
    _check(pkg, oracle, synth, w, h, nf, seed, lap, kind)


def test_ab_reference_kernels_stay_bit_exact(pkg, oracle, synth, monkeypatch):
    """The simple first-generation kernels (per-cell FAST, lane-0 quadtree, one-keypoint-per-wave descriptors, single
    stream) are kept as A/B references behind environment switches read at orbx_create: they must give the same bits."""
    img = synth.gen_image(752, 480, 21)
    n_ref, kps_ref, desc_ref, mono_ref = oracle.Extractor(1000)(img, (0, 1000))
    for env in (["ORBX_FAST_V1"], ["ORBX_QT_V1"], ["ORBX_OD_V1"], ["ORBX_SERIAL"], ["ORBX_QT_WIDE"],
                ["ORBX_FAST_V1", "ORBX_QT_V1", "ORBX_OD_V1", "ORBX_SERIAL"]):
        for e in env:
            monkeypatch.setenv(e, "1")
        ex = pkg.ORBextractor(1000, max_size=(752, 480))
        mono, kps, desc = ex(img, (0, 1000))
        ex.close()
        for e in env:
            monkeypatch.delenv(e)
        assert mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref), env


@pytest.mark.parametrize("qcap", [64, 128, 256])
def test_fast_queue_overflow_cells_are_redone(pkg, oracle, synth, monkeypatch, qcap):
    """k_fast3 keeps a bounded LDS queue per cell; cells whose quick-reject survivors exceed it are redone by k_fast_fix.
    ORBX_FAST_QCAP forces a tiny queue so that many (64) or a few (256) cells take that route; two batches in a row also
    check that the overflow list is re-armed."""
    imgs = [synth.gen_image(752, 480, 31 + i) for i in range(3)]
    ref = oracle.Extractor(1000)
    monkeypatch.setenv("ORBX_FAST_QCAP", str(qcap))
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=3)
    monkeypatch.delenv("ORBX_FAST_QCAP")
    try:
        for rep in range(2):
            out = ex.extract_batch(imgs, [(0, 1000)] * 3)
            for img, (mono, kps, desc) in zip(imgs, out):
                n_ref, kps_ref, desc_ref, mono_ref = ref(img, (0, 1000))
                assert mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
    finally:
        ex.close()
