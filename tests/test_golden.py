"""Golden fixtures (tests/golden/, written by tools/gen_golden.py from the CPU oracle -- the reference has no vectors of
its own and cannot be built here, so parity stays 'unpinned'; these fixtures pin the oracle against drift and give the
GPU path a target that needs no oracle at run time)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_oracle_reproduces_golden_extraction(oracle):
    g = _load("extract_320x240_nf300_l4.npz")
    nf, nl, ini, mn = [int(x) for x in g["params"]]
    ex = oracle.Extractor(nf, float(g["scale"]), nl, ini, mn)
    n, kps, desc, mono = ex(g["image"], tuple(g["lap"]))
    assert mono == int(g["mono"]) and kps.tobytes() == g["kps"].tobytes() and np.array_equal(desc, g["desc"])
    for l in range(nl):
        assert np.array_equal(ex.level_candidates(l), g["cand%d" % l])
    assert np.array_equal(ex.level_image(3), g["level3"]) and np.array_equal(ex.level_image(0, blurred=True), g["blur0"])
    g2 = _load("extract_320x240_lap100_200.npz")
    n, kps, desc, mono = ex(g["image"], tuple(g2["lap"]))
    assert mono == int(g2["mono"]) and kps.tobytes() == g2["kps"].tobytes() and np.array_equal(desc, g2["desc"])
    # lapping rule: rows [0, mono) are outside [100,200], the tail is inside (ORBextractor.cc:1644-1653)
    assert np.all((kps["x"][:mono] < 100) | (kps["x"][:mono] > 200)) and np.all((kps["x"][mono:] >= 100) & (kps["x"][mono:] <= 200))


def test_oracle_reproduces_golden_knn2(oracle):
    g = _load("knn2_120x150.npz")
    idx, dist = oracle.knn2(g["q"], g["t"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_extraction(pkg):
    g = _load("extract_320x240_nf300_l4.npz")
    nf, nl, ini, mn = [int(x) for x in g["params"]]
    ex = pkg.ORBextractor(nf, float(g["scale"]), nl, ini, mn, max_size=(320, 240))
    mono, kps, desc = ex(g["image"], tuple(g["lap"]))
    assert mono == int(g["mono"]) and kps.tobytes() == g["kps"].tobytes() and np.array_equal(desc, g["desc"])
    for l in range(nl):
        assert np.array_equal(ex.level_candidates(l), g["cand%d" % l])
    assert np.array_equal(ex.level_image(3), g["level3"]) and np.array_equal(ex.level_image(0, blurred=True), g["blur0"])
    g2 = _load("extract_320x240_lap100_200.npz")
    mono, kps, desc = ex(g["image"], tuple(g2["lap"]))
    assert mono == int(g2["mono"]) and kps.tobytes() == g2["kps"].tobytes() and np.array_equal(desc, g2["desc"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_knn2(pkg):
    g = _load("knn2_120x150.npz")
    idx, dist = pkg.ORBmatcher(0.7).knn2(g["q"], g["t"])
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])
