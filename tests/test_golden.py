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


def _geometry_outputs(M, gray, g):
    und = M.UndistortKeyPoints(g["kps"], g["K"], g["D"])
    bounds = M.ComputeImageBounds(752, 480, g["K"], g["D"])
    cnt, fr = M.isInFrustum(g["Pw"], g["normal"], g["min_dist"], g["max_dist"], g["R"], g["t"], g["Ow"], g["K"], g["bounds"],
                            47.90639, 0.5, float(g["lsf"]), 8)
    return und, bounds, cnt, fr


def _check_geometry(und, bounds, cnt, fr, g):
    assert und.tobytes() == g["und"].tobytes() and bounds.tobytes() == g["bounds"].tobytes() and cnt == int(g["cnt"])
    v = g["fr_in_view"].astype(bool)
    for k in ("in_view", "proj_x", "proj_y"):
        assert fr[k].tobytes() == g["fr_" + k].tobytes(), k
    for k in ("proj_xr", "depth", "level", "view_cos"):
        assert fr[k][v].tobytes() == g["fr_" + k][v].tobytes(), k


def test_oracle_reproduces_golden_geometry(oracle):
    """tools/gen_golden_geometry.py: undistort / image bounds / frustum / cvtColor vectors (oracle-generated, parity unpinned)."""
    g = _load("geometry.npz")
    M = oracle._oracle_matcher_class()()
    _check_geometry(*_geometry_outputs(M, None, g), g)
    assert np.array_equal(oracle.gray_from_color(g["rgb"], False, 14), g["gray14_rgb"])
    assert np.array_equal(oracle.gray_from_color(g["rgb"], True, 15), g["gray15_bgr"])
    assert np.array_equal(oracle.remap_linear(g["g96"], g["mapx"], g["mapy"]), g["remap"])
    assert np.array_equal(oracle.clahe(g["g96"], 3.0, (4, 3)), g["clahe"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_geometry(pkg):
    g = _load("geometry.npz")
    _check_geometry(*_geometry_outputs(pkg.ORBmatcher(), None, g), g)
    ex = pkg.ORBextractor(300, max_size=(320, 240))
    try:
        for blue_first, bits, key in ((False, 14, "gray14_rgb"), (True, 15, "gray15_bgr")):
            buf, stride = ex.gray_from_color([g["rgb"]], blue_first, bits)
            h, w = g["rgb"].shape[:2]
            assert np.array_equal(buf.download(np.uint8, stride * h).reshape(h, stride)[:, :w], g[key])
        h, w = g["g96"].shape
        buf, stride = ex.remap_linear([g["g96"]], g["mapx"], g["mapy"])
        assert np.array_equal(buf.download(np.uint8, stride * h).reshape(h, stride)[:, :w], g["remap"])
        buf, stride = ex.clahe([g["g96"]], 3.0, (4, 3))
        assert np.array_equal(buf.download(np.uint8, stride * h).reshape(h, stride)[:, :w], g["clahe"])
    finally:
        ex.close()
