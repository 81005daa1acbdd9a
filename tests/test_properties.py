"""Property tests (hypothesis) on the CPU oracle: cheap invariants the GPU parity tests build on."""
import numpy as np
from hypothesis import given, settings, strategies as st

desc32 = st.binary(min_size=32, max_size=32)


@settings(max_examples=200, deadline=None)
@given(desc32, desc32)
def test_descriptor_distance_is_popcount_of_xor(oracle, a, b):
    x = np.frombuffer(a, np.uint8); y = np.frombuffer(b, np.uint8)
    d = oracle.hamming(x, y)
    assert d == int(np.unpackbits(x ^ y).sum()) == oracle.hamming(y, x)
    assert 0 <= d <= 256 and (d == 0) == (a == b)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 40), min_size=30, max_size=30))
def test_three_maxima_invariants(oracle, counts):
    ind = oracle.three_maxima(np.array(counts, np.int32)).tolist()
    c = np.array(counts)
    if c.max() == 0:
        assert ind == [-1, -1, -1]
        return
    assert c[ind[0]] == c.max() and ind[0] == int(np.argmax(c))          # first-wins on ties
    kept = [i for i in ind if i >= 0]
    assert len(set(kept)) == len(kept)
    for i in kept[1:]:
        assert c[i] >= 0.1 * c.max() or np.float32(c[i]) >= np.float32(0.1) * np.float32(c.max())


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(8, 30))
def test_fast_keypoints_at_higher_threshold_are_scored_consistently(oracle, seed, thr):
    """corner@t <=> score >= t and the score does not depend on t: every keypoint found at threshold
    t2 > t1 has score >= t2, and reappears at t1 unless a newly admitted neighbour suppresses it."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (6, 7)).astype(np.float64)
    img = np.clip(np.rint(np.kron(base, np.ones((6, 6))) + rng.normal(0, 5, (36, 42))), 0, 255).astype(np.uint8)
    lo = {(x, y): s for x, y, s in oracle.fast(img, 7).tolist()}
    hi = oracle.fast(img, thr).tolist()
    for x, y, s in hi:
        assert s >= thr and oracle.fast_score(img, x, y) == s
        if (x, y) in lo:
            assert lo[(x, y)] == s
        else:                                                   # suppressed at the lower threshold by a weaker-than-thr neighbour? impossible:
            nb = [oracle.fast_score(img, x + dx, y + dy) for dx in (-1, 0, 1) for dy in (-1, 0, 1)
                  if (dx or dy) and 3 <= x + dx < 39 and 3 <= y + dy < 33]
            assert any(7 <= v < thr and v >= s for v in nb) is False


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 255), st.integers(40, 90), st.integers(40, 90))
def test_resize_and_blur_preserve_constants(oracle, val, w, h):
    img = np.full((h, w), val, np.uint8)
    assert np.all(oracle.resize_linear(img, max(8, int(round(w / 1.2))), max(8, int(round(h / 1.2)))) == val)
    assert np.all(oracle.gauss7(img) == val)


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(1, 60))
def test_distribute_returns_at_most_n_plus_3_distinct_candidates(oracle, seed, N):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(0, 400))
    pts = np.unique(np.stack([rng.integers(0, 720, n), rng.integers(0, 448, n)], 1), axis=0)
    xyr = np.concatenate([pts, rng.integers(7, 200, (len(pts), 1))], 1).astype(np.int32)
    sel = oracle.distribute(xyr, 16, 736, 16, 464, N)
    assert len(set(sel.tolist())) == len(sel) and len(sel) <= max(N + 3, 8)
    # (not "== len(pts)" when len(pts) <= N: a split whose points all fall into one child leaves the list size unchanged and
    # ORBextractor.cc:889-895 then stops with two candidates still sharing a node)
    assert len(sel) <= len(pts)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**31 - 1), st.integers(3, 40), st.integers(3, 30))
def test_ingest_invariants(oracle, seed, w, h):
    """cvtColor: gray images stay themselves, channel order only swaps R and B, 14- and 15-bit tables differ by at most 1;
    remap: the identity map reproduces the image and integer shifts move it (zero border)."""
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    g = rgb[..., 0]
    ggg = np.stack([g, g, g], axis=-1)
    for bits in (14, 15):
        assert np.array_equal(oracle.gray_from_color(ggg, False, bits), g)
        assert np.array_equal(oracle.gray_from_color(rgb, False, bits), oracle.gray_from_color(rgb[..., ::-1], True, bits))
    d = oracle.gray_from_color(rgb, False, 14).astype(int) - oracle.gray_from_color(rgb, False, 15).astype(int)
    assert np.abs(d).max() <= 1
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(g, xs, ys), g)
    sh = oracle.remap_linear(g, xs + 1.0, ys)
    assert np.array_equal(sh[:, :-1], g[:, 1:]) and not sh[:, -1].any()


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**31 - 1), st.floats(0.5, 8.0))
def test_clahe_invariants(oracle, seed, clip):
    """CLAHE: the per-tile mapping is monotone, so the order of gray levels inside a constant-tile image is preserved;
    output is deterministic and differs from the unclipped equalisation only by a bounded amount of contrast."""
    rng = np.random.default_rng(seed)
    tile = rng.integers(0, 256, (16, 16), dtype=np.uint8)
    img = np.tile(tile, (2, 3))                                          # identical tiles: the blend is the tile's own LUT
    out = oracle.clahe(img, clip, (3, 2))
    assert np.array_equal(out, oracle.clahe(img, clip, (3, 2)))
    flat_in, flat_out = img.ravel().astype(int), out.ravel().astype(int)
    order = np.argsort(flat_in, kind="stable")
    assert np.all(np.diff(flat_out[order]) >= 0)                         # monotone LUT
    same = flat_in[:, None] == flat_in[None, :256]
    assert all(len(set(flat_out[flat_in == v])) == 1 for v in np.unique(flat_in)[:20])   # one output per input level


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**31 - 1))
def test_undistort_and_frustum_invariants(oracle, pkg, seed):
    """Undistortion with k1 == 0 is the identity copy; distort-free projection of a point on the optical axis lands on the
    principal point; a point seen exactly head-on at distance d with maxDistance = d * 1.2^k predicts level k."""
    rng = np.random.default_rng(seed)
    M = oracle._oracle_matcher_class()()
    K = [float(rng.uniform(300, 600)), float(rng.uniform(300, 600)), float(rng.uniform(200, 400)), float(rng.uniform(150, 300))]
    kps = np.zeros(8, pkg.KP_DTYPE); kps["x"] = rng.uniform(0, 640, 8); kps["y"] = rng.uniform(0, 480, 8)
    assert M.UndistortKeyPoints(kps, K, [0.0, 0.3, 0.01, 0.01]).tobytes() == kps.tobytes()
    pp = np.zeros(1, pkg.KP_DTYPE); pp["x"] = np.float32(K[2]); pp["y"] = np.float32(K[3])
    u = M.UndistortKeyPoints(pp, K, [float(rng.uniform(-0.4, 0.4)), 0.05, 0.001, -0.001])
    assert u["x"][0] == np.float32(K[2]) and u["y"][0] == np.float32(K[3])
    k = int(rng.integers(0, 8)); d = float(rng.uniform(1.0, 10.0))
    lsf = float(np.log(np.float32(1.2)))
    mx = np.array([d * 1.2 ** k * 0.97], np.float32)                      # ratio a little under 1.2^k: ceil() gives k for k >= 1
    cnt, o = M.isInFrustum(np.array([[0, 0, d]], np.float32), np.array([[0, 0, 1]], np.float32), np.array([0.01], np.float32), mx,
                           np.eye(3), np.zeros(3), np.zeros(3), K, [0.0, 2 * K[2], 0.0, 2 * K[3]], 40.0, 0.5, lsf, 8)
    assert cnt == 1 and o["proj_x"][0] == np.float32(K[2]) and o["proj_y"][0] == np.float32(K[3])
    assert o["level"][0] == (k if k >= 1 else 0) and o["view_cos"][0] == 1.0
