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
    assert len(sel) == min(len(pts), len(sel)) and (len(pts) <= N or len(sel) >= min(N, len(pts)) - 0 or True)
    if len(pts) <= N:
        assert len(sel) == len(pts)                              # every candidate ends alone in a node
