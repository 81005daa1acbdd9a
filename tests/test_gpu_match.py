"""GPU parity of the matcher primitives vs the CPU oracle (bit-exact integer work)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


@pytest.mark.parametrize("nq,nt", [(1, 1), (1, 2), (5, 1), (64, 64), (65, 1500), (1500, 1500), (300, 2049), (7, 0)])
def test_knn2_random(pkg, oracle, nq, nt):
    rng = np.random.default_rng(nq * 10007 + nt)
    q, t = _rand_desc(rng, nq), _rand_desc(rng, nt)
    m = pkg.ORBmatcher(0.7)
    idx, dist = m.knn2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_knn2_ties_take_lower_index(pkg, oracle):
    rng = np.random.default_rng(3)
    base = _rand_desc(rng, 40)
    t = np.concatenate([base, base, base[:7]])                    # every distance appears 2-3 times
    q = base[:25].copy(); q[:, 0] ^= 1
    m = pkg.ORBmatcher()
    idx, dist = m.knn2(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert np.all(idx[:, 0] == np.arange(25)) and np.all(dist[:, 0] == 1) and np.all(idx[:, 1] == np.arange(25) + 40)


def test_knn2_on_real_descriptors_and_properties(pkg, oracle, synth):
    l, r = synth.gen_stereo_pair(512, 512, 21)
    ex = pkg.ORBextractor(1500, max_size=(512, 512))
    _, _, dl = ex(l, (0, 511)); _, _, dr = ex(r, (0, 511))
    m = pkg.ORBmatcher(0.7)
    idx, dist = m.knn2(dl, dr)
    ridx, rdist = oracle.knn2(dl, dr)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    # size-independent properties: sortedness, self-match distance 0
    assert np.all(dist[:, 0] <= dist[:, 1])
    sidx, sdist = m.knn2(dl, dl)
    assert np.all(sdist[:, 0] == 0)


def test_scalar_helpers(pkg, oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        a, b = _rand_desc(rng, 1)[0], _rand_desc(rng, 1)[0]
        assert pkg.ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b)
    for _ in range(20):
        c = rng.integers(0, 12, 30).astype(np.int32)
        assert np.array_equal(pkg.ORBmatcher.ComputeThreeMaxima(c), oracle.three_maxima(c))


@pytest.mark.gpu
def test_knn2_rejects_train_sets_beyond_the_packed_index_range(pkg):
    """k_knn2 packs (distance << 22 | train index) into one key: 2^22 train descriptors per pair is the documented limit."""
    import ctypes as C
    m = pkg.ORBmatcher()
    one = pkg.DeviceBuffer(64)
    rc = m.L.orbm_knn2_batch_async(m.h, one.ptr, 1, one.ptr, one.ptr, 1 << 22, one.ptr, 1, 1 << 22, one.ptr, one.ptr)
    assert rc < 0
    assert b"2^22" in m.L.orbm_last_error()
