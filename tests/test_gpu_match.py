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


@pytest.mark.parametrize("nq,nt", [(31, 31), (32, 32), (33, 33), (63, 65), (255, 95), (256, 96), (257, 97), (1000, 1000), (1, 4100)])
def test_knn2_tile_edges(pkg, oracle, nq, nt):
    # k_knn2_mfma works on 32 x 32 tiles, 64 queries per wave, 256 per workgroup, train tiles walked last to first with a
    # ragged last tile (the popcount kernel it replaced: tests/ab).
    rng = np.random.default_rng(nq * 31 + nt)
    q, t = _rand_desc(rng, nq), _rand_desc(rng, nt)
    q[0] = 0; q[-1] = 255                                        # popcount 0 / 256 queries
    t[0] = 255; t[-1] = 0                                        # distances 0 and 256 occur
    if nt > 40:
        t[nt - 3] = t[5]                                         # equal distances far apart: lower index first
    ridx, rdist = oracle.knn2(q, t)
    m = pkg.ORBmatcher()
    idx, dist = m.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert rdist.min() == 0
    # the extreme distances themselves: two train rows, all ones and all zeros
    t2 = np.stack([np.full(32, 255, np.uint8), np.zeros(32, np.uint8)])
    i2, d2 = m.knn2(np.stack([np.zeros(32, np.uint8), np.full(32, 255, np.uint8)]), t2)
    assert i2.tolist() == [[1, 0], [0, 1]] and d2.tolist() == [[0, 256], [0, 256]]


def test_knn2_batch_ragged_pairs(pkg, oracle):
    # several pairs in one launch, each with its own counts inside fixed strides (the bench / fisheye-stereo layout)
    rng = np.random.default_rng(99)
    import ctypes as C
    nqs, nts = [100, 0, 257, 64, 300], [333, 50, 1, 0, 300]
    qs, ts = 320, 352
    P = len(nqs)
    q = rng.integers(0, 256, (P, qs, 32), dtype=np.uint8); t = rng.integers(0, 256, (P, ts, 32), dtype=np.uint8)
    m = pkg.ORBmatcher()
    idx = np.full((P, qs, 2), -7, np.int32); dist = np.full((P, qs, 2), -7, np.int32)
    nq = np.array(nqs, np.int32); nt = np.array(nts, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = m.L.orbm_knn2_batch(m.h, pkg.HOST, p(q), qs, p(nq), p(t), ts, p(nt), P, p(idx), p(dist))
    assert rc == 0, m.L.orbm_last_error()
    for i in range(P):
        if nqs[i] == 0:
            continue
        ridx, rdist = oracle.knn2(q[i, :nqs[i]], t[i, :nts[i]])
        assert np.array_equal(idx[i, :nqs[i]], ridx) and np.array_equal(dist[i, :nqs[i]], rdist), i


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


def test_knn2_lowe_ratio_in_the_epilogue(pkg, oracle):
    """M16 'knn2 + ratio': good[q] as Frame.cc:1465 decides it -- two neighbours and (float)d0 < (float)d1 * 0.7 in double -- written
    by the kernel itself, incl. queries with fewer than two train rows and exact-boundary distances (7 vs 10)."""
    import ctypes as C
    rng = np.random.default_rng(2024)
    nqs, nts = [300, 5, 64, 1], [400, 1, 0, 2]
    qs, ts = 320, 416
    P = len(nqs)
    q = rng.integers(0, 256, (P, qs, 32), dtype=np.uint8); t = rng.integers(0, 256, (P, ts, 32), dtype=np.uint8)
    # boundary: query 0 of pair 0 (all zeros) at distance 7 from train row 0 and 10 from row 1: 7 < 10 * 0.7 is FALSE (the double product is 7.0)
    q[0, 0] = 0; t[0, 0] = 0; t[0, 0, 0] = 0x7F; t[0, 1] = 0; t[0, 1, 0] = 0xFF; t[0, 1, 1] = 0x03
    # query 1 (all ones) at distances 6 and 10 from rows 2 and 3: accepted
    q[0, 1] = 0xFF; t[0, 2] = 0xFF; t[0, 2, 0] = 0x03; t[0, 3] = 0xFF; t[0, 3, 0] = 0x00; t[0, 3, 1] = 0xFC
    dq, dt = pkg.DeviceBuffer(q.nbytes), pkg.DeviceBuffer(t.nbytes)
    dq.upload(q); dt.upload(t)
    nq = np.array(nqs, np.int32); nt = np.array(nts, np.int32)
    dnq, dnt = pkg.DeviceBuffer(4 * P), pkg.DeviceBuffer(4 * P)
    dnq.upload(nq); dnt.upload(nt)
    di, dd, dg = pkg.DeviceBuffer(P * qs * 8), pkg.DeviceBuffer(P * qs * 8), pkg.DeviceBuffer(P * qs)
    m = pkg.ORBmatcher()
    rc = m.L.orbm_knn2_ratio_batch_async(m.h, dq.ptr, qs, dnq.ptr, dt.ptr, ts, dnt.ptr, P, 0.7, di.ptr, dd.ptr, dg.ptr)
    assert rc == 0, m.L.orbm_last_error()
    m.sync()
    idx = di.download(np.int32, P * qs * 2).reshape(P, qs, 2); dist = dd.download(np.int32, P * qs * 2).reshape(P, qs, 2)
    good = dg.download(np.uint8, P * qs).reshape(P, qs)
    for i in range(P):
        ri, rd = oracle.knn2(q[i, :nqs[i]], t[i, :nts[i]])
        assert np.array_equal(idx[i, :nqs[i]], ri) and np.array_equal(dist[i, :nqs[i]], rd)
        want = np.array([1 if (a >= 0 and b >= 0 and float(np.float32(a)) < float(np.float32(b)) * 0.7) else 0 for a, b in rd.tolist()], np.uint8)
        assert np.array_equal(good[i, :nqs[i]], want), i
    assert dist[0, 0].tolist() == [7, 10] and good[0, 0] == 0
    assert dist[0, 1].tolist() == [6, 10] and good[0, 1] == 1
    assert good[1, :nqs[1]].sum() == 0 and good[2, :nqs[2]].sum() == 0  # one / zero train rows: never two neighbours
