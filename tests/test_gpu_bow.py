"""SURVEY 8(f).1 -- DBoW2 vocabulary transform on the GPU vs the oracle restatement of the vendored DBoW2 code.
ORBvoc.txt is a missing blob in the reference snapshot: the tree here is synthetic (same text format, k-ary, L levels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_vocab(path, k, L, seed):
    """DBoW2 text format (TemplatedVocabulary.h:1338-1440): 'k L scoring weighting' then 'parent isLeaf d0..d31 weight'."""
    rng = np.random.default_rng(seed)
    lines = ["%d %d 0 0" % (k, L)]
    frontier = [(0, 0)]                       # (node id, level)
    nid = 0
    while frontier:
        pid, lvl = frontier.pop(0)            # breadth-first like the k-means builder's node numbering is NOT required: ids only need parent < child
        kids = k if lvl < L - 1 or rng.random() < 0.9 else max(1, k // 2)
        for _ in range(kids):
            nid += 1
            leaf = lvl + 1 == L or (lvl + 1 >= 2 and rng.random() < 0.05)       # a few early leaves
            d = rng.integers(0, 256, 32)
            w = 0.0 if (leaf and rng.random() < 0.03) else float(rng.uniform(0.1, 9.0))     # some stopped words (weight 0)
            lines.append("%d %d %s %r" % (pid, 1 if leaf else 0, " ".join(map(str, d)), w))
            if not leaf:
                frontier.append((nid, lvl + 1))
    open(path, "w").write("\n".join(lines) + "\n")
    return nid + 1


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (6, 4, 2), (10, 4, 4), (3, 5, 2)])
def test_bow_transform_matches_oracle(pkg, oracle, synth, tmp_path, k, L, levelsup):
    path = str(tmp_path / "voc.txt")
    nnodes = _write_vocab(path, k, L, seed=k * 10 + L)
    m = pkg.ORBmatcher(0.7)
    voc = pkg.ORBVocabulary(m, path); ref = oracle.Vocabulary(path)
    assert voc.info() == ref.info() and voc.info()["nnodes"] == nnodes
    ex = pkg.ORBextractor(1000, max_size=(752, 480))
    _, _, desc = ex(synth.gen_image(752, 480, 31), (0, 0))
    rng = np.random.default_rng(1)
    desc = np.concatenate([desc, rng.integers(0, 256, (200, 32), dtype=np.uint8)])
    (bi, bv), (fn, fs, fi), w, nd, wt = voc.transform(desc, levelsup)
    (rbi, rbv), (rfn, rfs, rfi), rw, rnd, rwt = ref.transform(desc, levelsup)
    assert np.array_equal(w, rw) and np.array_equal(nd, rnd) and wt.tobytes() == rwt.tobytes()
    assert np.array_equal(bi, rbi) and bv.tobytes() == rbv.tobytes()                       # same summation order -> same doubles
    assert np.array_equal(fn, rfn) and np.array_equal(fs, rfs) and np.array_equal(fi, rfi)
    assert abs(bv.sum() - 1.0) < 1e-12 and np.all(np.diff(bi) > 0) and np.all(np.diff(fn) > 0)
    # the FeatureVector is exactly what SearchByBoW / SearchForTriangulation consume
    assert len(fi) == int((wt > 0).sum())


def test_bow_feeds_search_by_bow(pkg, oracle, synth, tmp_path):
    path = str(tmp_path / "voc.txt")
    _write_vocab(path, 8, 3, seed=5)
    m = pkg.ORBmatcher(0.7); OM = oracle._oracle_matcher_class()()
    voc = pkg.ORBVocabulary(m, path)
    l, r = synth.gen_stereo_pair(752, 480, 100)
    ex = pkg.ORBextractor(1200, max_size=(752, 480))
    _, kl, dl = ex(l, (0, 0)); _, kr, dr = ex(r, (0, 0))
    fvl = voc.transform(dl, 1)[1]; fvr = voc.transform(dr, 1)[1]
    good = np.ones(len(kl), np.uint8)
    a = m.SearchByBoW(kl, dl, good, fvl, kr, dr, fvr, 0.7, True)
    b = OM.SearchByBoW(kl, dl, good, fvl, kr, dr, fvr, 0.7, True)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[0] > 20


def test_vocab_errors(pkg, tmp_path):
    m = pkg.ORBmatcher(0.7)
    with pytest.raises(pkg.OrbError):
        pkg.ORBVocabulary(m, str(tmp_path / "missing.txt"))
    bad = tmp_path / "bad.txt"; bad.write_text("99 3 0 0\n")
    with pytest.raises(pkg.OrbError):
        pkg.ORBVocabulary(m, str(bad))


def test_bow_transform_at_orbvoc_scale(pkg, oracle, synth):
    """8(f).1 at the size of the stock vocabulary: k = 10, L = 6 (1 111 111 nodes, 35.6 MB of node descriptors -- more than the
    32 MB of aggregate L2), levelsup = 4 as Frame::ComputeBoW calls it (Frame.cc:905-918).  Word id, node id and weight of every
    feature of two 1200-feature frames + random rows equal the oracle's; ties between two children resolve to the earlier child."""
    tree = synth.gen_vocabulary(10, 6, seed=7)
    m = pkg.ORBmatcher(0.7)
    voc = pkg.ORBVocabulary(m, tree); ref = oracle.Vocabulary(tree)
    assert voc.info() == ref.info() == dict(k=10, L=6, nnodes=1111111, nwords=1000000)
    ex = pkg.ORBextractor(1200, max_size=(752, 480))
    l, r = synth.gen_stereo_pair(752, 480, 77)
    _, _, dl = ex(l, (0, 0)); _, _, dr = ex(r, (0, 0))
    rng = np.random.default_rng(3)
    # rows that are EXACT copies of tied node descriptors walk into the tie at distance 0
    tied = tree["desc"][np.arange(0, 111111, 997) * 10 + 2][:64]
    desc = np.concatenate([dl, dr, rng.integers(0, 256, (500, 32), dtype=np.uint8), tied])
    (bi, bv), (fn, fs, fi), w, nd, wt = voc.transform(desc, 4)
    (rbi, rbv), (rfn, rfs, rfi), rw, rnd, rwt = ref.transform(desc, 4)
    assert np.array_equal(w, rw) and np.array_equal(nd, rnd) and wt.tobytes() == rwt.tobytes()
    assert np.array_equal(bi, rbi) and bv.tobytes() == rbv.tobytes()
    assert np.array_equal(fn, rfn) and np.array_equal(fs, rfs) and np.array_equal(fi, rfi)
    assert nd.min() >= 11 and nd.max() <= 110                           # levelsup = 4 of L = 6: the 100 nodes of tree level 2
    # the batched device form (ComputeBoW buckets of whole result blocks) agrees
    dd = pkg.DeviceBuffer(desc.nbytes); dd.upload(desc)
    dn = pkg.DeviceBuffer(4 * len(desc))
    assert m.L.orbm_bow_nodes_batch_async(m.h, voc.h, dd.ptr, len(desc), 4, dn.ptr) == 0
    m.sync()
    assert np.array_equal(dn.download(np.int32, len(desc)), rnd)
