"""Config C3's device-resident step against the oracle, pair by pair: batched ComputeStereoMatches (row-band Hamming + SAD slide +
parabola on the extractor's pyramids in HBM, then the median cut as a kernel), the FeatureVector buckets of every descriptor
(vocabulary tree descent) and the batched SearchForTriangulation_ bucket match.  Nothing leaves HBM between the extraction and
the match lists; everything compared is bit-exact (the float outputs come from the reference's own expression sequence)."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_bow import _write_vocab

pytestmark = pytest.mark.gpu

MBF = 47.90639384423901
MB = MBF / 435.2046959714599            # Examples/Stereo/EuRoC.yaml:9,28


def _fv(nodes):
    """FeatureVector CSR of DBoW2 (std::map<node, vector<idx>>): nodes ascending, indices in feature order."""
    order = np.argsort(nodes, kind="stable")
    un, start = np.unique(nodes[order], return_index=True)
    return un.astype(np.int32), np.append(start, len(nodes)).astype(np.int32), order.astype(np.int32)


def test_stereo_bow_triangulation_batch(pkg, oracle, synth, tmp_path):
    W, H, NF, P = 752, 480, 1200, 3
    pairs = [synth.gen_stereo_pair(W, H, 500 + i) for i in range(P)]
    imgs = [p[0] for p in pairs] + [p[1] for p in pairs]                    # frames [0,P) left, [P,2P) right
    stride = (W + 63) // 64 * 64
    dev = pkg.DeviceBuffer(2 * P * stride * H)
    for i, im in enumerate(imgs):
        pad = np.zeros((H, stride), np.uint8); pad[:, :W] = im
        dev.upload(pad, offset=i * stride * H)
    arr = (C.c_void_p * (2 * P))(*[dev.ptr + i * stride * H for i in range(2 * P)])
    L = pkg.lib()
    ex = pkg.ORBextractor(NF, max_size=(W, H), max_batch=2 * P)
    mt = pkg.ORBmatcher(0.6)
    assert L.orbm_set_stream(mt.h, L.orbx_stream(ex.h)) == 0
    cap = ex.cap
    ex.enqueue_device(arr, W, H, stride, np.zeros(4 * P, np.int32))
    r = ex.result_device()
    ur = pkg.DeviceBuffer(P * cap * 4); dp = pkg.DeviceBuffer(P * cap * 4); sad = pkg.DeviceBuffer(P * cap * 4); kept = pkg.DeviceBuffer(P * 4)
    assert L.orbm_stereo_batch_async(mt.h, ex.h, 0, P, P, r["kps"], r["desc"], r["counts"], cap, MB, MBF, ur.ptr, dp.ptr, sad.ptr, kept.ptr) == 0, L.orbm_last_error()
    # FeatureVector buckets of every descriptor row of the block
    path = str(tmp_path / "voc.txt")
    _write_vocab(path, 10, 3, seed=7)
    voc = pkg.ORBVocabulary(mt, path); ovoc = oracle.Vocabulary(path)
    nodes = pkg.DeviceBuffer(2 * P * cap * 4)
    assert L.orbm_bow_nodes_batch_async(mt.h, voc.h, r["desc"], 2 * P * cap, 1, nodes.ptr) == 0, L.orbm_last_error()
    # SearchForTriangulation_: KeyFrame 1 = the left images (with their fresh mvuRight), KeyFrame 2 = the right images
    sf = ex.GetScaleFactors(); sig2 = ex.GetScaleSigmaSquares()
    F12 = np.array([0, 0, 0, 0, 0, 0.11, 0, -0.11, 0], np.float32)          # pure x-baseline between identical pinhole cameras
    ep = (1e4, 240.0)
    m12 = pkg.DeviceBuffer(P * cap * 4); nm = pkg.DeviceBuffer(P * 4)
    rc = L.orbm_triangulation_batch_async(mt.h, P, cap, r["kps"], r["desc"], r["counts"], nodes.ptr, ur.ptr,
                                          r["kps"] + P * cap * 28, r["desc"] + P * cap * 32, r["counts"] + 4 * P, nodes.ptr + P * cap * 4, None,
                                          F12.ctypes.data_as(C.c_void_p), ep[0], ep[1], sf.ctypes.data_as(C.c_void_p), sig2.ctypes.data_as(C.c_void_p), 8, 0, 0,
                                          m12.ptr, nm.ptr)
    assert rc == 0, L.orbm_last_error()
    ex.sync()
    res = ex.fetch_all()
    ur_h = ur.download(np.float32, P * cap).reshape(P, cap); dp_h = dp.download(np.float32, P * cap).reshape(P, cap)
    kept_h = kept.download(np.int32, P); nodes_h = nodes.download(np.int32, 2 * P * cap).reshape(2 * P, cap)
    m12_h = m12.download(np.int32, P * cap).reshape(P, cap); nm_h = nm.download(np.int32, P)
    OM = oracle._oracle_matcher_class()()
    for p in range(P):
        ol, orr = oracle.Extractor(NF), oracle.Extractor(NF)
        nl, kl, dl, _ = ol(imgs[p], (0, 0)); nr, kr, dr, _ = orr(imgs[P + p], (0, 0))
        assert res[p][1].tobytes() == kl.tobytes() and res[P + p][1].tobytes() == kr.tobytes()
        n_ref, ur_r, dp_r = OM.ComputeStereoMatches(ol, orr, kl, dl, kr, dr, MB, MBF)
        assert kept_h[p] == n_ref and n_ref > 300, (p, kept_h[p], n_ref)
        assert ur_h[p, :nl].tobytes() == ur_r.tobytes() and dp_h[p, :nl].tobytes() == dp_r.tobytes(), p
        # buckets
        nd_l = ovoc.transform(dl, 1)[3]; nd_r = ovoc.transform(dr, 1)[3]
        assert np.array_equal(nodes_h[p, :nl], nd_l) and np.array_equal(nodes_h[P + p, :nr], nd_r), p
        # triangulation
        n_t, m_ref = OM.SearchForTriangulation(kl, dl, np.zeros(nl, np.uint8), ur_r, _fv(nd_l), kr, dr, np.zeros(nr, np.uint8), None, _fv(nd_r),
                                               F12, ep, sf, sig2, only_stereo=False, coarse=False, check_ori=False)
        assert nm_h[p] == n_t and np.array_equal(m12_h[p, :nl], m_ref), (p, nm_h[p], n_t)
        assert n_t > 100
    ex.close(); mt.close()
