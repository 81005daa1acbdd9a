"""A/B reference kernels, scheduling variants and test knobs: compiled only into liborbslam3_amd_ab.so (-DORBX_AB) and read from the
environment there.  tests/test_ab_child.py runs this directory in a child process with ORB_LIB pointing at that build; the product
library (liborbslam3_amd.so) carries none of these switches (tests/test_abi.py checks that it does not even contain their names)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
from test_gpu_search import scene, _queries   # noqa: F401  (fixture + helper)


def test_window_capacity_overflow_is_retried(pkg, scene, monkeypatch):
    """Frame::GetFeaturesInArea is unbounded; the device windows have a default capacity.  A window that overflows it is run
    again with room for every keypoint of the frame (ORBM_WINDOW_CAP forces a tiny first capacity): same result as before."""
    rng = np.random.default_rng(5)
    kr = scene["kr"]
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng)
    args = dict(cur_blocked=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"], valid=rng.random(n) < 0.85, u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"], qdesc=scene["dl"],
                mp_obs=rng.random(n) < 0.9, th=30, forward=False, backward=False, mbf=47.9, check_ori=True)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(views[1], **args)
    monkeypatch.setenv("ORBM_WINDOW_CAP", "4")
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrame(views[0], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref) and n_ref > 50


def _rand_desc(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


@pytest.mark.parametrize("nq,nt", [(31, 31), (32, 32), (33, 33), (63, 65), (255, 95), (256, 96), (257, 97), (1000, 1000), (1, 4100)])
def test_knn2_popcount_kernel(pkg, oracle, monkeypatch, nq, nt):
    """k_knn2 (the popcount kernel the matrix-core k_knn2_mfma replaced; still the product's route for train sets beyond 2^19 rows)
    forced by ORBM_KNN2_VALU: equal to the oracle on the matrix-core kernel's tile-edge cases, Lowe-ratio epilogue included."""
    import ctypes as C
    rng = np.random.default_rng(nq * 31 + nt)
    q, t = _rand_desc(rng, nq), _rand_desc(rng, nt)
    q[0] = 0; q[-1] = 255; t[0] = 255; t[-1] = 0
    if nt > 40:
        t[nt - 3] = t[5]
    ridx, rdist = oracle.knn2(q, t)
    monkeypatch.setenv("ORBM_KNN2_VALU", "1")
    m = pkg.ORBmatcher()
    idx, dist = m.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    dq, dt = pkg.DeviceBuffer(q.nbytes), pkg.DeviceBuffer(t.nbytes)
    dq.upload(q); dt.upload(t)
    dn = pkg.DeviceBuffer(8); dn.upload(np.array([nq, nt], np.int32))
    di, dd, dg = pkg.DeviceBuffer(nq * 8), pkg.DeviceBuffer(nq * 8), pkg.DeviceBuffer(nq)
    assert m.L.orbm_knn2_ratio_batch_async(m.h, dq.ptr, nq, dn.ptr, dt.ptr, nt, dn.ptr + 4, 1, 0.7, di.ptr, dd.ptr, dg.ptr) == 0
    m.sync()
    good = dg.download(np.uint8, nq)
    want = np.array([1 if (a >= 0 and b >= 0 and float(np.float32(a)) < float(np.float32(b)) * 0.7) else 0 for a, b in rdist.tolist()], np.uint8)
    assert np.array_equal(good, want)


@pytest.mark.parametrize("knob", ["ORBM_TOPK_WAVE", "ORBM_CLAIM_V1"])
def test_first_generation_track_kernels(pkg, oracle, synth, monkeypatch, knob):
    """The wave-per-query top-8 kernel (k_track_topk) and the eight-queries-per-step claim replay (k_track_claim) that k_track_topk16 /
    k_track_claim64 replaced: same final match rows as the oracle on a small batch with blocked slots and a repeated frame."""
    import ctypes as C
    monkeypatch.setenv(knob, "1")
    NB = 4
    imgs = [synth.gen_image(752, 480, 900 + i) for i in range(NB)]
    imgs[3] = imgs[2]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=NB)
    res = ex.extract_batch(imgs, [(0, 1000)] * NB)
    m = pkg.ORBmatcher(0.9)
    OM = oracle._oracle_matcher_class()()
    L = pkg.lib()
    r = ex.result_device(); cap = r["cap"]
    gs = pkg.DeviceBuffer(NB * 3073 * 4); gi = pkg.DeviceBuffer(NB * cap * 4)
    inv_w = np.float32(64) / np.float32(752); inv_h = np.float32(48) / np.float32(480)
    assert L.orbm_grid_build_batch_async(m.h, r["kps"], r["counts"], NB, cap, 0.0, 0.0, float(inv_w), float(inv_h), gs.ptr, gi.ptr) == 0
    rng = np.random.default_rng(3)
    blocked = (rng.random((NB, cap)) < 0.4).astype(np.uint8)
    dblk = pkg.DeviceBuffer(NB * cap); dblk.upload(blocked)
    NP = NB - 1
    dm = pkg.DeviceBuffer(NP * cap * 4); dn = pkg.DeviceBuffer(NP * 4)
    sf = ex.GetScaleFactors()
    rc = L.orbm_search_by_projection_batch_async(m.h, r["kps"], r["desc"], r["counts"], cap, gs.ptr, gi.ptr, 0.0, 0.0, float(inv_w), float(inv_h),
                                                 1, 0, NP, 30.0, sf.ctypes.data_as(C.c_void_p), 8, 1.0, 0.5, dblk.ptr, None, 1, dm.ptr, dn.ptr)
    assert rc == 0, L.orbm_last_error()
    m.sync()
    match = dm.download(np.int32, NP * cap).reshape(NP, cap); nm = dn.download(np.int32, NP)
    for p in range(NP):
        (_, kq, dq), (_, kt, dt) = res[p + 1], res[p]
        nq, nt = len(kq), len(kt)
        n_ref, m_ref = OM.SearchByProjectionFrame(pkg.FrameView(kt, dt, 752, 480, backend=OM), cur_blocked=blocked[p, :nt], scale_factors=sf,
                                                  valid=np.ones(nq, np.uint8), u=kq["x"] + np.float32(1.0), v=kq["y"] + np.float32(0.5),
                                                  invzc=np.zeros(nq, np.float32), octave=kq["octave"], angle=kq["angle"], qdesc=dq,
                                                  mp_obs=np.ones(nq, np.uint8), th=30.0, check_ori=True)
        assert nm[p] == n_ref and np.array_equal(match[p, :nt], m_ref), p
