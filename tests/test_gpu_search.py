"""GPU parity of the flattened ORBmatcher searches (GPU distance phase + host replay) vs the CPU oracle's
reference-structured restatements.  Everything compared is integer / index work -> bit-exact; the stereo outputs
(uright, depth) are floats produced by the same expression sequence and are compared bit-for-bit as well."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# ORB_SCENE_SEEDS=100,101,... repeats every test of this module on more synthetic scenes (a wider parity sweep on the GPU box)
_SEEDS = [int(x) for x in os.environ.get("ORB_SCENE_SEEDS", "100").split(",")]


@pytest.fixture(scope="module", params=_SEEDS)
def scene(request, pkg, oracle, synth):
    """Two views (synthetic rectified stereo pair) extracted on the GPU, plus oracle extractors holding the same pyramids."""
    l, r = synth.gen_stereo_pair(752, 480, request.param)
    exl = pkg.ORBextractor(1200, max_size=(752, 480)); exr = pkg.ORBextractor(1200, max_size=(752, 480))
    _, kl, dl = exl(l, (0, 0)); _, kr, dr = exr(r, (0, 0))
    ol, orr = oracle.Extractor(1200), oracle.Extractor(1200)
    ol(l, (0, 0)); orr(r, (0, 0))
    OM = oracle._oracle_matcher_class()()
    m = pkg.ORBmatcher(0.7)
    return dict(l=l, r=r, exl=exl, exr=exr, kl=kl, dl=dl, kr=kr, dr=dr, ol=ol, orr=orr, OM=OM, m=m,
                sf=exl.GetScaleFactors(), sigma2=exl.GetScaleSigmaSquares())


def test_grid_build_matches_oracle(pkg, scene):
    a = pkg.FrameView(scene["kl"], scene["dl"], 752, 480, backend=scene["m"])
    b = pkg.FrameView(scene["kl"], scene["dl"], 752, 480, backend=scene["OM"])
    assert a.placed == b.placed == len(scene["kl"])
    assert np.array_equal(a.grid_start, b.grid_start) and np.array_equal(a.grid_idx, b.grid_idx)
    # keypoints outside the grid are dropped (posX == 64 after rounding) -- move some to the right edge
    k2 = scene["kl"].copy(); k2["x"][:7] = 751.9; k2["y"][7:11] = 479.8
    a = pkg.FrameView(k2, scene["dl"], 752, 480, backend=scene["m"]); b = pkg.FrameView(k2, scene["dl"], 752, 480, backend=scene["OM"])
    assert a.placed == b.placed < len(k2)
    assert np.array_equal(a.grid_start, b.grid_start) and np.array_equal(a.grid_idx[:a.placed], b.grid_idx[:b.placed])
    # empty frame
    e = pkg.FrameView(scene["kl"][:0], scene["dl"][:0], 752, 480, backend=scene["m"])
    assert e.placed == 0 and e.grid_start[-1] == 0


def test_window_candidates_order_and_distances(pkg, oracle, scene):
    f = pkg.FrameView(scene["kr"], scene["dr"], 752, 480, backend=scene["m"])
    rng = np.random.default_rng(1)
    nq = 300
    qx = rng.uniform(-20, 780, nq).astype(np.float32); qy = rng.uniform(-20, 500, nq).astype(np.float32)
    qr = rng.uniform(2, 90, nq).astype(np.float32)
    lo = rng.integers(-1, 6, nq).astype(np.int32); hi = (lo + rng.integers(-2, 3, nq)).astype(np.int32)
    qd = scene["dl"][:nq]
    cnt, idx, dist = scene["m"].window_candidates(f, qx, qy, qr, lo, hi, qd, cap=f.n)
    for i in range(nq):
        ref = scene["OM"].features_in_area(f, qx[i], qy[i], qr[i], lo[i], hi[i])
        assert cnt[i] == len(ref) and np.array_equal(idx[i, :cnt[i]], ref)
        for c in range(0, cnt[i], 7):
            assert dist[i, c] == oracle.hamming(qd[i], scene["dr"][idx[i, c]])


def _queries(scene, rng, jitter=3.0):
    kl = scene["kl"]; n = len(kl)
    u = (kl["x"] - 12.0 + rng.normal(0, jitter, n)).astype(np.float32)
    v = (kl["y"] + rng.normal(0, jitter / 3, n)).astype(np.float32)
    return n, u, v


@pytest.mark.parametrize("th,fwd,bwd,stereo,ori", [(15, 0, 0, False, True), (7, 0, 0, True, True), (7, 1, 0, True, False),
                                                    (7, 0, 1, True, True), (30, 0, 0, False, True)])
def test_search_by_projection_frame(pkg, scene, th, fwd, bwd, stereo, ori):
    rng = np.random.default_rng(th * 7 + fwd + 2 * bwd)
    kr = scene["kr"]
    ur = None
    if stereo:
        ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32)
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, uright=ur, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng)
    args = dict(cur_blocked=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"], valid=rng.random(n) < 0.85, u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"], qdesc=scene["dl"],
                mp_obs=rng.random(n) < 0.9, th=th, forward=bool(fwd), backward=bool(bwd), mbf=47.9, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrame(views[0], **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    assert n_ref > 50


@pytest.mark.parametrize("th,nnratio,stereo", [(1.0, 0.8, False), (3.0, 0.8, True), (5.0, 0.9, False)])
def test_search_by_projection_points(pkg, scene, th, nnratio, stereo):
    rng = np.random.default_rng(int(th * 10))
    kr = scene["kr"]
    ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, uright=ur, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng, jitter=2.0)
    args = dict(blocked=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"], in_view=rng.random(n) < 0.8, px=u, py=v,
                pxr=(u - rng.uniform(2, 40, n)).astype(np.float32), view_cos=rng.uniform(0.99, 1.0, n),
                level=scene["kl"]["octave"], qdesc=scene["dl"], mp_obs=rng.random(n) < 0.9, th=th, nnratio=nnratio)
    n_gpu, m_gpu = scene["m"].SearchByProjectionPoints(views[0], **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionPoints(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    assert n_ref > 20


@pytest.mark.parametrize("ori", [True, False])
def test_search_for_initialization(pkg, oracle, synth, scene, ori):
    # Tracking.cc:1113: the initialisation extractor uses 5*nFeatures; window 100 (Tracking.cc:2684-2688)
    ex = pkg.ORBextractor(5000, max_size=(752, 480))
    _, k1, d1 = ex(scene["l"], (0, 1000)); _, k2, d2 = ex(scene["r"], (0, 1000))
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    out = []
    for b in (scene["m"], scene["OM"]):
        f1 = pkg.FrameView(k1, d1, 752, 480, backend=b); f2 = pkg.FrameView(k2, d2, 752, 480, backend=b)
        out.append(b.SearchForInitialization(f1, f2, prev, 100, 0.9, ori))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert out[0][0] > 30


def _fv(pkg, desc, bits):
    return pkg.feature_vector_csr(desc[:, 0].astype(np.int64) & ((1 << bits) - 1))


@pytest.mark.parametrize("bits,only_stereo,coarse,ori", [(6, False, False, False), (4, False, True, True), (8, True, False, False)])
def test_search_for_triangulation(pkg, scene, bits, only_stereo, coarse, ori):
    rng = np.random.default_rng(bits)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    # fundamental matrix of a sideways motion + a little rotation (rows of x1' F12 = epipolar lines in image 2)
    F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
    args = dict(k1=kl, d1=dl, has_mp1=rng.random(len(kl)) < 0.3, ur1=np.where(rng.random(len(kl)) < 0.5, 5.0, -1.0),
                fv1=_fv(pkg, dl, bits), k2=kr, d2=dr, has_mp2=rng.random(len(kr)) < 0.3,
                ur2=np.where(rng.random(len(kr)) < 0.5, 5.0, -1.0), fv2=_fv(pkg, dr, bits), F12=F12, ep=(900.0, 240.0),
                sf2=scene["sf"], sigma2_2=scene["sigma2"], only_stereo=only_stereo, coarse=coarse, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchForTriangulation(**args)
    n_ref, m_ref = scene["OM"].SearchForTriangulation(**args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    if coarse:
        assert n_ref > 10


@pytest.mark.parametrize("bits,nnratio,ori", [(6, 0.7, True), (4, 0.75, False), (9, 0.9, True)])
def test_search_by_bow(pkg, scene, bits, nnratio, ori):
    rng = np.random.default_rng(bits + 100)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    args = dict(kkf=kl, dkf=dl, kf_good=rng.random(len(kl)) < 0.7, fvk=_fv(pkg, dl, bits), kf_=kr, df=dr, fvf=_fv(pkg, dr, bits),
                nnratio=nnratio, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchByBoW(**args)
    n_ref, m_ref = scene["OM"].SearchByBoW(**args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)


@pytest.mark.parametrize("th,orb_dist,ori", [(10, 100, True), (3, 64, True), (10, 100, False)])
def test_search_by_projection_kf(pkg, scene, th, orb_dist, ori):
    # relocalisation refinement: Tracking.cc:4309 (th=10, ORBdist=100) and :4334 (th=3, ORBdist=64)
    rng = np.random.default_rng(th + orb_dist)
    kr = scene["kr"]
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng)
    lvl = np.clip(scene["kl"]["octave"] + rng.integers(-1, 2, n), 0, 7)
    args = dict(blocked=rng.random(len(kr)) < 0.2, scale_factors=scene["sf"], valid=rng.random(n) < 0.7, u=u, v=v, level=lvl,
                angle=scene["kl"]["angle"], qdesc=scene["dl"], th=th, orb_dist=orb_dist, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchByProjectionKF(views[0], **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionKF(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref) and n_ref > 10     # (a scene-dependent sanity floor, not parity: th = 3 leaves ~20 matches)


@pytest.mark.parametrize("bits,nnratio,ori", [(6, 0.75, True), (3, 0.9, True), (8, 0.75, False)])
def test_search_by_bow_kf(pkg, scene, bits, nnratio, ori):
    rng = np.random.default_rng(bits + 300)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    args = dict(k1=kl, d1=dl, good1=rng.random(len(kl)) < 0.8, fv1=_fv(pkg, dl, bits), k2=kr, d2=dr,
                good2=rng.random(len(kr)) < 0.8, fv2=_fv(pkg, dr, bits), nnratio=nnratio, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchByBoWKF(**args)
    n_ref, m_ref = scene["OM"].SearchByBoWKF(**args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)


@pytest.mark.parametrize("bits,coarse,ori", [(5, True, True), (6, False, True), (4, True, False)])
def test_search_for_triangulation_legacy(pkg, scene, bits, coarse, ori):
    rng = np.random.default_rng(bits + 500)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
    args = dict(k1=kl, d1=dl, has_mp1=rng.random(len(kl)) < 0.3, ur1=np.where(rng.random(len(kl)) < 0.5, 5.0, -1.0),
                fv1=_fv(pkg, dl, bits), k2=kr, d2=dr, has_mp2=rng.random(len(kr)) < 0.3,
                ur2=np.where(rng.random(len(kr)) < 0.5, 5.0, -1.0), fv2=_fv(pkg, dr, bits), F12=F12, ep=(900.0, 240.0),
                sf2=scene["sf"], sigma2_2=scene["sigma2"], only_stereo=False, coarse=coarse, check_ori=ori, legacy=True)
    n_gpu, m_gpu = scene["m"].SearchForTriangulation(**args)
    n_ref, m_ref = scene["OM"].SearchForTriangulation(**args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    if coarse:
        taken = m_ref[m_ref >= 0]
        assert len(np.unique(taken)) == len(taken)          # vbMatched2 keeps the matching one-to-one


@pytest.mark.parametrize("bits,accept,ori", [(5, 0.5, True), (6, 0.8, False), (3, 0.3, True), (4, 1.0, True), (4, 0.0, False)])
def test_search_for_triangulation_gated(pkg, scene, bits, accept, ori):
    # M10 with a second camera / M12 (ORBmatcher.cc:1632-1821): the camera model's gate is the caller's; a seeded
    # pseudo-random predicate of the pair stands in for KannalaBrandt8::epipolarConstrain_ / matchAndtriangulate.  The gate
    # must be consulted for the same pairs in the same order as by the reference loop (a stateful gate — M12's x3D — sees
    # the same history), and the selected matches must agree.
    rng = np.random.default_rng(bits + 700)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    table = rng.random((257, 263)) < accept
    args = dict(k1=kl, d1=dl, has_mp1=rng.random(len(kl)) < 0.3, fv1=_fv(pkg, dl, bits), k2=kr, d2=dr,
                has_mp2=rng.random(len(kr)) < 0.3, fv2=_fv(pkg, dr, bits), check_ori=ori)
    calls = ([], [])
    out = []
    for b, log in zip((scene["m"], scene["OM"]), calls):
        def gate(i1, i2, log=log):
            log.append((i1, i2))
            return table[i1 % 257, i2 % 263]
        out.append(b.SearchForTriangulationGated(gate=gate, **args))
    assert calls[0] == calls[1] and len(calls[1]) > 50
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    if accept == 0.0:
        assert out[1][0] == 0
    if accept == 1.0:
        assert out[1][0] > 10
        last = {}
        for i1, i2 in calls[1]:
            last[i1] = i2                                    # every accepted call replaces the best: the last one wins
        if not ori:
            assert all(out[1][1][i1] == i2 for i1, i2 in last.items())


@pytest.mark.parametrize("th,ratio", [(8, 1.5), (4, 1.0), (15, 2.0)])
def test_search_by_projection_sim3(pkg, scene, th, ratio):
    # loop closing: LoopClosing.cc uses th = 8, ratioHamming = 1.5 for the Sim3 guided search
    rng = np.random.default_rng(th * 3)
    kr = scene["kr"]
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng)
    lvl = np.clip(scene["kl"]["octave"] + rng.integers(0, 2, n), 0, 7)
    args = dict(matched_in=rng.random(len(kr)) < 0.15, scale_factors=scene["sf"], valid=rng.random(n) < 0.7, u=u, v=v, level=lvl,
                qdesc=scene["dl"], th=th, ratio_hamming=ratio)
    n_gpu, m_gpu = scene["m"].SearchByProjectionSim3(views[0], **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionSim3(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref) and n_ref > 10


@pytest.mark.parametrize("th,chi2,stereo", [(3.0, True, True), (3.0, True, False), (4.0, False, False)])
def test_fuse_search_core(pkg, scene, th, chi2, stereo):
    rng = np.random.default_rng(int(th) + chi2 + 2 * stereo)
    kr = scene["kr"]
    urk = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, uright=urk, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng, jitter=1.0)
    lvl = np.clip(scene["kl"]["octave"] + rng.integers(0, 2, n), 0, 7)
    inv_sigma2 = (1.0 / scene["sigma2"]).astype(np.float32)
    args = dict(scale_factors=scene["sf"], inv_sigma2=inv_sigma2, valid=rng.random(n) < 0.8, u=u, v=v,
                ur=(u - rng.uniform(2, 40, n)).astype(np.float32), level=lvl, qdesc=scene["dl"], th=th, chi2_gate=chi2)
    n_gpu, b_gpu = scene["m"].Fuse(views[0], **args)
    n_ref, b_ref = scene["OM"].Fuse(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(b_gpu, b_ref)


def test_search_by_sim3(pkg, scene):
    rng = np.random.default_rng(77)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    # true correspondences of the synthetic pair are a pure x-shift (disparity): project each set onto the other roughly
    q1 = dict(valid=rng.random(len(kl)) < 0.8, u=(kl["x"] - 20 + rng.normal(0, 4, len(kl))).astype(np.float32), v=kl["y"].copy(),
              level=np.clip(kl["octave"] + rng.integers(0, 2, len(kl)), 0, 7), qdesc=dl)
    q2 = dict(valid=rng.random(len(kr)) < 0.8, u=(kr["x"] + 20 + rng.normal(0, 4, len(kr))).astype(np.float32), v=kr["y"].copy(),
              level=np.clip(kr["octave"] + rng.integers(0, 2, len(kr)), 0, 7), qdesc=dr)
    out = []
    for b in (scene["m"], scene["OM"]):
        f1 = pkg.FrameView(kl, dl, 752, 480, backend=b); f2 = pkg.FrameView(kr, dr, 752, 480, backend=b)
        out.append(b.SearchBySim3(f1, f2, scene["sf"], scene["sf"], q1, q2, 7.5))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    assert out[0][0] > 20


@pytest.mark.parametrize("th,fwd,bwd,ori", [(15, 0, 0, True), (7, 1, 0, True), (7, 0, 1, False)])
def test_search_by_projection_frame_fisheye(pkg, scene, th, fwd, bwd, ori):
    # fisheye stereo (TUM-VI, BASELINE config C4): the current frame has separate left / right keypoint sets and grids
    rng = np.random.default_rng(900 + th + fwd)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    n = len(kl)
    args = dict(blocked_l=rng.random(len(kl)) < 0.05, blocked_r=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"],
                valid=rng.random(n) < 0.85, u=(kl["x"] + rng.normal(0, 3, n)).astype(np.float32), v=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32),
                ur=(kl["x"] - 20 + rng.normal(0, 3, n)).astype(np.float32), vr=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32),
                octave=kl["octave"], angle=kl["angle"], qdesc=dl, mp_obs=rng.random(n) < 0.9, th=th, forward=bool(fwd), backward=bool(bwd), check_ori=ori)
    out = []
    for b in (scene["m"], scene["OM"]):
        vl = pkg.FrameView(kl, dl, 752, 480, backend=b); vr_ = pkg.FrameView(kr, dr, 752, 480, backend=b)
        out.append(b.SearchByProjectionFrameFisheye(vl, vr_, **args))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert (out[0][1] >= 0).sum() > 50 and (out[0][2] >= 0).sum() > 20


@pytest.mark.parametrize("th,nnratio", [(1.0, 0.8), (3.0, 0.9)])
def test_search_by_projection_points_fisheye(pkg, scene, th, nnratio):
    rng = np.random.default_rng(950 + int(th))
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    n = len(kl)
    l2r = np.where(rng.random(len(kl)) < 0.3, rng.integers(0, len(kr), len(kl)), -1).astype(np.int32)
    r2l = np.where(rng.random(len(kr)) < 0.3, rng.integers(0, len(kl), len(kr)), -1).astype(np.int32)
    left = dict(in_view=rng.random(n) < 0.8, px=(kl["x"] + rng.normal(0, 2, n)).astype(np.float32), py=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32),
                view_cos=rng.uniform(0.99, 1.0, n), level=kl["octave"])
    right = dict(in_view=rng.random(n) < 0.6, px=(kl["x"] - 20 + rng.normal(0, 2, n)).astype(np.float32), py=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32),
                 view_cos=rng.uniform(0.99, 1.0, n), level=np.where(rng.random(n) < 0.1, -1, kl["octave"]))
    args = dict(blocked_l=rng.random(len(kl)) < 0.05, blocked_r=rng.random(len(kr)) < 0.05, l2r=l2r, r2l=r2l, scale_factors=scene["sf"],
                left=left, right=right, qdesc=dl, mp_obs=rng.random(n) < 0.9, th=th, nnratio=nnratio)
    out = []
    for b in (scene["m"], scene["OM"]):
        vl = pkg.FrameView(kl, dl, 752, 480, backend=b); vr_ = pkg.FrameView(kr, dr, 752, 480, backend=b)
        out.append(b.SearchByProjectionPointsFisheye(vl, vr_, **args))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert out[0][0] > 50


@pytest.mark.parametrize("bits,ori", [(5, True), (7, False)])
def test_search_by_bow_fisheye(pkg, scene, bits, ori):
    rng = np.random.default_rng(bits + 700)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    # the frame's descriptor block is vconcat(left, right) (Frame.cc:1428); the keyframe is the left set here
    kf = np.concatenate([kr, kl[::-1]]); df = np.concatenate([dr, dl[::-1]])
    args = dict(kkf=kl, dkf=dl, kf_good=rng.random(len(kl)) < 0.8, fvk=_fv(pkg, dl, bits), kf_=kf, df=df, nleft=len(kr), fvf=_fv(pkg, df, bits),
                nnratio=0.7, check_ori=ori)
    a = scene["m"].SearchByBoWFisheye(**args); b = scene["OM"].SearchByBoWFisheye(**args)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[0] > 10


def test_compute_stereo_matches(pkg, scene):
    # EuRoC stereo: bf = 47.906, fx = 435.2 -> mb = bf/fx (Examples/Stereo/EuRoC.yaml:9,28)
    mbf = 47.90639384423901; mb = mbf / 435.2046959714599
    n_gpu, ur_g, dp_g = scene["m"].ComputeStereoMatches(scene["exl"], scene["exr"], scene["kl"], scene["dl"], scene["kr"], scene["dr"], mb, mbf)
    n_ref, ur_r, dp_r = scene["OM"].ComputeStereoMatches(scene["ol"], scene["orr"], scene["kl"], scene["dl"], scene["kr"], scene["dr"], mb, mbf)
    assert n_gpu == n_ref and n_ref > 300
    assert ur_g.tobytes() == ur_r.tobytes() and dp_g.tobytes() == dp_r.tobytes()
    ok = ur_g >= 0
    d = scene["kl"]["x"][ok] - ur_g[ok]
    assert np.all(d > 0) and np.all(d < mbf / mb)


def test_searches_handle_empty_inputs(pkg, scene):
    m, kl, dl = scene["m"], scene["kl"], scene["dl"]
    empty = pkg.FrameView(kl[:0], dl[:0], 752, 480, backend=m)
    full = pkg.FrameView(kl, dl, 752, 480, backend=m)
    z = np.zeros(0)
    n, match = m.SearchByProjectionFrame(full, np.zeros(len(kl)), scene["sf"], z, z, z, z, z, z, dl[:0], z, 15)
    assert n == 0 and np.all(match == -1)
    n, m12, _ = m.SearchForInitialization(empty, full, np.zeros((0, 2), np.float32), 100, 0.9)
    assert n == 0 and len(m12) == 0


def test_device_resident_tracking_window_batch(pkg, oracle, synth):
    """Batched device-resident grid build + tracking window (no host round trip) against the host entry points that are
    themselves pinned to the oracle: same grids, same first-minimum best / runner-up per keypoint."""
    import ctypes as C
    imgs = [synth.gen_image(752, 480, 500 + i) for i in range(4)]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=4)
    res = ex.extract_batch(imgs, [(0, 1000)] * 4)
    m = pkg.ORBmatcher(0.9)
    L = pkg.lib()
    r = ex.result_device(); cap = r["cap"]
    gs = pkg.DeviceBuffer(4 * 3073 * 4); gi = pkg.DeviceBuffer(4 * cap * 4)
    inv_w = np.float32(64) / np.float32(752); inv_h = np.float32(48) / np.float32(480)
    assert L.orbm_grid_build_batch_async(m.h, r["kps"], r["counts"], 4, cap, 0.0, 0.0, float(inv_w), float(inv_h), gs.ptr, gi.ptr) == 0
    bi = pkg.DeviceBuffer(3 * cap * 4); bd = pkg.DeviceBuffer(3 * cap * 4); sd = pkg.DeviceBuffer(3 * cap * 4)
    sf = ex.GetScaleFactors()
    assert L.orbm_track_window_batch_async(m.h, r["kps"], r["desc"], r["counts"], cap, gs.ptr, gi.ptr, 0.0, 0.0, float(inv_w), float(inv_h),
                                           1, 0, 3, 15.0, sf.ctypes.data_as(C.c_void_p), 8, 2.0, -1.0, bi.ptr, bd.ptr, sd.ptr) == 0
    m.sync()
    g_start = gs.download(np.int32, 4 * 3073).reshape(4, 3073); g_idx = gi.download(np.int32, 4 * cap).reshape(4, cap)
    best_i = bi.download(np.int32, 3 * cap).reshape(3, cap); best_d = bd.download(np.int32, 3 * cap).reshape(3, cap)
    sec_d = sd.download(np.int32, 3 * cap).reshape(3, cap)
    OM = oracle._oracle_matcher_class()()
    for p in range(3):
        (_, kq, dq), (_, kt, dt) = res[p + 1], res[p]
        fo = pkg.FrameView(kt, dt, 752, 480, backend=OM)
        assert np.array_equal(g_start[p], fo.grid_start) and np.array_equal(g_idx[p, :fo.placed], fo.grid_idx[:fo.placed])
        f = pkg.FrameView(kt, dt, 752, 480, backend=m)
        qr = (np.float32(15.0) * sf[kq["octave"]]).astype(np.float32)
        cnt, idx, dist = m.window_candidates(f, kq["x"] + np.float32(2.0), kq["y"] + np.float32(-1.0), qr, kq["octave"] - 1, kq["octave"] + 1,
                                             dq, cap=f.n)
        for q in range(len(kq)):
            if cnt[q] == 0:
                assert best_i[p, q] == -1 and best_d[p, q] == 256
                continue
            d = dist[q, :cnt[q]]
            b = int(np.argmin(d))                                   # first minimum in candidate order
            assert best_i[p, q] == idx[q, b] and best_d[p, q] == d[b]
            rest = np.delete(d, b)
            assert sec_d[p, q] == (int(rest.min()) if len(rest) else 256)


@pytest.mark.parametrize("check_ori,th", [(True, 15.0), (False, 15.0), (True, 45.0)])
def test_batched_search_by_projection_final_matches(pkg, oracle, synth, check_ori, th):
    """M4 end to end on the device for a batch of frame pairs (claims in query order, TH_HIGH, rotation histogram + three-maxima
    cull): the final match row and count of every pair equal orbm_search_by_projection_frame (single-frame host replay) AND the
    oracle, entry for entry (th = 45: windows of more than 16 grid columns and more than 64 grid entries, i.e. several passes of the
    16-lanes-per-query list kernel).  The batch holds an EMPTY frame (both as the searched and as the searching one), pairs whose searched
    frame has 35 % / 97 % of its slots blocked (the second drives queries through all eight listed candidates into the in-place
    rescan), queries without observations (their slot can be taken again), and a frame matched against itself (distance-0 ties)."""
    import ctypes as C
    NB = 10
    imgs = [synth.gen_image(752, 480, 700 + i) for i in range(NB)]
    imgs[9] = imgs[8]                                                          # pair 8 matches a frame against itself
    imgs[3] = np.full((480, 752), 128, np.uint8)                               # no corner anywhere: an empty frame
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=NB)
    res = ex.extract_batch(imgs, [(0, 1000)] * NB)
    assert len(res[3][1]) == 0
    m = pkg.ORBmatcher(0.9)
    OM = oracle._oracle_matcher_class()()
    L = pkg.lib()
    r = ex.result_device(); cap = r["cap"]
    gs = pkg.DeviceBuffer(NB * 3073 * 4); gi = pkg.DeviceBuffer(NB * cap * 4)
    inv_w = np.float32(64) / np.float32(752); inv_h = np.float32(48) / np.float32(480)
    assert L.orbm_grid_build_batch_async(m.h, r["kps"], r["counts"], NB, cap, 0.0, 0.0, float(inv_w), float(inv_h), gs.ptr, gi.ptr) == 0
    rng = np.random.default_rng(11)
    blocked = np.zeros((NB, cap), np.uint8); obs = np.ones((NB, cap), np.uint8)
    blocked[1] = rng.random(cap) < 0.35
    blocked[6] = rng.random(cap) < 0.97
    obs[2] = rng.random(cap) < 0.5
    obs[7] = 0
    dblk = pkg.DeviceBuffer(NB * cap); dobs = pkg.DeviceBuffer(NB * cap)
    dblk.upload(blocked); dobs.upload(obs)
    NP = NB - 1
    dm = pkg.DeviceBuffer(NP * cap * 4); dn = pkg.DeviceBuffer(NP * 4)
    sf = ex.GetScaleFactors()
    dx, dy = 2.0, -1.0
    for _ in (0,):                                                             # pair p: frame p+1 (queries) searches frame p
        rc = L.orbm_search_by_projection_batch_async(m.h, r["kps"], r["desc"], r["counts"], cap, gs.ptr, gi.ptr, 0.0, 0.0, float(inv_w), float(inv_h),
                                                     1, 0, NP, th, sf.ctypes.data_as(C.c_void_p), 8, dx, dy, dblk.ptr, dobs.ptr, int(check_ori),
                                                     dm.ptr, dn.ptr)
        assert rc == 0, L.orbm_last_error()
    m.sync()
    match = dm.download(np.int32, NP * cap).reshape(NP, cap); nm = dn.download(np.int32, NP)
    saw_pruned = False
    for p in range(NP):
        (_, kq, dq), (_, kt, dt) = res[p + 1], res[p]
        nq, nt = len(kq), len(kt)
        args = dict(cur_blocked=blocked[p, :nt], scale_factors=sf, valid=np.ones(nq, np.uint8), u=kq["x"] + np.float32(dx), v=kq["y"] + np.float32(dy),
                    invzc=np.zeros(nq, np.float32), octave=kq["octave"], angle=kq["angle"], qdesc=dq, mp_obs=obs[p + 1, :nq], th=th, check_ori=check_ori)
        if nt == 0:
            assert nm[p] == 0 and np.all(match[p] == -1)
            continue
        n_ref, m_ref = OM.SearchByProjectionFrame(pkg.FrameView(kt, dt, 752, 480, backend=OM), **args)
        n_host, m_host = m.SearchByProjectionFrame(pkg.FrameView(kt, dt, 752, 480, backend=m), **args)
        assert n_host == n_ref and np.array_equal(m_host, m_ref)
        assert nm[p] == n_ref, (p, nm[p], n_ref)
        assert np.array_equal(match[p, :nt], m_ref), (p, np.flatnonzero(match[p, :nt] != m_ref)[:10])
        assert np.all(match[p, nt:] == -1)
        saw_pruned = saw_pruned or bool((m_ref == -2).any())
    assert nm[8] > 300                                                          # the self-pair matches well
    if check_ori:
        assert saw_pruned


def test_pruned_slots_are_marked(pkg, scene):
    """M4 / M5 write into an existing mvpMapPoints: a slot that was assigned and then culled by the rotation-consistency check
    comes back as -2 (the reference NULLs it, ORBmatcher.cc:2700-2708), distinct from -1 = never touched; product == oracle."""
    rng = np.random.default_rng(11)
    kr = scene["kr"]
    views = [pkg.FrameView(kr, scene["dr"], 752, 480, backend=b) for b in (scene["m"], scene["OM"])]
    n, u, v = _queries(scene, rng, jitter=6.0)
    ang = (scene["kl"]["angle"] + rng.uniform(0, 360, n)).astype(np.float32) % 360          # scattered rotations: most bins get culled
    args = dict(cur_blocked=np.zeros(len(kr), bool), scale_factors=scene["sf"], valid=np.ones(n, bool), u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=ang, qdesc=scene["dl"],
                mp_obs=np.ones(n, bool), th=15, forward=False, backward=False, mbf=47.9, check_ori=True)
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrame(views[0], **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(views[1], **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    assert (m_ref == -2).sum() > 10 and (m_ref >= 0).sum() == n_ref


@pytest.mark.parametrize("th,fwd,bwd,stereo,ori,pblock", [(15, 0, 0, False, True, 0.05), (7, 0, 0, True, True, 0.05), (7, 1, 0, True, False, 0.3),
                                                           (30, 0, 0, False, True, 0.97), (7, 0, 1, True, True, 0.6)])
def test_resident_frame_search_by_projection_frame(pkg, scene, th, fwd, bwd, stereo, ori, pblock):
    """M4 against a frame kept in HBM: only the queries travel and only the 8 best candidates per query come back; the claim
    replay must still equal the oracle's -- also when most slots are blocked, so that cut lists run dry and the call falls
    back to the full candidate lists by itself (pblock 0.97)."""
    rng = np.random.default_rng(th * 11 + fwd + 2 * bwd)
    kr = scene["kr"]
    ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    view = pkg.FrameView(kr, scene["dr"], 752, 480, uright=ur, backend=scene["OM"])
    res = pkg.ResidentFrame(scene["m"], view)
    n, u, v = _queries(scene, rng)
    args = dict(cur_blocked=rng.random(len(kr)) < pblock, scale_factors=scene["sf"], valid=rng.random(n) < 0.85, u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"], qdesc=scene["dl"],
                mp_obs=rng.random(n) < 0.9, th=th, forward=bool(fwd), backward=bool(bwd), mbf=47.9, check_ori=ori)
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrameResident(res, **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(view, **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    res.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_resident_frame_searches_on_tie_heavy_descriptors(pkg, scene, seed):
    """The resident searches return the 8 best candidates per window in (distance, VISITING ORDER) rank; the replay relies on that
    order whenever distances tie.  Here the searched frame's descriptors are drawn from only 12 distinct rows (and the queries from
    the same 12), so that nearly every window holds several candidates at the same distance -- often more than eight: M4 and M3 on
    the resident frame must still equal the oracle entry for entry, over several seeds."""
    rng = np.random.default_rng(1000 + seed)
    kr = scene["kr"]; n_t = len(kr)
    pool = rng.integers(0, 256, (12, 32), dtype=np.uint8)
    dt = pool[rng.integers(0, 12, n_t)]
    n, u, v = _queries(scene, rng, jitter=4.0)
    dq = pool[rng.integers(0, 12, n)]
    view = pkg.FrameView(kr, dt, 752, 480, backend=scene["OM"])
    res = pkg.ResidentFrame(scene["m"], view)
    args = dict(cur_blocked=rng.random(n_t) < 0.3, scale_factors=scene["sf"], valid=rng.random(n) < 0.9, u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"], qdesc=dq,
                mp_obs=rng.random(n) < 0.8, th=30, forward=False, backward=False, mbf=47.9, check_ori=True)
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrameResident(res, **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(view, **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref) and n_ref > 30, (n_gpu, n_ref, np.flatnonzero(m_gpu != m_ref)[:8])
    args3 = dict(blocked=rng.random(n_t) < 0.3, scale_factors=scene["sf"], in_view=rng.random(n) < 0.9, px=u, py=v,
                 pxr=(u - rng.uniform(2, 40, n)).astype(np.float32), view_cos=rng.uniform(0.99, 1.0, n),
                 level=scene["kl"]["octave"], qdesc=dq, mp_obs=rng.random(n) < 0.8, th=8.0, nnratio=0.9)
    n_gpu, m_gpu = scene["m"].SearchByProjectionPointsResident(res, **args3)
    n_ref, m_ref = scene["OM"].SearchByProjectionPoints(view, **args3)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    res.close()


@pytest.mark.parametrize("th,nnratio,stereo,pblock", [(1.0, 0.8, False, 0.05), (3.0, 0.8, True, 0.05), (5.0, 0.9, False, 0.9), (8.0, 0.9, True, 0.5)])
def test_resident_frame_search_by_projection_points(pkg, scene, th, nnratio, stereo, pblock):
    rng = np.random.default_rng(int(th * 10) + 1)
    kr = scene["kr"]
    ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    view = pkg.FrameView(kr, scene["dr"], 752, 480, uright=ur, backend=scene["OM"])
    res = pkg.ResidentFrame(scene["m"], view)
    n, u, v = _queries(scene, rng, jitter=2.0)
    args = dict(blocked=rng.random(len(kr)) < pblock, scale_factors=scene["sf"], in_view=rng.random(n) < 0.8, px=u, py=v,
                pxr=(u - rng.uniform(2, 40, n)).astype(np.float32), view_cos=rng.uniform(0.99, 1.0, n),
                level=scene["kl"]["octave"], qdesc=scene["dl"], mp_obs=rng.random(n) < 0.9, th=th, nnratio=nnratio)
    n_gpu, m_gpu = scene["m"].SearchByProjectionPointsResident(res, **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionPoints(view, **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref)
    res.close()


def test_resident_frame_adopts_extractor_results(pkg, scene, synth):
    """A resident frame made from device pointers (one frame of an extractor's result block, no copy of keypoints or
    descriptors) gives the same search result as one uploaded from the host arrays."""
    import ctypes as C
    rng = np.random.default_rng(3)
    ex = scene["exr"]
    ex(scene["r"], (0, 0))                                      # the right view's results are the extractor's current block
    r = ex.result_device()
    kr, dr = scene["kr"], scene["dr"]
    res_dev = pkg.ResidentFrame(scene["m"], device=dict(kps=r["kps"], desc=r["desc"], n=len(kr)), width=752, height=480)
    view = pkg.FrameView(kr, dr, 752, 480, backend=scene["OM"])
    n, u, v = _queries(scene, rng)
    args = dict(cur_blocked=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"], valid=rng.random(n) < 0.85, u=u, v=v,
                invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"], qdesc=scene["dl"],
                mp_obs=rng.random(n) < 0.9, th=15, forward=False, backward=False, mbf=47.9, check_ori=True)
    n_gpu, m_gpu = scene["m"].SearchByProjectionFrameResident(res_dev, **args)
    n_ref, m_ref = scene["OM"].SearchByProjectionFrame(view, **args)
    assert n_gpu == n_ref and np.array_equal(m_gpu, m_ref) and n_ref > 50
    res_dev.close()


def test_batched_search_by_projection_refuses_more_slots_than_the_claim_replay_holds(pkg):
    """k_track_claim64 keeps 4 B per keypoint slot of the searched frame in LDS (64 KB): ~16 000 slots per frame is the documented limit."""
    m = pkg.ORBmatcher()
    one = pkg.DeviceBuffer(256)
    sf = np.ones(8, np.float32)
    import ctypes as C
    rc = m.L.orbm_search_by_projection_batch_async(m.h, one.ptr, one.ptr, one.ptr, 20000, one.ptr, one.ptr, 0.0, 0.0, 0.1, 0.1, 1, 0, 1, 15.0,
                                                   sf.ctypes.data_as(C.c_void_p), 8, 0.0, 0.0, None, None, 1, one.ptr, one.ptr)
    assert rc < 0 and b"LDS" in m.L.orbm_last_error()
