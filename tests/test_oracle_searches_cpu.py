"""CPU-only run of the oracle's matcher / Frame restatements (oracle/orbref_match.cpp, orbref_frame.cpp) on a synthetic
stereo pair: every search is driven once with the argument shapes the GPU parity tests use and checked against properties
the reference's loops guarantee (counts, index ranges, one-to-one claims, thresholds).  It keeps the checker itself under
test without a GPU and is the workload tests/test_oracle_sanitize.py replays under ASan/UBSan."""
import numpy as np
import pytest

W, H, NF = 376, 240, 500


@pytest.fixture(scope="module")
def scene(pkg, oracle, synth):
    l, r = synth.gen_stereo_pair(W, H, 321)
    ol, orr = oracle.Extractor(NF), oracle.Extractor(NF)
    _, kl, dl, _ = ol(l, (0, 0)); _, kr, dr, _ = orr(r, (0, 0))
    OM = oracle._oracle_matcher_class()()
    sf = np.array([np.float32(1.2) ** i for i in range(8)], np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    return dict(l=l, r=r, ol=ol, orr=orr, kl=kl, dl=dl, kr=kr, dr=dr, OM=OM, sf=sf, sigma2=(sf * sf).astype(np.float32))


def _fv(pkg, desc, bits):
    return pkg.feature_vector_csr(desc[:, 0].astype(np.int64) & ((1 << bits) - 1))


def _queries(scene, rng, jitter=3.0):
    kl = scene["kl"]; n = len(kl)
    return n, (kl["x"] - 12.0 + rng.normal(0, jitter, n)).astype(np.float32), (kl["y"] + rng.normal(0, jitter / 3, n)).astype(np.float32)


def _one_to_one(m):
    t = m[m >= 0]
    return len(np.unique(t)) == len(t)


def test_scene_is_usable(scene):
    assert len(scene["kl"]) > 300 and len(scene["kr"]) > 300


def test_grid_and_area_query(pkg, oracle, scene):
    f = pkg.FrameView(scene["kr"], scene["dr"], W, H, backend=scene["OM"])
    assert f.placed == len(scene["kr"]) and f.grid_start[-1] == f.placed
    assert sorted(f.grid_idx[:f.placed].tolist()) == list(range(f.placed))
    rng = np.random.default_rng(3)
    for _ in range(60):
        x, y, r = rng.uniform(-20, W + 20), rng.uniform(-20, H + 20), rng.uniform(2, 60)
        lo = int(rng.integers(-1, 6)); hi = lo + int(rng.integers(-2, 3))
        got = scene["OM"].features_in_area(f, x, y, r, lo, hi)
        k = scene["kr"]
        box = (np.abs(k["x"] - np.float32(x)) < np.float32(r)) & (np.abs(k["y"] - np.float32(y)) < np.float32(r))
        if lo > 0 or hi >= 0:                                      # Frame.cc:828: bCheckLevels quirk
            box &= k["octave"] >= lo
            if hi >= 0:
                box &= k["octave"] <= hi
        assert sorted(got.tolist()) == np.nonzero(box)[0].tolist()


@pytest.mark.parametrize("th,fwd,bwd,stereo,ori", [(15, 0, 0, False, True), (7, 1, 0, True, False), (7, 0, 1, True, True)])
def test_search_by_projection_frame(pkg, scene, th, fwd, bwd, stereo, ori):
    rng = np.random.default_rng(th + fwd)
    kr = scene["kr"]
    ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    v = pkg.FrameView(kr, scene["dr"], W, H, uright=ur, backend=scene["OM"])
    n, qu, qv = _queries(scene, rng)
    blocked = rng.random(len(kr)) < 0.05
    cnt, m = scene["OM"].SearchByProjectionFrame(v, cur_blocked=blocked, scale_factors=scene["sf"], valid=rng.random(n) < 0.85, u=qu, v=qv,
                                                 invzc=rng.uniform(0.05, 1.0, n), octave=scene["kl"]["octave"], angle=scene["kl"]["angle"],
                                                 qdesc=scene["dl"], mp_obs=rng.random(n) < 0.9, th=th, forward=bool(fwd), backward=bool(bwd),
                                                 mbf=47.9, check_ori=ori)
    assert cnt >= int((m >= 0).sum()) and m.min() >= -2 and m.max() < max(len(kr), n)   # a claim by a point without observations may be overwritten (ORBmatcher.cc:2565-2567)
    assert cnt > 10


@pytest.mark.parametrize("th,nnratio,stereo", [(1.0, 0.8, False), (3.0, 0.8, True)])
def test_search_by_projection_points(pkg, scene, th, nnratio, stereo):
    rng = np.random.default_rng(int(th * 10))
    kr = scene["kr"]
    ur = np.where(rng.random(len(kr)) < 0.6, kr["x"] - rng.uniform(2, 40, len(kr)), -1).astype(np.float32) if stereo else None
    v = pkg.FrameView(kr, scene["dr"], W, H, uright=ur, backend=scene["OM"])
    n, qu, qv = _queries(scene, rng, jitter=2.0)
    cnt, m = scene["OM"].SearchByProjectionPoints(v, blocked=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"], in_view=rng.random(n) < 0.8,
                                                  px=qu, py=qv, pxr=(qu - rng.uniform(2, 40, n)).astype(np.float32), view_cos=rng.uniform(0.99, 1.0, n),
                                                  level=scene["kl"]["octave"], qdesc=scene["dl"], mp_obs=rng.random(n) < 0.9, th=th, nnratio=nnratio)
    assert cnt >= int((m >= 0).sum()) > 0 and m.min() >= -1


def test_search_for_initialization(pkg, oracle, scene):
    ex = oracle.Extractor(5 * NF)
    _, k1, d1, _ = ex(scene["l"], (0, 1000)); _, k2, d2, _ = ex(scene["r"], (0, 1000))
    OM = scene["OM"]
    f1 = pkg.FrameView(k1, d1, W, H, backend=OM); f2 = pkg.FrameView(k2, d2, W, H, backend=OM)
    n, m12, _ = OM.SearchForInitialization(f1, f2, np.stack([k1["x"], k1["y"]], 1).astype(np.float32), 100, 0.9, True)
    assert n == int((m12 >= 0).sum()) > 10 and _one_to_one(m12)
    assert np.all(k1["octave"][m12 >= 0] == 0)                    # level-0 keypoints only (ORBmatcher.cc:825)


@pytest.mark.parametrize("legacy,coarse,ori", [(False, False, False), (False, True, True), (True, True, True)])
def test_search_for_triangulation(pkg, scene, legacy, coarse, ori):
    rng = np.random.default_rng(5 + legacy)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
    mp1 = rng.random(len(kl)) < 0.3; mp2 = rng.random(len(kr)) < 0.3
    args = dict(k1=kl, d1=dl, has_mp1=mp1, ur1=np.where(rng.random(len(kl)) < 0.5, 5.0, -1.0), fv1=_fv(pkg, dl, 4), k2=kr, d2=dr,
                has_mp2=mp2, ur2=np.where(rng.random(len(kr)) < 0.5, 5.0, -1.0), fv2=_fv(pkg, dr, 4), F12=F12, ep=(900.0, 240.0),
                sf2=scene["sf"], sigma2_2=scene["sigma2"], only_stereo=False, coarse=coarse, check_ori=ori)
    if legacy:
        args["legacy"] = True
    n, m = scene["OM"].SearchForTriangulation(**args)
    assert n == int((m >= 0).sum()) and not np.any(m[mp1] >= 0) and not np.any(mp2[m[m >= 0]])
    if coarse:
        assert n > 5
        d = np.array([oracle_h(dl[i], dr[j]) for i, j in enumerate(m) if j >= 0])
        assert d.max() <= 50                                       # TH_LOW
    if legacy:
        assert _one_to_one(m)


def oracle_h(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


@pytest.mark.parametrize("bits,ori", [(4, True), (7, False)])
def test_search_by_bow_variants(pkg, scene, bits, ori):
    rng = np.random.default_rng(bits)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    OM = scene["OM"]
    good = rng.random(len(kl)) < 0.7
    n, m = OM.SearchByBoW(kkf=kl, dkf=dl, kf_good=good, fvk=_fv(pkg, dl, bits), kf_=kr, df=dr, fvf=_fv(pkg, dr, bits), nnratio=0.8, check_ori=ori)
    assert n == int((m >= 0).sum()) and m.min() >= -2
    g1 = rng.random(len(kl)) < 0.8; g2 = rng.random(len(kr)) < 0.8
    n, m = OM.SearchByBoWKF(k1=kl, d1=dl, good1=g1, fv1=_fv(pkg, dl, bits), k2=kr, d2=dr, good2=g2, fv2=_fv(pkg, dr, bits), nnratio=0.8, check_ori=ori)
    assert n == int((m >= 0).sum()) and _one_to_one(m) and not np.any(m[~g1] >= 0)
    kf = np.concatenate([kr, kl[::-1]]); df = np.concatenate([dr, dl[::-1]])
    out = OM.SearchByBoWFisheye(kkf=kl, dkf=dl, kf_good=good, fvk=_fv(pkg, dl, bits), kf_=kf, df=df, nleft=len(kr), fvf=_fv(pkg, df, bits),
                                nnratio=0.7, check_ori=ori)
    assert out[0] >= 0 and out[1].max() < len(kf)


def test_projection_kf_sim3_fuse_and_sim3(pkg, scene):
    rng = np.random.default_rng(11)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    OM = scene["OM"]
    v = pkg.FrameView(kr, dr, W, H, backend=OM)
    n, qu, qv = _queries(scene, rng)
    lvl = np.clip(kl["octave"] + rng.integers(-1, 2, n), 0, 7)
    blocked = rng.random(len(kr)) < 0.2
    c, m = OM.SearchByProjectionKF(v, blocked=blocked, scale_factors=scene["sf"], valid=rng.random(n) < 0.7, u=qu, v=qv, level=lvl,
                                   angle=kl["angle"], qdesc=dl, th=10, orb_dist=100, check_ori=True)
    assert c == int((m >= 0).sum()) > 5
    c, m = OM.SearchByProjectionSim3(v, matched_in=rng.random(len(kr)) < 0.15, scale_factors=scene["sf"], valid=rng.random(n) < 0.7, u=qu, v=qv,
                                     level=lvl, qdesc=dl, th=8, ratio_hamming=1.5)
    assert c >= 0 and m.max() < max(n, len(kr))
    inv_sigma2 = (1.0 / scene["sigma2"]).astype(np.float32)
    c, b = OM.Fuse(v, scale_factors=scene["sf"], inv_sigma2=inv_sigma2, valid=rng.random(n) < 0.8, u=qu, v=qv,
                   ur=(qu - rng.uniform(2, 40, n)).astype(np.float32), level=lvl, qdesc=dl, th=3.0, chi2_gate=True)
    assert c == int((b >= 0).sum()) and b.max() < len(kr)
    q1 = dict(valid=rng.random(len(kl)) < 0.8, u=(kl["x"] - 12).astype(np.float32), v=kl["y"].copy(), level=np.clip(kl["octave"], 0, 7), qdesc=dl)
    q2 = dict(valid=rng.random(len(kr)) < 0.8, u=(kr["x"] + 12).astype(np.float32), v=kr["y"].copy(), level=np.clip(kr["octave"], 0, 7), qdesc=dr)
    f1 = pkg.FrameView(kl, dl, W, H, backend=OM); f2 = pkg.FrameView(kr, dr, W, H, backend=OM)
    c, m = OM.SearchBySim3(f1, f2, scene["sf"], scene["sf"], q1, q2, 7.5)
    assert c == int((m >= 0).sum()) and _one_to_one(m)


def test_fisheye_projection_searches(pkg, scene):
    rng = np.random.default_rng(13)
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    OM = scene["OM"]; n = len(kl)
    vl = pkg.FrameView(kl, dl, W, H, backend=OM); vr_ = pkg.FrameView(kr, dr, W, H, backend=OM)
    out = OM.SearchByProjectionFrameFisheye(vl, vr_, blocked_l=rng.random(n) < 0.05, blocked_r=rng.random(len(kr)) < 0.05, scale_factors=scene["sf"],
                                            valid=rng.random(n) < 0.85, u=(kl["x"] + rng.normal(0, 3, n)).astype(np.float32),
                                            v=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32), ur=(kl["x"] - 12 + rng.normal(0, 3, n)).astype(np.float32),
                                            vr=(kl["y"] + rng.normal(0, 1, n)).astype(np.float32), octave=kl["octave"], angle=kl["angle"], qdesc=dl,
                                            mp_obs=rng.random(n) < 0.9, th=15, forward=False, backward=False, check_ori=True)
    assert out[0] >= int((out[1] >= 0).sum() + (out[2] >= 0).sum()) and (out[1] >= 0).sum() > 20
    l2r = np.where(rng.random(n) < 0.3, rng.integers(0, len(kr), n), -1).astype(np.int32)
    r2l = np.where(rng.random(len(kr)) < 0.3, rng.integers(0, n, len(kr)), -1).astype(np.int32)
    left = dict(in_view=rng.random(n) < 0.8, px=(kl["x"] + rng.normal(0, 2, n)).astype(np.float32), py=kl["y"].copy(), view_cos=rng.uniform(0.99, 1.0, n), level=kl["octave"])
    right = dict(in_view=rng.random(n) < 0.6, px=(kl["x"] - 12 + rng.normal(0, 2, n)).astype(np.float32), py=kl["y"].copy(), view_cos=rng.uniform(0.99, 1.0, n),
                 level=np.where(rng.random(n) < 0.1, -1, kl["octave"]))
    out = OM.SearchByProjectionPointsFisheye(vl, vr_, blocked_l=rng.random(n) < 0.05, blocked_r=rng.random(len(kr)) < 0.05, l2r=l2r, r2l=r2l,
                                             scale_factors=scene["sf"], left=left, right=right, qdesc=dl, mp_obs=rng.random(n) < 0.9, th=3.0, nnratio=0.9)
    assert out[0] > 10


def test_stereo_matches_and_gated_triangulation(pkg, scene):
    mbf = 47.90639384423901; mb = mbf / 435.2046959714599
    kl, kr, dl, dr = scene["kl"], scene["kr"], scene["dl"], scene["dr"]
    n, ur, dp = scene["OM"].ComputeStereoMatches(scene["ol"], scene["orr"], kl, dl, kr, dr, mb, mbf)
    ok = ur >= 0
    assert n == int(ok.sum()) > 50 and np.all(dp[ok] > 0) and np.all(dp[~ok] == -1) and np.all(kl["x"][ok] - ur[ok] > 0)
    calls = []
    def gate(i1, i2):
        calls.append((i1, i2))
        return (i1 + i2) % 3 != 0
    c, m = scene["OM"].SearchForTriangulationGated(k1=kl, d1=dl, has_mp1=np.zeros(len(kl), bool), fv1=_fv(pkg, dl, 4), k2=kr, d2=dr,
                                                   has_mp2=np.zeros(len(kr), bool), fv2=_fv(pkg, dr, 4), gate=gate, check_ori=False)
    assert c == int((m >= 0).sum()) and all((i + j) % 3 != 0 for i, j in enumerate(m) if j >= 0) and len(calls) >= c
