"""Known-answer tests pinning the CPU oracle (the reference ships no golden vectors: SURVEY F6, 8(c)).
Every expected value here is hand-derivable from the reference source / OpenCV's published algorithm."""
import hashlib

import numpy as np


def test_tables_match_reference_constructor(oracle):
    # ORBextractor.cc:509-526 (feature split) and :542-570 (umax), SURVEY 8(a) E0
    t = oracle.Extractor(1000, 1.2, 8, 20, 7).tables()
    assert t["nfeat"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert t["sf"][0] == 1.0 and abs(t["sf"][7] - 1.2 ** 7) < 1e-5
    t = oracle.Extractor(1200, 1.2, 8, 20, 7).tables()
    assert t["nfeat"].tolist() == [261, 217, 181, 151, 126, 105, 87, 72]
    t = oracle.Extractor(4000, 1.2, 8, 20, 7).tables()
    assert t["nfeat"].tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    # patch sizes int(31*sf): 31,37,44,53,64,77,92,111
    sf = oracle.Extractor(1000).tables()["sf"]
    assert [int(np.float32(31) * s) for s in sf] == [31, 37, 44, 53, 64, 77, 92, 111]


def test_pattern_table_checksum(oracle):
    p = oracle.pattern()
    assert hashlib.sha256(p.astype(np.int8).tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    assert p[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]          # ORBextractor.cc:208-209
    assert p[-4:].tolist() == [-1, -6, 0, -11]                    # ORBextractor.cc:463
    r = np.hypot(p[0::2].astype(float), p[1::2].astype(float)).max()
    assert r < 18.4                                                # max sampling radius


def test_level_sizes(oracle, synth):
    ex = oracle.Extractor(1000)
    n, *_ = ex(synth.gen_image(752, 480, 3))
    assert n > 0
    assert [ex.level_size(l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278),
                                                    (363, 231), (302, 193), (252, 161), (210, 134)]


def test_hamming_kats(oracle):
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, o) == 256
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


def test_gauss7_kernel_and_rounding(oracle):
    # impulse response = outer([18,34,48,56,48,34,18]) * 255 / 65536 rounded half up
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    img = np.zeros((21, 21), np.uint8); img[10, 10] = 255
    out = oracle.gauss7(img)
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], exp) and out.sum() == exp.sum()
    # constant image stays constant, incl. reflect-101 borders
    assert np.all(oracle.gauss7(np.full((9, 13), 77, np.uint8)) == 77)
    # reflect-101 at a corner: brute-force check
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (12, 10), dtype=np.uint8)
    pad = np.pad(im.astype(np.int64), 3, mode="reflect")
    ref = np.zeros_like(im)
    for y in range(12):
        for x in range(10):
            ref[y, x] = (int((pad[y:y + 7, x:x + 7] * np.outer(k, k)).sum()) + 32768) >> 16
    assert np.array_equal(oracle.gauss7(im), ref)


def test_fast_atan2(oracle):
    L = oracle.lib()
    assert L.orbref_fast_atan2(0.0, 0.0) == 0.0
    for y, x in [(1, 1), (1, -1), (-1, -1), (-1, 1), (0, 5), (5, 0), (0, -5), (-5, 0), (3, 7), (-120000, 64000)]:
        a = L.orbref_fast_atan2(float(y), float(x))
        e = np.degrees(np.arctan2(y, x)) % 360.0
        assert abs(a - e) < 0.02 or abs(abs(a - e) - 360) < 0.02      # polynomial accuracy ~0.3 deg worst-case; these are tight
    assert 0.0 <= L.orbref_fast_atan2(-1e-3, 1.0) < 360.0


def test_fast_single_dot(oracle):
    # isolated bright pixel on flat background: ring all darker by (I-bg) -> score = I-bg-1
    img = np.full((15, 15), 50, np.uint8); img[7, 7] = 150
    kp = oracle.fast(img, 20)
    assert kp.tolist() == [[7, 7, 99]]
    assert oracle.fast_score(img, 7, 7) == 99
    assert len(oracle.fast(np.full((30, 30), 128, np.uint8), 7)) == 0
    # below threshold: I-bg = 20 is NOT > 20
    img[7, 7] = 70
    assert len(oracle.fast(img, 20)) == 0 and oracle.fast(img, 7).tolist() == [[7, 7, 19]]


def test_fast_matches_bruteforce_definition(oracle):
    """FAST-9/16 + 3x3 strict NMS, score=max(A,B)-1, against a slow independent numpy statement."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (8, 9)).astype(np.float64)
    img = np.kron(base, np.ones((5, 5)))[:38, :42] + rng.normal(0, 6, (38, 42))
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    h, w = img.shape
    for thr in (7, 20):
        S = np.zeros((h, w), np.int64)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                v = int(img[y, x]); d = [v - int(img[y + dy[k], x + dx[k]]) for k in range(16)]
                A = max(min(d[(s + t) % 16] for t in range(9)) for s in range(16))
                B = max(min(-d[(s + t) % 16] for t in range(9)) for s in range(16))
                if max(A, B) > thr:
                    S[y, x] = max(A, B) - 1
        exp = []
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                if S[y, x] > 0:
                    nb = S[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                    if S[y, x] > nb.max():
                        exp.append([x, y, int(S[y, x])])
        assert oracle.fast(img, thr).tolist() == exp


def test_resize_linear_fixed_point(oracle):
    assert np.all(oracle.resize_linear(np.full((40, 60), 201, np.uint8), 50, 33) == 201)
    # independent numpy statement of the published fixed-point algorithm
    rng = np.random.default_rng(2)
    src = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    dw, dh = 44, 31
    sh, sw = src.shape

    def coeffs(dn, sn, clamp):
        sc = 1.0 / (dn / sn)
        f = ((np.arange(dn) + 0.5) * sc - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64); f = (f - s).astype(np.float32)
        if clamp:
            f[s < 0] = 0; s[s < 0] = 0
            f[s >= sn - 1] = 0; s[s >= sn - 1] = sn - 1
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, a0, a1
    sx, xa0, xa1 = coeffs(dw, sw, True)
    sy, ya0, ya1 = coeffs(dh, sh, False)
    s = src.astype(np.int64)
    H = s[:, sx] * xa0 + s[:, np.minimum(sx + 1, sw - 1)] * xa1
    r0 = np.clip(sy, 0, sh - 1); r1 = np.clip(sy + 1, 0, sh - 1)
    exp = ((((ya0[:, None] * (H[r0] >> 4)) >> 16) + ((ya1[:, None] * (H[r1] >> 4)) >> 16) + 2) >> 2)
    assert np.array_equal(oracle.resize_linear(src, dw, dh), exp.astype(np.uint8))


def test_distribute_small_cases(oracle):
    # fewer candidates than N: every candidate survives (each ends in its own node)
    pts = np.array([[10, 10, 50], [300, 40, 60], [700, 400, 70], [20, 300, 80]], np.int32)
    sel = oracle.distribute(pts, 16, 736, 16, 464, 10)
    assert sorted(sel.tolist()) == [0, 1, 2, 3]
    # two candidates in the same final cell -> strictly larger response wins, first on ties
    pts = np.array([[10, 10, 50], [11, 11, 90], [400, 10, 30]], np.int32)
    assert sorted(oracle.distribute(pts, 16, 736, 16, 464, 2).tolist()) == [1, 2]
    pts = np.array([[10, 10, 90], [11, 11, 90], [400, 10, 30]], np.int32)
    assert sorted(oracle.distribute(pts, 16, 736, 16, 464, 2).tolist()) == [0, 2]
    assert len(oracle.distribute(np.zeros((0, 3), np.int32), 16, 736, 16, 464, 5)) == 0


def test_three_maxima(oracle):
    c = np.zeros(30, np.int32); c[3] = 10; c[7] = 9; c[20] = 1
    assert oracle.three_maxima(c).tolist() == [3, 7, -1] or oracle.three_maxima(c).tolist() == [3, 7, 20]
    # 1 < 0.1*10 is False (1 < 1.0f false) -> third kept
    assert oracle.three_maxima(c).tolist() == [3, 7, 20]
    c[20] = 0
    assert oracle.three_maxima(c).tolist() == [3, 7, -1]
    c[:] = 0; c[5] = 4; c[6] = 4                       # ties: first wins the top slot
    assert oracle.three_maxima(c).tolist() == [5, 6, -1]


def test_extract_degenerate_inputs(oracle, synth):
    ex = oracle.Extractor(1000)
    n, kps, desc, mono = ex(synth.gen_image(752, 480, 0, kind="constant"))
    assert n == 0 and mono == 0
    n, kps, desc, mono = ex(synth.gen_image(752, 480, 9, kind="lowcontrast"), (0, 0))
    assert n >= 0 and mono == n          # lap {0,0}: only x==0 would be "lapping"; none are (x>=19)
    # output ordering rule (ORBextractor.cc:1644-1653): mono call {0,1000} on width<=1000 -> fully reversed
    img = synth.gen_image(752, 480, 4)
    n1, k1, d1, m1 = ex(img, (0, 1000))
    n0, k0, d0, m0 = ex(img, (0, 0))
    assert n0 == n1 and m1 == 0 and m0 == n0
    assert np.array_equal(k1, k0[::-1]) and np.array_equal(d1, d0[::-1])
    assert set(np.unique(k0["octave"])) <= set(range(8)) and np.all(np.diff(k0["octave"]) >= 0)


def test_undistort_and_frustum_known_answers(oracle, pkg):
    """Hand-checkable cases of the 8(f).2 / 8(f).3 restatements."""
    ref = oracle._oracle_matcher_class()()
    K = [500.0, 500.0, 320.0, 240.0]
    k = np.zeros(3, pkg.KP_DTYPE); k["x"] = [320.0, 0.0, 639.0]; k["y"] = [240.0, 0.0, 100.0]; k["octave"] = [0, 3, 7]
    # zero first coefficient: copy (Frame.cc:928-932)
    assert ref.UndistortKeyPoints(k, K, [0.0, 0.5, 0.1, 0.1]).tobytes() == k.tobytes()
    # the principal point is a fixed point of any radial/tangential model
    u = ref.UndistortKeyPoints(k, K, [-0.3, 0.1, 0.001, 0.002])
    assert u["x"][0] == 320.0 and u["y"][0] == 240.0 and list(u["octave"]) == [0, 3, 7]
    # barrel distortion (k1 < 0): undistorted corners move outwards
    assert u["x"][1] < 0.0 and u["y"][1] < 0.0
    b = ref.ComputeImageBounds(640, 480, K, [-0.3, 0.1, 0.0, 0.0])
    assert b[0] < 0 and b[1] > 640 and b[2] < 0 and b[3] > 480
    assert list(ref.ComputeImageBounds(640, 480, K, [0.0])) == [0.0, 640.0, 0.0, 480.0]
    # frustum: a point on the optical axis at distance 4, seen head-on, maxDist 4*1.2^3 -> level ceil(3) = 3
    lsf = float(np.log(np.float32(1.2)))
    Pw = np.array([[0, 0, 4.0], [0, 0, -4.0], [100.0, 0, 4.0], [0, 0, 4.0], [0, 0, 4.0]], np.float32)
    nm = np.array([[0, 0, 1.0], [0, 0, 1.0], [0, 0, 1.0], [0, 0, -1.0], [0, 0, 1.0]], np.float32)
    mx = np.array([4.0 * 1.2 ** 3 * 0.999, 10, 10, 10, 3.0], np.float32); mn = np.full(5, 0.5, np.float32)
    cnt, o = ref.isInFrustum(Pw, nm, mn, mx, np.eye(3), np.zeros(3), np.zeros(3), K, [0.0, 640.0, 0.0, 480.0], 40.0, 0.5, lsf, 8)
    assert cnt == 1 and list(o["in_view"]) == [1, 0, 0, 0, 0]          # behind, outside, facing away, too far (4 > 1.2*3)
    assert o["proj_x"][0] == 320.0 and o["proj_y"][0] == 240.0 and o["level"][0] == 3 and o["depth"][0] == 4.0
    assert o["proj_xr"][0] == np.float32(320.0) - np.float32(40.0) * np.float32(0.25) and o["view_cos"][0] == 1.0
    assert o["proj_x"][1] == -1.0 and o["proj_x"][2] == -1.0 and o["proj_x"][3] == 320.0   # bounds test passed before the later rejects


def test_triangulation_gated_known_answers(oracle, pkg):
    # hand-built case for ORBmatcher.cc:1632-1821: one shared vocabulary node, three KF1 features, four KF2 features.
    # dist <= TH_LOW (50) and dist <= bestDist gate the callback; later candidates win ties (`dist > bestDist` rejects).
    ref = oracle._oracle_matcher_class()()
    kp = np.zeros(4, pkg.KP_DTYPE)
    d1 = np.zeros((3, 32), np.uint8)
    d2 = np.zeros((4, 32), np.uint8)
    d2[0, :2] = 0xFF                                   # 16 bits from d1[*]
    d2[1, :1] = 0xFF                                   # 8 bits
    d2[2, :1] = 0x0F; d2[2, 1] = 0x0F                  # 8 bits as well: a tie with d2[1], later index wins
    d2[3, :7] = 0xFF                                   # 56 bits > TH_LOW: never reaches the gate
    fv1 = pkg.feature_vector_csr(np.array([5, 5, 9]))  # feature 2 sits in a node KF2 does not have
    fv2 = pkg.feature_vector_csr(np.array([5, 5, 5, 5]))
    calls = []
    def gate(i1, i2):
        calls.append((i1, i2))
        return not (i1 == 1 and i2 == 2)               # feature 1 may not take candidate 2
    n, m = ref.SearchForTriangulationGated(kp[:3], d1, [0, 0, 0], fv1, kp, d2, [0, 0, 0, 0], fv2, gate)
    assert n == 2 and m.tolist() == [2, 1, -1]
    assert calls == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
    # MapPoints on either side take features out before any distance is looked at
    calls.clear()
    n, m = ref.SearchForTriangulationGated(kp[:3], d1, [1, 0, 0], fv1, kp, d2, [0, 1, 0, 0], fv2, gate)
    assert n == 1 and m.tolist() == [-1, 0, -1] and calls == [(1, 0), (1, 2)]


def test_gray_from_color_known_answers(oracle):
    """OpenCV's RGB2Gray<uchar> fixed point: coefficients sum to 1 << bits, so gray levels map to themselves."""
    for bits, (ry, gy, by) in ((14, (4899, 9617, 1868)), (15, (9798, 19235, 3735))):
        assert ry + gy + by == 1 << bits
        img = np.zeros((1, 4, 3), np.uint8)
        img[0, 0] = (255, 255, 255); img[0, 1] = (255, 0, 0); img[0, 2] = (0, 255, 0); img[0, 3] = (0, 0, 255)
        half = 1 << (bits - 1)
        exp_rgb = [255, (255 * ry + half) >> bits, (255 * gy + half) >> bits, (255 * by + half) >> bits]
        assert list(oracle.gray_from_color(img, False, bits)[0]) == exp_rgb
        assert list(oracle.gray_from_color(img, True, bits)[0]) == [exp_rgb[0], exp_rgb[3], exp_rgb[2], exp_rgb[1]]
        rgba = np.concatenate([img, np.full((1, 4, 1), 77, np.uint8)], axis=-1)
        assert list(oracle.gray_from_color(rgba, False, bits)[0]) == exp_rgb       # alpha ignored


def test_remap_linear_known_answers(oracle):
    """cv::remap fixed point: identity, half-pixel average with round-half-up of the 15-bit sum, zero border."""
    img = np.array([[10, 20, 30], [40, 50, 60]], np.uint8)
    ys, xs = np.mgrid[0:2, 0:3].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(img, xs, ys), img)
    half = oracle.remap_linear(img, xs + 0.5, ys)                        # (a + b + 1) >> 1 ; the last column blends with the border 0
    assert list(half[0]) == [15, 25, 15] and list(half[1]) == [45, 55, 30]
    assert not oracle.remap_linear(img, xs - 5.0, ys).any()
    q = oracle.remap_linear(img, xs + 0.25, ys + 0.5)                    # weights 24*16, 8*16 (x32): exact bilinear, then round
    assert q[0, 0] == (10 * 24 * 16 * 32 + 20 * 8 * 16 * 32 + 40 * 24 * 16 * 32 + 50 * 8 * 16 * 32 + (1 << 14)) >> 15


def test_clahe_known_answers(oracle):
    """CLAHE restatement: a tile holding every gray level once has LUT[i] = cvRound((i+1) * 255/256) when unclipped; a constant
    image maps to one value; clipping at 1 count per bin flattens any histogram to the same (identity-like) LUT."""
    tile = np.arange(256, dtype=np.uint8).reshape(16, 16)
    img = np.tile(tile, (2, 2))                                          # 32x32, 2x2 tiles, every tile identical
    out = oracle.clahe(img, 0.0, (2, 2))
    lut = np.array([int(np.rint(np.float32(i + 1) * np.float32(255.0 / 256.0))) for i in range(256)], np.uint8)
    assert np.array_equal(out, lut[img])
    const = np.full((32, 32), 77, np.uint8)
    o2 = oracle.clahe(const, 0.0, (2, 2))
    assert np.all(o2 == 255)                                             # the whole mass sits at or below 77: cumulative = area
    o3 = oracle.clahe(const, 1e-9, (2, 2))                               # clip limit max(int(..), 1) = 1 -> 255 spread evenly, residual 255 by steps of 1
    assert len(np.unique(o3)) == 1 and 60 <= int(o3[0, 0]) <= 100        # ~ (77+1)/256 * 255
