"""Frame::UndistortKeyPoints, ComputeImageBounds, isInFrustum + PredictScale on the GPU vs the
oracle restatement (OpenCV's undistortPoints and the platform logf are unpinned third-party arithmetic: parity unpinned)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EUROC_K = [458.654, 457.296, 367.215, 248.375]                       # Examples/Monocular/EuRoC.yaml:9-17
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]


def _kps(pkg, n, w, h, seed):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, pkg.KP_DTYPE)
    k["x"] = rng.uniform(0, w, n).astype(np.float32); k["y"] = rng.uniform(0, h, n).astype(np.float32)
    k["size"] = 31; k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["response"] = rng.integers(7, 255, n)
    k["octave"] = rng.integers(0, 8, n); k["class_id"] = -1
    return k


@pytest.mark.parametrize("dist", [EUROC_D, EUROC_D + [0.01], [0.0, 0.1, 0.0, 0.0], [0.3, -0.2, 1e-3, -2e-3, 0.05], [-0.9, 0.0, 0.0, 0.0]])
@pytest.mark.parametrize("n", [0, 1, 255, 1000, 5000])
def test_undistort_keypoints_bit_exact(pkg, oracle, dist, n):
    m = pkg.ORBmatcher(); ref = oracle._oracle_matcher_class()()
    k = _kps(pkg, n, 752, 480, 7 + n)
    got = m.UndistortKeyPoints(k, EUROC_K, dist); exp = ref.UndistortKeyPoints(k, EUROC_K, dist)
    assert got.tobytes() == exp.tobytes()
    if dist[0] == 0.0:
        assert got.tobytes() == k.tobytes()                              # Frame.cc:928-932: plain copy
    newk = [400.0, 410.0, 320.5, 240.25]
    assert m.UndistortKeyPoints(k, EUROC_K, dist, newk).tobytes() == ref.UndistortKeyPoints(k, EUROC_K, dist, newk).tobytes()


@pytest.mark.parametrize("dist", [EUROC_D, [0.0, 0.0, 0.0, 0.0], [0.2, 0.0, 0.0, 0.0, 0.0]])
@pytest.mark.parametrize("size", [(752, 480), (512, 512), (1920, 1080)])
def test_image_bounds(pkg, oracle, dist, size):
    m = pkg.ORBmatcher(); ref = oracle._oracle_matcher_class()()
    got = m.ComputeImageBounds(size[0], size[1], EUROC_K, dist); exp = ref.ComputeImageBounds(size[0], size[1], EUROC_K, dist)
    assert got.tobytes() == exp.tobytes()
    if dist[0] == 0.0:
        assert list(got) == [0.0, float(size[0]), 0.0, float(size[1])]


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    Pw = rng.uniform(-6, 6, (n, 3)).astype(np.float32); Pw[:, 2] += 5.0
    ang = rng.uniform(-0.3, 0.3, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
         np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])).astype(np.float32)
    t = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    Ow = (-R.T @ t).astype(np.float32)
    d = Pw - Ow
    nm = d / np.linalg.norm(d, axis=1, keepdims=True)
    nm = (nm + rng.normal(0, 0.6, (n, 3))).astype(np.float32)
    nm /= np.linalg.norm(nm, axis=1, keepdims=True)
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    mx = (dist * rng.uniform(0.6, 4.0, n)).astype(np.float32); mn = (mx / 1.2 ** 7 * rng.uniform(0.5, 1.5, n)).astype(np.float32)
    return Pw, nm.astype(np.float32), mn, mx, R, t, Ow


@pytest.mark.parametrize("n", [0, 1, 300, 4000])
@pytest.mark.parametrize("cos_limit", [0.5, 0.6])
def test_is_in_frustum_matches_oracle(pkg, oracle, n, cos_limit):
    m = pkg.ORBmatcher(); ref = oracle._oracle_matcher_class()()
    Pw, nm, mn, mx, R, t, Ow = _scene(n, 100 + n)
    lsf = float(np.log(np.float32(1.2)))
    args = (Pw, nm, mn, mx, R, t, Ow, EUROC_K, [-20.0, 770.0, -15.0, 495.0], 47.90639384423901, cos_limit, lsf, 8)
    c1, got = m.isInFrustum(*args); c0, exp = ref.isInFrustum(*args)
    assert c1 == c0
    if n >= 300:
        assert 0 < c0 < n                                                # the scene exercises accepts and every reject branch
    for key in ("in_view", "proj_x", "proj_y"):
        assert got[key].tobytes() == exp[key].tobytes(), key
    v = exp["in_view"].astype(bool)
    for key in ("proj_xr", "depth", "view_cos", "level"):
        assert got[key][v].tobytes() == exp[key][v].tobytes(), key
        assert np.array_equal(got[key][~v], exp[key][~v]), key           # untouched where rejected


def test_frustum_feeds_search_by_projection(pkg, oracle):
    """The frustum outputs are exactly the arrays SearchByProjection(Frame&, vector<MapPoint*>&) consumes."""
    m = pkg.ORBmatcher()
    Pw, nm, mn, mx, R, t, Ow = _scene(500, 9)
    cnt, o = m.isInFrustum(Pw, nm, mn, mx, R, t, Ow, EUROC_K, [0.0, 752.0, 0.0, 480.0], 47.9, 0.5, float(np.log(np.float32(1.2))), 8)
    assert cnt == int(o["in_view"].sum())
    assert np.all(o["level"][o["in_view"] == 1] >= 0) and np.all(o["level"][o["in_view"] == 1] < 8)
    assert np.all(o["proj_x"][o["in_view"] == 1] >= 0.0)


@pytest.mark.parametrize("ch,blue_first", [(3, False), (3, True), (4, False), (4, True)])
@pytest.mark.parametrize("bits", [14, 15])
@pytest.mark.parametrize("size", [(752, 480), (101, 37), (4, 3), (1, 1)])
def test_gray_from_color_bit_exact(pkg, oracle, ch, blue_first, bits, size):
    """SURVEY 8(f).4 ingest: cvtColor *2GRAY (Tracking.cc:1264-1290) on the GPU vs the restated OpenCV fixed point."""
    w, h = size
    rng = np.random.default_rng(w * 31 + h + ch)
    imgs = [rng.integers(0, 256, (h, w, ch), dtype=np.uint8) for _ in range(2)]
    imgs[1][..., :3] = np.array([[255, 255, 255]], np.uint8) if w > 1 else imgs[1][..., :3]      # saturation corner: white stays 255
    ex = pkg.ORBextractor(500, max_size=(752, 480), max_batch=2)
    try:
        buf, stride = ex.gray_from_color(imgs, blue_first, bits)
        for k, img in enumerate(imgs):
            got = buf.download(np.uint8, stride * h, offset=k * stride * h).reshape(h, stride)[:, :w]
            assert np.array_equal(got, oracle.gray_from_color(img, blue_first, bits))
        if w > 1:
            assert np.all(buf.download(np.uint8, stride * h, offset=stride * h).reshape(h, stride)[:, :w] == 255)
    finally:
        ex.close()


def test_gray_then_extract_matches_gray_input(pkg, oracle, synth):
    """The converted images are usable in place by the extractor (16-byte aligned rows)."""
    import ctypes as C
    g = synth.gen_image(752, 480, 5)
    rgb = np.stack([g, g, g], axis=-1)                                   # R = G = B -> gray == the channel (coefficients sum to 1)
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=1)
    try:
        buf, stride = ex.gray_from_color([rgb], False, 15)
        ex.enqueue_device((C.c_void_p * 1)(buf.ptr), 752, 480, stride, [(0, 1000)])
        ex.sync()
        mono, kps, desc = ex.fetch(0)
        n_ref, kps_ref, desc_ref, mono_ref = oracle.Extractor(1000)(g, (0, 1000))
        assert mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
    finally:
        ex.close()


def _rectify_maps(w, h, seed):
    """Smooth synthetic rectification maps (a mild radial + shear warp around the identity), partly leaving the image."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    cx, cy = w * 0.5, h * 0.5
    r2 = ((xs - cx) ** 2 + (ys - cy) ** 2) / (cx * cx + cy * cy)
    k = np.float32(rng.uniform(-0.08, 0.08))
    mx = (cx + (xs - cx) * (1 + k * r2) + np.float32(rng.uniform(-3, 3)) + 0.01 * (ys - cy)).astype(np.float32)
    my = (cy + (ys - cy) * (1 + k * r2) + np.float32(rng.uniform(-3, 3))).astype(np.float32)
    return mx, my


@pytest.mark.parametrize("size", [(752, 480), (101, 37), (5, 3)])
def test_remap_linear_bit_exact(pkg, oracle, synth, size):
    """SURVEY 8(f).4 stereo rectification: cv::remap INTER_LINEAR (stereo_euroc.cc:168-169) vs the restated OpenCV fixed point."""
    w, h = size
    imgs = [synth.gen_image(max(w, 96), max(h, 96), 40 + k)[:h, :w] for k in range(2)]
    mx, my = _rectify_maps(w, h, w + h)
    ex = pkg.ORBextractor(500, max_size=(752, 480), max_batch=2)
    try:
        buf, stride = ex.remap_linear(imgs, mx, my)
        for k, im in enumerate(imgs):
            got = buf.download(np.uint8, stride * h, offset=k * stride * h).reshape(h, stride)[:, :w]
            exp = oracle.remap_linear(im, mx, my)
            assert np.array_equal(got, exp)
        # identity maps reproduce the image; maps far outside read the constant border 0
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
        buf, stride = ex.remap_linear(imgs[:1], xs, ys)
        assert np.array_equal(buf.download(np.uint8, stride * h).reshape(h, stride)[:, :w], imgs[0])
        buf, stride = ex.remap_linear(imgs[:1], xs + 10000.0, ys)
        assert not buf.download(np.uint8, stride * h).reshape(h, stride)[:, :w].any()
    finally:
        ex.close()


@pytest.mark.parametrize("size,tiles,clip", [((512, 512), (8, 8), 3.0), ((752, 480), (8, 8), 3.0), ((101, 37), (8, 8), 3.0),
                                             ((320, 240), (4, 6), 40.0), ((256, 128), (8, 8), 0.0), ((33, 65), (2, 3), 1.5)])
def test_clahe_bit_exact(pkg, oracle, synth, size, tiles, clip):
    """cv::createCLAHE(clip, tiles)->apply (mono_tum_vi.cc:101-109) on the GPU vs the restated OpenCV 4.x CLAHE,
    including the reflect-101 extension of sizes that are not a multiple of the tile grid and the unclipped (clip 0) variant."""
    w, h = size
    imgs = [synth.gen_image(max(w, 96), max(h, 96), 60 + k, "lowcontrast" if k else "textured")[:h, :w] for k in range(2)]
    ex = pkg.ORBextractor(500, max_size=(752, 480), max_batch=2)
    try:
        buf, stride = ex.clahe(imgs, clip, tiles)
        for k, im in enumerate(imgs):
            got = buf.download(np.uint8, stride * h, offset=k * stride * h).reshape(h, stride)[:, :w]
            assert np.array_equal(got, oracle.clahe(im, clip, tiles)), k
    finally:
        ex.close()
