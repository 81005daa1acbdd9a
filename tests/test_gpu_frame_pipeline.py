"""End-to-end replay of the monocular Frame constructor (src/Frame.cc:324-420) through the drop-in entry points, in the
reference's order: [cvtColor ->] ExtractORB -> UndistortKeyPoints -> ComputeImageBounds -> AssignFeaturesToGrid, then one
Tracking::SearchLocalPoints-style pass (isInFrustum -> SearchByProjection).  GPU path vs the CPU oracle at every step."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = [458.654, 457.296, 367.215, 248.375]                                  # Examples/Monocular/EuRoC.yaml:9-17
D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]


def _frame(backend_extract, matcher, pkg, gray, W, H):
    mono, kps, desc = backend_extract(gray)
    un = matcher.UndistortKeyPoints(kps, K, D)                            # mvKeysUn
    b = matcher.ComputeImageBounds(W, H, K, D)                           # mnMinX, mnMaxX, mnMinY, mnMaxY
    fv = pkg.FrameView(un, desc, W, H)
    fv.min_x, fv.min_y = np.float32(b[0]), np.float32(b[2])
    fv.inv_w = np.float32(pkg.GRID_COLS) / (np.float32(b[1]) - np.float32(b[0]))     # Frame.cc:401-402
    fv.inv_h = np.float32(pkg.GRID_ROWS) / (np.float32(b[3]) - np.float32(b[2]))
    matcher.grid_build(fv)
    return kps, un, desc, b, fv


def test_mono_frame_construction_and_local_point_search(pkg, oracle, synth):
    W, H = 752, 480
    g = synth.gen_image(W, H, 91)
    rgb = np.stack([g, g, g], axis=-1)
    ex = pkg.ORBextractor(1000, max_size=(W, H), max_batch=1)
    ref_ex = oracle.Extractor(1000)
    m = pkg.ORBmatcher(); om = oracle._oracle_matcher_class()()
    try:
        def gpu_extract(_gray):
            buf, stride = ex.gray_from_color([rgb], False, 15)             # Tracking::GrabImageMonocular: cvtColor first
            ex.enqueue_device((C.c_void_p * 1)(buf.ptr), W, H, stride, [(0, 1000)])
            ex.sync()
            return ex.fetch(0)

        def cpu_extract(gray):
            n, kps, desc, mono = ref_ex(oracle.gray_from_color(rgb, False, 15), (0, 1000))
            return mono, kps, desc

        kps, un, desc, b, fv = _frame(gpu_extract, m, pkg, g, W, H)
        kps0, un0, desc0, b0, fv0 = _frame(cpu_extract, om, pkg, g, W, H)
        assert kps.tobytes() == kps0.tobytes() and un.tobytes() == un0.tobytes() and np.array_equal(desc, desc0)
        assert b.tobytes() == b0.tobytes()
        assert np.array_equal(fv.grid_start, fv0.grid_start) and np.array_equal(fv.grid_idx[:fv.placed], fv0.grid_idx[:fv0.placed])
        assert fv.placed == fv0.placed

        # local map points: back-project a subset of the frame's own undistorted keypoints to 3-D, perturb, and search them
        rng = np.random.default_rng(5)
        sel = rng.choice(len(un), 400, replace=False)
        z = rng.uniform(2.0, 8.0, len(sel)).astype(np.float32)
        Pw = np.stack([(un["x"][sel] - K[2]) / K[0] * z, (un["y"][sel] - K[3]) / K[1] * z, z], axis=1).astype(np.float32)
        Pw += rng.normal(0, 0.01, Pw.shape).astype(np.float32)
        nm = (Pw / np.linalg.norm(Pw, axis=1, keepdims=True)).astype(np.float32)
        dist = np.linalg.norm(Pw, axis=1).astype(np.float32)
        lvl = un["octave"][sel].astype(np.float32)
        mx = (dist * np.float32(1.2) ** lvl * np.float32(0.99)).astype(np.float32); mn = (mx / np.float32(1.2) ** 7).astype(np.float32)
        args = (Pw, nm, mn, mx, np.eye(3), np.zeros(3), np.zeros(3), K, b, 0.0, 0.5, float(np.log(np.float32(1.2))), 8)
        c1, fr = m.isInFrustum(*args); c0, fr0 = om.isInFrustum(*args)
        assert c1 == c0 and c1 > 100
        for k in ("in_view", "proj_x", "proj_y"):
            assert fr[k].tobytes() == fr0[k].tobytes()
        v = fr0["in_view"].astype(bool)
        assert np.array_equal(fr["level"][v], fr0["level"][v]) and fr["view_cos"][v].tobytes() == fr0["view_cos"][v].tobytes()
        sf = ref_ex.tables()["sf"]
        qdesc = desc[sel]
        blocked = np.zeros(fv.n, np.uint8); obs = np.zeros(len(sel), np.uint8)
        got = m.SearchByProjectionPoints(fv, blocked, sf, fr["in_view"], fr["proj_x"], fr["proj_y"], fr["proj_xr"], fr["view_cos"], fr["level"], qdesc, obs, 3.0, 0.8)
        exp = om.SearchByProjectionPoints(fv0, blocked, sf, fr0["in_view"], fr0["proj_x"], fr0["proj_y"], fr0["proj_xr"], fr0["view_cos"], fr0["level"], qdesc, obs, 3.0, 0.8)
        assert got[0] == exp[0] and np.array_equal(got[1], exp[1]) and got[0] > 50
    finally:
        ex.close()


def test_guarded_reader_overlaps_the_next_batch(pkg, oracle, synth):
    # orbx_guard_results: a matcher on its own stream reads batch k's descriptors while batch k+1 is already being extracted;
    # batch k+1 may overwrite the result block only after the reader has finished.  Many rounds with alternating inputs: the
    # 2-NN of every round must be the one of that round's frames (oracle), never a mix with the next round's.
    W, H, B = 752, 480, 4
    sets = [[synth.gen_image(W, H, 500 + 10 * s + i) for i in range(B)] for s in range(2)]
    stride = 768
    bufs = []
    for imgs in sets:
        b = pkg.DeviceBuffer(stride * H * B)
        for i, im in enumerate(imgs):
            pad = np.zeros((H, stride), np.uint8); pad[:, :W] = im
            b.upload(pad, i * stride * H)
        bufs.append(b)
    ex = pkg.ORBextractor(1000, max_size=(W, H), max_batch=B)
    m = pkg.ORBmatcher()
    L = ex.L
    L.orbx_stream_wait_results.argtypes = [C.c_void_p, C.c_void_p]
    L.orbm_stream.restype = C.c_void_p; L.orbm_stream.argtypes = [C.c_void_p]
    ms = L.orbm_stream(m.h)
    rounds = 6
    cap = ex.result_device()["cap"]
    outs = [(pkg.DeviceBuffer((B - 1) * cap * 8), pkg.DeviceBuffer((B - 1) * cap * 8)) for _ in range(rounds)]
    for k in range(rounds):
        ptrs = (C.c_void_p * B)(*[bufs[k & 1].ptr + i * stride * H for i in range(B)])
        ex.enqueue_device(ptrs, W, H, stride, [(0, 0)] * B)
        r = ex.result_device()
        assert L.orbx_stream_wait_results(ex.h, ms) == 0
        rc = L.orbm_knn2_batch_async(m.h, r["desc"] + cap * 32, cap, r["counts"] + 4, r["desc"], cap, r["counts"], B - 1, cap,
                                     outs[k][0].ptr, outs[k][1].ptr)
        assert rc == 0
        assert L.orbx_guard_results(ex.h, ms) == 0
    ex.sync(); m.sync()
    want = []
    for imgs in sets:
        ref = oracle.Extractor(1000)
        descs = [ref(im, (0, 0))[2] for im in imgs]
        want.append([oracle.knn2(descs[i + 1], descs[i]) for i in range(B - 1)])
    for k in range(rounds):
        idx = outs[k][0].download(np.int32, (B - 1) * cap * 2).reshape(B - 1, cap, 2)
        dist = outs[k][1].download(np.int32, (B - 1) * cap * 2).reshape(B - 1, cap, 2)
        for i in range(B - 1):
            ridx, rdist = want[k & 1][i]
            n = len(ridx)
            assert np.array_equal(idx[i, :n], ridx) and np.array_equal(dist[i, :n], rdist), (k, i)
