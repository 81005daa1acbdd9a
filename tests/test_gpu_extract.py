"""GPU parity: the HIP extractor through the C ABI vs the CPU oracle, bit-exact (integer/byte/index work;
the three float fields -- angle, scaled x/y -- are compared bit-for-bit too)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # (w, h, nfeatures, seed, lapping, kind)      BASELINE.json configs C1/C2, C3, C4 + degenerate frames
    (752, 480, 1000, 1, (0, 1000), "textured"),
    (752, 480, 1200, 100, (0, 0), "textured"),
    (512, 512, 1500, 200, (0, 511), "textured"),
    (752, 480, 1000, 9, (0, 0), "lowcontrast"),
    (752, 480, 1000, 0, (0, 0), "constant"),
    (640, 480, 1000, 5, (100, 400), "textured"),
    (421, 307, 500, 6, (0, 0), "textured"),          # odd sizes, ragged last cells
    (752, 480, 5000, 9, (0, 1000), "textured"),      # the initialisation extractor (5 * nFeatures, Tracking.cc:1113): > 512 nodes per level -> four fused quadtree iterations
    (2400, 420, 1500, 11, (0, 0), "textured"),       # wide frame: six quadtree roots per level
    (752, 480, 1000, 41, (0, 1000), "sparse"),       # camera-like corner density (~2 % of the level-0 pixels; "textured" is ~6 %)
    (1920, 1080, 4000, 42, (0, 0), "sparse"),
]


def _check(pkg, oracle, synth, w, h, nf, seed, lap, kind, stages=True):
    img = synth.gen_image(w, h, seed, kind)
    ref = oracle.Extractor(nf)
    n_ref, kps_ref, desc_ref, mono_ref = ref(img, lap)
    ex = pkg.ORBextractor(nf, max_size=(w, h), max_batch=1)
    mono, kps, desc = ex(img, lap)
    if stages:
        for l in range(8):
            assert np.array_equal(ex.level_image(l), ref.level_image(l)), "pyramid level %d" % l
            assert np.array_equal(ex.level_candidates(l), ref.level_candidates(l)), "FAST candidates level %d" % l
            assert np.array_equal(ex.level_selected(l), ref.level_keypoints(l)[0]), "quadtree level %d" % l
            bl = ref.level_image(l, blurred=True)
            if bl is not None:
                assert np.array_equal(ex.level_image(l, blurred=True), bl), "blur level %d" % l
    assert len(kps) == n_ref and mono == mono_ref
    assert kps.tobytes() == kps_ref.tobytes()                   # all 7 fields, bit for bit, same order
    assert np.array_equal(desc, desc_ref)
    ex.close()
    return n_ref


@pytest.mark.parametrize("w,h,nf,seed,lap,kind", CASES)
def test_extract_bit_exact(pkg, oracle, synth, w, h, nf, seed, lap, kind):
    n = _check(pkg, oracle, synth, w, h, nf, seed, lap, kind)
    if kind == "textured":
        assert n >= nf // 2


def test_blur_is_inside_the_pass_by_default(pkg):
    ex = pkg.ORBextractor(100, max_size=(752, 480), max_batch=1)
    assert ex.blur_in_pass()
    ex.close()


def test_extract_full_hd_4000(pkg, oracle, synth):
    # BASELINE config C5 at full size
    assert _check(pkg, oracle, synth, 1920, 1080, 4000, 300, (0, 0), "textured", stages=False) >= 3900


def test_mono_init_extractor_5x(pkg, oracle, synth):
    # Tracking.cc:1113: the monocular-initialisation extractor asks for 5*nFeatures
    _check(pkg, oracle, synth, 752, 480, 5000, 12, (0, 1000), "textured", stages=False)


def test_batch_equals_single(pkg, oracle, synth):
    imgs = [synth.gen_image(752, 480, 40 + i) for i in range(6)] + [synth.gen_image(752, 480, 0, "constant")]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=len(imgs))
    laps = [(0, 1000)] * 3 + [(0, 0)] * 4
    res = ex.extract_batch(imgs, laps)
    ref = oracle.Extractor(1000)
    for img, lap, (mono, kps, desc) in zip(imgs, laps, res):
        n_ref, kps_ref, desc_ref, mono_ref = ref(img, lap)
        assert len(kps) == n_ref and mono == mono_ref
        assert kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)


@pytest.mark.parametrize("nframes", [48, 71])
def test_large_batch_paths(pkg, oracle, synth, nframes):
    # from ~40 frames of this size on, k_blur3's workgroups walk eight tiles each instead of one (orbx_api.hip: `walk`);
    # every frame of such a batch -- odd one out included -- must still equal the oracle
    imgs = [synth.gen_image(752, 480, 900 + (i % 5)) for i in range(nframes - 1)] + [synth.gen_image(752, 480, 3, "lowcontrast")]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=len(imgs))
    res = ex.extract_batch(imgs, [(0, 0)] * len(imgs))
    ref = oracle.Extractor(1000)
    want = {}
    for i, img in enumerate(imgs):
        key = 900 + (i % 5) if i < nframes - 1 else -1
        if key not in want:
            want[key] = ref(img, (0, 0))
        n_ref, kps_ref, desc_ref, mono_ref = want[key]
        mono, kps, desc = res[i]
        assert len(kps) == n_ref and mono == mono_ref, i
        assert kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref), i


def test_mixed_texture_batch(pkg, oracle, synth):
    """One batch of 66 DISTINCT frames of three texture classes side by side (dense / sparse / low contrast): per-cell ini->min threshold
    retries and queue overflows (k_fast_fix) of some frames next to dense frames that need neither, every frame equal to the oracle."""
    kinds = ("textured", "sparse", "lowcontrast")
    imgs = [synth.gen_image(752, 480, 4000 + i, kinds[i % 3]) for i in range(66)]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=len(imgs))
    res = ex.extract_batch(imgs, [(0, 1000)] * len(imgs))
    ref = oracle.Extractor(1000)
    ns = {k: [] for k in kinds}
    for i, img in enumerate(imgs):
        n_ref, kps_ref, desc_ref, mono_ref = ref(img, (0, 1000))
        mono, kps, desc = res[i]
        assert len(kps) == n_ref and mono == mono_ref, i
        assert kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref), i
        ns[kinds[i % 3]].append(n_ref)
    assert min(ns["textured"]) >= 1000 and min(ns["sparse"]) >= 900
    ex.close()


def test_idempotent_and_geometry_switch(pkg, synth):
    # the extractor is stateful (pyramid overwritten per call): same input twice -> same output; a new image
    # size rebuilds the geometry
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=1)
    a = synth.gen_image(752, 480, 3); b = synth.gen_image(640, 400, 3)
    r1 = ex(a); r2 = ex(b); r3 = ex(a)
    assert r1[0] == r3[0] and r1[1].tobytes() == r3[1].tobytes() and np.array_equal(r1[2], r3[2])
    assert len(r2[1]) > 0


def test_error_paths(pkg, synth):
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=1)
    assert ex(np.zeros((0, 0), np.uint8))[0] == -1                # empty image: reference returns -1
    with pytest.raises(pkg.OrbError):
        ex(synth.gen_image(1024, 768, 1))                         # larger than configured maximum
    with pytest.raises(pkg.OrbError):
        ex(synth.gen_image(120, 100, 1))                          # top level narrower than one FAST cell
    with pytest.raises(pkg.OrbError):
        pkg.ORBextractor(1000, scale_factor=1.0)


def test_getters_match_oracle_tables(pkg, oracle):
    ex = pkg.ORBextractor(1000, max_size=(752, 480))
    t = oracle.Extractor(1000).tables()
    assert np.array_equal(ex.GetScaleFactors(), t["sf"]) and np.array_equal(ex.GetInverseScaleFactors(), t["inv_sf"])
    assert np.array_equal(ex.GetScaleSigmaSquares(), t["sig2"]) and np.array_equal(ex.GetInverseScaleSigmaSquares(), t["inv_sig2"])
    assert np.array_equal(ex.features_per_level(), t["nfeat"]) and ex.GetLevels() == 8


def test_cpp_facade_runs(pkg, tmp_path):
    """ORB_SLAM3::ORBextractor / ORBmatcher facade classes end to end on the GPU (argv[1] makes 'no GPU' an error)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "facade_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "facade_smoke.cpp"),
                           "-L", os.path.join(root, "orb-slam3_amd"), "-lorbslam3_amd",
                           "-Wl,-rpath," + os.path.join(root, "orb-slam3_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, "need-gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "keypoints" in out.stdout


def test_device_resident_inputs_aligned_and_misaligned(pkg, oracle, synth):
    """orbx_extract_batch_async with device pointers: 16-byte aligned inputs are read in place, anything else
    (odd stride / odd base) is staged device-to-device; both must equal the host-image path bit for bit."""
    import ctypes as C
    for (w, h, stride, shift) in [(752, 480, 768, 0), (421, 307, 421, 0), (421, 307, 432, 4), (752, 480, 752, 16)]:
        imgs = [synth.gen_image(w, h, 60 + i) for i in range(3)]
        ex = pkg.ORBextractor(800, max_size=(w, h), max_batch=3)
        want = ex.extract_batch(imgs)
        dev = pkg.DeviceBuffer(3 * stride * h + 64)
        ptrs = []
        for i, im in enumerate(imgs):
            pad = np.zeros((h, stride), np.uint8); pad[:, :w] = im
            dev.upload(pad, offset=shift + i * stride * h)
            ptrs.append(dev.ptr + shift + i * stride * h)
        arr = (C.c_void_p * 3)(*ptrs)
        ex.enqueue_device(arr, w, h, stride)
        ex.sync()
        for i in range(3):
            mono, kps, desc = ex.fetch(i)
            assert mono == want[i][0] and kps.tobytes() == want[i][1].tobytes() and np.array_equal(desc, want[i][2])
        ex.close()


def test_stage_timing_switch(pkg, synth):
    """orbx_set_stage_timing(0): only the dependency events are recorded; total and the two pass spans stay available."""
    img = synth.gen_image(752, 480, 3)
    ex = pkg.ORBextractor(1000, max_size=(752, 480))
    try:
        ex(img, (0, 1000))
        full, n = ex.mean_timings()
        assert n >= 1 and full["fast"] > 0 and full["blur"] > 0 and full["total"] >= full["pyramid_fast_span"] > 0
        assert ex.L.orbx_set_stage_timing(ex.h, 0) == 0
        ex(img, (0, 1000)); ex(img, (0, 1000))
        lean, n = ex.mean_timings()
        # (slot 1 then holds first launch -> end of the last FAST launch: the pyramid+FAST span without an in-pass blur's tail)
        assert n == 2 and lean["blur"] == 0 and lean["quadtree"] == 0 and lean["total"] >= lean["pyramid_fast_span"] >= lean["fast"] > 0
        assert ex.L.orbx_set_stage_timing(ex.h, 1) == 0
        ex(img, (0, 1000))
        again, n = ex.mean_timings()
        assert n == 1 and again["fast"] > 0
    finally:
        ex.close()


def test_result_fetch_all_layouts(pkg, oracle, synth):
    """orbx_result_fetch_all: one copy per array at the device capacity, strided copies at any other caller capacity, and the
    capacity error when a frame does not fit."""
    import ctypes as C
    imgs = [synth.gen_image(752, 480, 70 + i) for i in range(3)]
    ref = [oracle.Extractor(1000)(im, (0, 1000)) for im in imgs]
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=3)
    try:
        out = ex.extract_batch(imgs, [(0, 1000)] * 3)                    # fetch_all at cap == orbx_max_keypoints
        for (mono, kps, desc), (n0, k0, d0, m0) in zip(out, ref):
            assert mono == m0 and kps.tobytes() == k0.tobytes() and np.array_equal(desc, d0)
        for cap in (ex.cap + 17, max(r[0] for r in ref)):
            kps = np.zeros((3, cap), pkg.KP_DTYPE); desc = np.zeros((3, cap, 32), np.uint8)
            n = np.zeros(3, np.int32); m = np.zeros(3, np.int32)
            rc = ex.L.orbx_result_fetch_all(ex.h, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                            n.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
            assert rc == 3
            for i, (n0, k0, d0, m0) in enumerate(ref):
                assert n[i] == n0 and m[i] == m0 and kps[i, :n0].tobytes() == k0.tobytes() and np.array_equal(desc[i, :n0], d0)
        assert ex.L.orbx_result_fetch_all(ex.h, None, None, 10, None, None) < 0      # 10 keypoints per frame are not enough
    finally:
        ex.close()


def test_failed_geometry_rebuild_leaves_no_stale_state(pkg, oracle, synth):
    """A call that fails while the geometry is being rebuilt (too small / too large an image) must not leave the handle looking
    as if the previous size were still set up: the next valid call rebuilds everything and is bit-exact again."""
    ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=1)
    img = synth.gen_image(752, 480, 21)
    n_ref, kps_ref, desc_ref, mono_ref = oracle.Extractor(1000)(img, (0, 1000))
    for bad in (synth.gen_image(40, 40, 1), synth.gen_image(120, 100, 1)):
        mono, kps, desc = ex(img, (0, 1000))
        assert mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
        with pytest.raises(pkg.OrbError):
            ex(bad)                                               # ORBX_E_TOO_SMALL after the old geometry is torn down
        with pytest.raises(pkg.OrbError):
            ex.fetch(0)                                           # nothing valid to fetch (no dangling result pointers)
    mono, kps, desc = ex(img, (0, 1000))
    assert mono == mono_ref and len(kps) == n_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
    ex.close()


def _device_batch(pkg, synth, w, h, n, seed0):
    import ctypes as C
    stride = (w + 63) // 64 * 64
    imgs = [synth.gen_image(w, h, seed0 + i) for i in range(n)]
    dev = pkg.DeviceBuffer(n * stride * h)
    for i, im in enumerate(imgs):
        pad = np.zeros((h, stride), np.uint8); pad[:, :w] = im
        dev.upload(pad, offset=i * stride * h)
    arr = (C.c_void_p * n)(*[dev.ptr + i * stride * h for i in range(n)])
    return imgs, dev, arr, stride


def test_graph_replay_and_result_blocks_bit_exact(pkg, oracle, synth):
    """bench.py's timed loop in small: the step (extraction into result block k & 1 + the dense match) captured into HIP
    graphs and replayed, the block copied to a pinned host block of the same layout on the copy stream beside the next step.
    Every replay's host copy must equal the eager results and the oracle, bit for bit; the replay timings must be sane."""
    import ctypes as C
    w, h, n = 421, 307, 5
    imgs, dev, arr, stride = _device_batch(pkg, synth, w, h, n, 70)
    L = pkg.lib()
    ex = pkg.ORBextractor(600, max_size=(w, h), max_batch=n)
    mt = pkg.ORBmatcher(0.7)
    assert L.orbm_set_stream(mt.h, L.orbx_stream(ex.h)) == 0
    cap = ex.cap
    ref = oracle.Extractor(600)
    want = [ref(im, (0, 0)) for im in imgs]
    offk, offd, offn, offm, nbytes = ex.result_block_layout()
    host = [pkg.PinnedBuffer(nbytes) for _ in (0, 1)]
    idx2 = pkg.DeviceBuffer(n * cap * 8); dist2 = pkg.DeviceBuffer(n * cap * 8)
    res = []
    for blk in (0, 1):
        assert L.orbx_set_result_block(ex.h, blk) == 0
        res.append(ex.result_device())

    def enqueue(blk):
        assert L.orbx_set_result_block(ex.h, blk) == 0
        ex.enqueue_device(arr, w, h, stride)
        r = res[blk]
        assert L.orbm_knn2_batch_async(mt.h, r["desc"] + cap * 32, cap, r["counts"] + 4, r["desc"], cap, r["counts"], n - 1, cap,
                                       idx2.ptr + cap * 8, dist2.ptr + cap * 8) == 0

    assert L.orbx_capture_begin(ex.h, 0) == 0                    # nothing has run eagerly yet: the per-frame tables are not in place,
    with pytest.raises(pkg.OrbError):                            # so the enqueue is refused and the capture comes back empty-handed
        ex.enqueue_device(arr, w, h, stride)
    assert L.orbx_capture_end(ex.h) < 0 and L.orbx_graph_launch(ex.h, 0) < 0
    enqueue(0); enqueue(1)
    ex.sync()
    for slot in range(4):
        assert L.orbx_capture_begin(ex.h, slot) == 0, L.orbx_last_error()
        assert L.orbx_result_download_async(ex.h, host[0].ptr) < 0      # the copy is never part of a capture
        enqueue(slot & 1)
        assert L.orbx_capture_end(ex.h) == 0, L.orbx_last_error()

    def check_block(blk):
        hb = host[blk].bytes
        cnt = hb[offn:offn + 4 * n].view(np.int32); mono = hb[offm:offm + 4 * n].view(np.int32)
        kp = hb[offk:offk + 28 * cap * n].reshape(n, cap, 28); de = hb[offd:offd + 32 * cap * n].reshape(n, cap, 32)
        for i in range(n):
            n_ref, kps_ref, desc_ref, mono_ref = want[i]
            assert cnt[i] == n_ref and mono[i] == mono_ref, (blk, i)
            assert kp[i, :n_ref].tobytes() == kps_ref.tobytes() and np.array_equal(de[i, :n_ref], desc_ref), (blk, i)

    for k in range(10):                                          # replays, host never waits inside the loop
        assert L.orbx_graph_launch(ex.h, k % 4) == 0, L.orbx_last_error()
        assert L.orbx_result_download_async(ex.h, host[k & 1].ptr) == 0
    ex.sync()
    check_block(0); check_block(1)
    for b in host:
        b.bytes[:] = 0
    assert L.orbx_graph_launch(ex.h, 1) == 0                     # a single replay into block 1, fetched through the ordinary path too
    assert L.orbx_result_download_async(ex.h, host[1].ptr) == 0
    ex.sync()
    check_block(1)
    for i in range(n):
        mono, kps, desc = ex.fetch(i)
        assert kps.tobytes() == want[i][1].tobytes() and np.array_equal(desc, want[i][2])
    # match results of the replay == the eager host-array call
    got_i = idx2.download(np.int32, n * cap * 2).reshape(n, cap, 2); got_d = dist2.download(np.int32, n * cap * 2).reshape(n, cap, 2)
    for i in range(1, n):
        wi, wd = oracle.knn2(want[i][2], want[i - 1][2])
        nq = want[i][0]
        assert np.array_equal(got_i[i, :nq], wi) and np.array_equal(got_d[i, :nq], wd)
    tm, ns = ex.mean_timings()
    assert ns >= 1 and 0 < tm["pyramid_fast_span"] <= tm["total"] < 50
    # a new image size drops the graphs (they hold the old geometry)
    ex(synth.gen_image(400, 300, 1))
    assert L.orbx_graph_launch(ex.h, 0) < 0
    ex.close(); mt.close()


def test_single_frame_graph_replay(pkg, oracle, synth):
    """orbx_extract replays the whole single-frame sequence as one graph from the third call with the same geometry on (per-stage
    timing off): images, lapping areas and sizes change between calls; every result must equal the oracle's."""
    ex = pkg.ORBextractor(800, max_size=(752, 480), max_batch=1)
    assert ex.L.orbx_set_stage_timing(ex.h, 0) == 0
    ref = oracle.Extractor(800)
    plan = [(752, 480, 21, (0, 0)), (752, 480, 22, (0, 1000)), (752, 480, 23, (100, 400)), (752, 480, 21, (0, 0)), (752, 480, 24, (300, 310)),
            (640, 360, 25, (0, 0)), (640, 360, 26, (0, 700)), (640, 360, 27, (5, 50)), (640, 360, 25, (0, 0)),       # geometry switch: new graph
            (752, 480, 22, (0, 1000)), (752, 480, 28, (0, 0)), (752, 480, 29, (0, 0))]
    for w, h, seed, lap in plan:
        img = synth.gen_image(w, h, seed)
        n_ref, kps_ref, desc_ref, mono_ref = ref(img, lap)
        mono, kps, desc = ex(img, lap)
        assert len(kps) == n_ref and mono == mono_ref, (w, h, seed, lap)
        assert kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref), (w, h, seed, lap)
    # a batch call with device-resident images rewrites the handle's image-pointer table: the next single-frame call must notice
    dev = pkg.DeviceBuffer(752 * 480)
    other = synth.gen_image(752, 480, 31)
    dev.upload(np.ascontiguousarray(other))
    import ctypes as C
    ptrs = (C.c_void_p * 1)(int(dev.ptr))
    ex.enqueue_device(ptrs, 752, 480, 752, [0, 0]); ex.sync()
    img = synth.gen_image(752, 480, 28)
    n_ref, kps_ref, desc_ref, mono_ref = ref(img, (0, 0))
    for _ in range(4):
        mono, kps, desc = ex(img, (0, 0))
        assert len(kps) == n_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
    # a non-contiguous caller stride and the batch API in between must not confuse the replay path
    big = np.zeros((480, 800), np.uint8); big[:, :752] = synth.gen_image(752, 480, 30)
    view = big[:, :752]
    kps = np.zeros(ex.cap, pkg.KP_DTYPE); desc = np.zeros((ex.cap, 32), np.uint8)
    import ctypes as C
    mono = C.c_int32(0)
    n = ex.L.orbx_extract(ex.h, big.ctypes.data_as(C.c_void_p), 752, 480, 800, 0, 0, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), ex.cap, C.byref(mono))
    n_ref, kps_ref, desc_ref, mono_ref = ref(np.ascontiguousarray(view), (0, 0))
    assert n == n_ref and kps[:n].tobytes() == kps_ref.tobytes() and np.array_equal(desc[:n], desc_ref)
    ex.close()


def test_pyramid_fetch_equals_per_level_fetch(pkg, synth):
    """orbx_pyramid_fetch (all levels with one copy; level 0 from the staging buffer) against orbx_level_image level by level,
    for a host image and for a frame of a device-resident batch."""
    img = synth.gen_image(641, 479, 41)
    ex = pkg.ORBextractor(500, max_size=(641, 479), max_batch=2)
    ex(img, (0, 0))
    for lv, got in enumerate(ex.pyramid(0)):
        assert np.array_equal(got, ex.level_image(lv)), lv
    assert np.array_equal(ex.pyramid(0)[0], img)
    imgs = [synth.gen_image(641, 479, 42), synth.gen_image(641, 479, 43)]
    ex.extract_batch(imgs, [(0, 0), (0, 0)])
    for f in range(2):
        pyr = ex.pyramid(f)
        assert np.array_equal(pyr[0], imgs[f])
        for lv in range(1, 8):
            assert np.array_equal(pyr[lv], ex.level_image(lv, frame=f)), (f, lv)
    ex.close()


def test_pyramid_map_views(pkg, synth):
    img = synth.gen_image(641, 479, 44)
    ex = pkg.ORBextractor(500, max_size=(641, 479), max_batch=1)
    ex(img, (0, 0))
    views = ex.pyramid_views(0)
    assert np.array_equal(views[0], img)
    for lv in range(1, 8):
        assert np.array_equal(np.array(views[lv]), ex.level_image(lv)), lv
    ex.close()


def test_pyramid_level0_after_a_misaligned_device_batch(pkg, synth):
    """A host-image call leaves its frame in the pinned staging buffer; a later DEVICE batch with a misaligned stride is staged
    device-to-device into the same level-0 area.  The host views / fetches of level 0 must then show the DEVICE batch's image,
    not the older host image still sitting in the staging buffer (round-2 advisor finding)."""
    import ctypes as C
    w, h = 641, 479
    a = synth.gen_image(w, h, 51); b = synth.gen_image(w, h, 52)
    assert not np.array_equal(a, b)
    ex = pkg.ORBextractor(500, max_size=(w, h), max_batch=1)
    ex(a, (0, 0))                                                          # host path: staged through hPinned
    assert np.array_equal(ex.pyramid_views(0)[0], a)
    stride = w + 3                                                         # neither the stride nor the pointer is 16-byte aligned
    dev = pkg.DeviceBuffer(stride * h + 64)
    padded = np.zeros((h, stride), np.uint8); padded[:, :w] = b
    dev.upload(padded, offset=5)
    ptrs = (C.c_void_p * 1)(int(dev.ptr + 5))
    ex.enqueue_device(ptrs, w, h, stride, [(0, 0)])
    ex.sync()
    views = ex.pyramid_views(0)
    # (level 0 of a device batch has no host copy: the map reports none -- what it must never report is the older host frame)
    assert views[0] is None or np.array_equal(np.array(views[0]), b), "level 0 of the device batch came back as the older host image"
    assert np.array_equal(ex.level_image(0), b)
    fetched = ex.pyramid(0)
    assert np.array_equal(fetched[0], b), "orbx_pyramid_fetch returned the stale staging copy for level 0"
    for lv in range(1, 8):
        assert np.array_equal(np.array(views[lv]), ex.level_image(lv)), lv
    ex(a, (0, 0))                                                          # and back: the host path is zero-copy again
    assert np.array_equal(ex.pyramid_views(0)[0], a)
    ex.close()


def test_gfx950_instruction_semantics():
    """The three gfx950 instructions the kernels reach through inline asm behave as the kernels assume: v_ashr_pk_u8_i32 (k_resize2's
    shift-and-pack), v_pk_minimum3_f16 / v_pk_maximum3_f16 on 0..1023 taken as binary16 bit patterns (FAST score network: integer
    minimum / maximum, which needs f16 denormals preserved).  tools/ubench/isa_probe.hip, built by __graft_entry__.build()."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ubench", "isa_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-result", "-o", exe, exe + ".hip"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "DIFFERS" not in p.stdout and p.stdout.count("\n") >= 2, p.stdout
