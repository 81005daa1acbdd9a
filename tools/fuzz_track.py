#!/usr/bin/env python3
"""Fuzz of the batched SearchByProjection(Frame, Frame) (orbm_search_by_projection_batch_async: k_track_pack + k_track_topk16 +
k_track_claim) against the CPU oracle: random image sizes, feature counts, window radii (th 5..70: single- and multi-pass windows,
more than 16 grid columns), shifts, blocked / observed rates, orientation check on and off.  Every pair's final match row and count
must equal the oracle's entry for entry.   usage: tools/fuzz_track.py [rounds] [seed]"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
import orbref  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
L = pkg.lib()
OM = orbref._oracle_matcher_class()()
t0 = time.time()
bad = 0
pairs = 0
for it in range(rounds):
    W, H = [(752, 480), (512, 512), (640, 360), (960, 540)][int(rng.integers(0, 4))]
    nf = int(rng.choice([300, 1000, 2000, 3000]))
    NB = int(rng.integers(3, 7))
    kind = ["textured", "sparse", "lowcontrast"][int(rng.integers(0, 3))]
    imgs = [synth.gen_image(W, H, int(rng.integers(0, 1 << 30)), kind=kind) for _ in range(NB)]
    if rng.random() < 0.3:
        imgs[int(rng.integers(0, NB))] = np.full((H, W), 128, np.uint8)          # an empty frame somewhere
    if rng.random() < 0.3:
        imgs[NB - 1] = imgs[NB - 2]                                             # a frame against itself: distance-0 ties
    ex = pkg.ORBextractor(nf, max_size=(W, H), max_batch=NB)
    res = ex.extract_batch(imgs, [(0, 0)] * NB)
    m = pkg.ORBmatcher(0.9)
    r = ex.result_device(); cap = r["cap"]
    gs = pkg.DeviceBuffer(NB * 3073 * 4); gi = pkg.DeviceBuffer(NB * cap * 4)
    inv_w = np.float32(64) / np.float32(W); inv_h = np.float32(48) / np.float32(H)
    assert L.orbm_grid_build_batch_async(m.h, r["kps"], r["counts"], NB, cap, 0.0, 0.0, float(inv_w), float(inv_h), gs.ptr, gi.ptr) == 0
    th = float(rng.choice([5.0, 7.0, 15.0, 15.0, 30.0, 45.0, 70.0]))
    dx, dy = float(np.float32(rng.uniform(-6, 6))), float(np.float32(rng.uniform(-6, 6)))
    check_ori = bool(rng.integers(0, 2))
    blocked = (rng.random((NB, cap)) < rng.choice([0.0, 0.1, 0.5, 0.95], size=(NB, 1))).astype(np.uint8)
    obs = (rng.random((NB, cap)) < rng.choice([1.0, 0.5, 0.0], size=(NB, 1))).astype(np.uint8)
    use_blk, use_obs = rng.random() < 0.8, rng.random() < 0.8
    dblk = pkg.DeviceBuffer(NB * cap); dobs = pkg.DeviceBuffer(NB * cap)
    dblk.upload(blocked); dobs.upload(obs)
    NP = NB - 1
    dm = pkg.DeviceBuffer(NP * cap * 4); dn = pkg.DeviceBuffer(NP * 4)
    sf = ex.GetScaleFactors()
    rc = L.orbm_search_by_projection_batch_async(m.h, r["kps"], r["desc"], r["counts"], cap, gs.ptr, gi.ptr, 0.0, 0.0, float(inv_w), float(inv_h),
                                                 1, 0, NP, th, sf.ctypes.data_as(C.c_void_p), 8, dx, dy, dblk.ptr if use_blk else None,
                                                 dobs.ptr if use_obs else None, int(check_ori), dm.ptr, dn.ptr)
    assert rc == 0, L.orbm_last_error()
    m.sync()
    match = dm.download(np.int32, NP * cap).reshape(NP, cap); nm = dn.download(np.int32, NP)
    for p in range(NP):
        (_, kq, dq), (_, kt, dt) = res[p + 1], res[p]
        nq, nt = len(kq), len(kt)
        pairs += 1
        if nt == 0:
            ok = nm[p] == 0 and np.all(match[p] == -1)
        else:
            args = dict(cur_blocked=blocked[p, :nt] if use_blk else np.zeros(nt, np.uint8), scale_factors=sf, valid=np.ones(nq, np.uint8),
                        u=kq["x"] + np.float32(dx), v=kq["y"] + np.float32(dy), invzc=np.zeros(nq, np.float32), octave=kq["octave"],
                        angle=kq["angle"], qdesc=dq, mp_obs=obs[p + 1, :nq] if use_obs else np.ones(nq, np.uint8), th=th, check_ori=check_ori)
            n_ref, m_ref = OM.SearchByProjectionFrame(pkg.FrameView(kt, dt, W, H, backend=OM), **args)
            ok = nm[p] == n_ref and np.array_equal(match[p, :nt], m_ref) and np.all(match[p, nt:] == -1)
        if not ok:
            bad += 1
            print("MISMATCH round %d pair %d: %dx%d nf %d th %g dx %g dy %g ori %d kind %s" % (it, p, W, H, nf, th, dx, dy, check_ori, kind), flush=True)
    del ex, m
    if it % 5 == 4:
        print("round %d ok so far, %d pairs, %.0fs" % (it, pairs, time.time() - t0), flush=True)
print("done: %d rounds, %d pairs, %d mismatching pairs, %.0fs" % (rounds, pairs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
