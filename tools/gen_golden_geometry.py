#!/usr/bin/env python3
"""Writes tests/golden/geometry.npz: inputs + expected outputs of the SURVEY 8(f).2-4 functions (undistort, image bounds,
frustum/PredictScale, cvtColor, remap, CLAHE) from the CPU ORACLE -- same status as tools/gen_golden.py: the reference has no vectors
and cannot be built here, so these pin the oracle against drift (PARITY UNPINNED).  Regenerate with:
    python tools/gen_golden_geometry.py
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orbref
pkg = importlib.import_module("orb-slam3_amd")
M = orbref._oracle_matcher_class()()
rng = np.random.default_rng(424242)
K = np.array([458.654, 457.296, 367.215, 248.375], np.float32)               # Examples/Monocular/EuRoC.yaml:9-12
D = np.array([-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], np.float32)
n = 200
kps = np.zeros(n, pkg.KP_DTYPE)
kps["x"] = rng.uniform(0, 752, n).astype(np.float32); kps["y"] = rng.uniform(0, 480, n).astype(np.float32)
kps["size"] = 31; kps["angle"] = rng.uniform(0, 360, n).astype(np.float32); kps["response"] = 50; kps["octave"] = rng.integers(0, 8, n); kps["class_id"] = -1
und = M.UndistortKeyPoints(kps, K, D)
bounds = M.ComputeImageBounds(752, 480, K, D)
m = 300
Pw = rng.uniform(-6, 6, (m, 3)).astype(np.float32); Pw[:, 2] += 5
R = np.eye(3, dtype=np.float32); t = np.array([0.1, -0.05, 0.2], np.float32); Ow = (-R.T @ t).astype(np.float32)
d = Pw - Ow
nm = (d / np.linalg.norm(d, axis=1, keepdims=True) + rng.normal(0, 0.5, (m, 3))).astype(np.float32)
nm /= np.linalg.norm(nm, axis=1, keepdims=True)
dist = np.linalg.norm(d, axis=1).astype(np.float32)
mx = (dist * rng.uniform(0.6, 4.0, m)).astype(np.float32); mn = (mx / 3.5).astype(np.float32)
lsf = np.float32(np.log(np.float32(1.2)))
cnt, fr = M.isInFrustum(Pw, nm, mn, mx, R, t, Ow, K, bounds, 47.90639, 0.5, float(lsf), 8)
rgb = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
synth = importlib.import_module("orb-slam3_amd.synth")
g96 = synth.gen_image(96, 96, 11)[:50, :70]
ys, xs = np.mgrid[0:50, 0:70].astype(np.float32)
mapx = (xs * np.float32(1.03) - np.float32(2.3) + np.float32(0.02) * ys).astype(np.float32)
mapy = (ys * np.float32(0.97) + np.float32(1.7)).astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "geometry.npz"), K=K, D=D, kps=kps, und=und, bounds=bounds,
                    Pw=Pw, normal=nm.astype(np.float32), min_dist=mn, max_dist=mx, R=R, t=t, Ow=Ow, lsf=lsf, cnt=np.int32(cnt),
                    **{"fr_" + k: v for k, v in fr.items()}, rgb=rgb,
                    gray14_rgb=orbref.gray_from_color(rgb, False, 14), gray15_bgr=orbref.gray_from_color(rgb, True, 15),
                    g96=g96, mapx=mapx, mapy=mapy, remap=orbref.remap_linear(g96, mapx, mapy), clahe=orbref.clahe(g96, 3.0, (4, 3)))
print("wrote geometry.npz: in view", cnt, "of", m)
