#!/usr/bin/env python3
"""PCIe-inclusive extraction rate: host images in (pageable numpy), keypoints + descriptors back on the host.
Not the headline metric (bench.py keeps inputs resident in HBM); DESIGN.md section 7 quotes this number."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs = [synth.gen_image(752, 480, 1 + i % 16) for i in range(B)]
ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=B)
for _ in range(2):
    ex.extract_batch(imgs, [(0, 1000)] * B)
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    out = ex.extract_batch(imgs, [(0, 1000)] * B)               # H2D of every image, kernels, D2H of every frame's results
dt = (time.perf_counter() - t0) / reps
print("host-in / host-out: %.0f frames/s (%d-frame batch, %.2f ms per batch, %d keypoints in the last frame)" % (B / dt, B, dt * 1e3, len(out[-1][1])))
ex.close()
