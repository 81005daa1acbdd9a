#!/usr/bin/env python3
"""PCIe-inclusive extraction rate through the C ABI: host images in (pageable memory), every frame's keypoints + descriptors
back in host arrays (orbx_extract_batch + orbx_result_fetch_all into preallocated buffers).
Not the headline metric (bench.py keeps inputs resident in HBM); DESIGN.md section 7 quotes this number."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs = [np.ascontiguousarray(synth.gen_image(752, 480, 1 + i % 16)) for i in range(B)]
ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=B)
L = ex.L
ptrs = (C.c_void_p * B)(*[i.ctypes.data for i in imgs])
lap = np.tile(np.array([0, 1000], np.int32), B)
kps = np.zeros((B, ex.cap), pkg.KP_DTYPE); desc = np.zeros((B, ex.cap, 32), np.uint8)
n = np.zeros(B, np.int32); m = np.zeros(B, np.int32)
vp = lambda a: a.ctypes.data_as(C.c_void_p)


def once():
    rc = L.orbx_extract_batch(ex.h, ptrs, pkg.HOST, B, 752, 480, 752, vp(lap), vp(n), vp(m))
    assert rc == 0, rc
    rc = L.orbx_result_fetch_all(ex.h, vp(kps), vp(desc), ex.cap, vp(n), vp(m))
    assert rc == B, rc


for _ in range(2):
    once()
t0 = time.perf_counter(); reps = 5
for _ in range(reps):
    once()
dt = (time.perf_counter() - t0) / reps
print("host-in / host-out: %.0f frames/s (%d-frame batch, %.2f ms per batch, %d keypoints in the last frame)" % (B / dt, B, dt * 1e3, int(n[-1])))
ex.close()
