"""Diagnostic: per-phase cycle shares of k_fast2 (needs a -DORBX_STAMPS build in /tmp/stamps.so)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("orb-slam3_amd")
pkg.LIB_PATH = os.environ.get("ORB_LIB", pkg.LIB_PATH)
synth = importlib.import_module("orb-slam3_amd.synth")
B = 64
ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=B)
imgs = [synth.gen_image(752, 480, 1 + i % 8) for i in range(B)]
ex.extract_batch(imgs)
out = np.zeros(16, np.uint64)
ex.L.orbx_debug_stamps(ex.h, out.ctypes.data_as(C.c_void_p), 16)
ex.extract_batch(imgs)
ex.L.orbx_debug_stamps(ex.h, out.ctypes.data_as(C.c_void_p), 16)
tot = out.sum()
print("stamps:", out[:10].tolist(), "shares:", (out[:10] / max(tot, 1)).round(3).tolist(), ex.timings())
