"""Single-frame call: host time of the enqueue alone vs the whole call (752x480, 1200 features), median of 200."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb-slam3_amd"); synth = importlib.import_module("orb-slam3_amd.synth")
img = np.ascontiguousarray(synth.gen_image(752, 480, 100))
ex = pkg.ORBextractor(1200, max_size=(752, 480), max_batch=1)
if os.environ.get("STAGE_TIMING", "0") == "0":
    ex.L.orbx_set_stage_timing(ex.h, 0)
L = ex.L
kps = np.zeros(ex.cap, pkg.KP_DTYPE); desc = np.zeros((ex.cap, 32), np.uint8); mono = C.c_int32(0)
p = lambda a: a.ctypes.data_as(C.c_void_p)
ptr = (C.c_void_p * 1)(img.ctypes.data)
lap = np.array([0, 0], np.int32)
def enq():
    assert L.orbx_extract_batch_async(ex.h, ptr, pkg.HOST, 1, 752, 480, 752, p(lap)) == 0
def full():
    assert L.orbx_extract(ex.h, p(img), 752, 480, 752, 0, 0, p(kps), p(desc), ex.cap, C.byref(mono)) > 0
for _ in range(20): full()
te, tf, ts = [], [], []
for _ in range(200):
    t0 = time.perf_counter(); enq(); t1 = time.perf_counter(); L.orbx_sync(ex.h); t2 = time.perf_counter()
    te.append(t1 - t0); ts.append(t2 - t0)
    t0 = time.perf_counter(); full(); tf.append(time.perf_counter() - t0)
print("enqueue only %.3f ms; enqueue + sync %.3f ms; orbx_extract (with results on the host) %.3f ms" % tuple(1e3 * float(np.median(x)) for x in (te, ts, tf)))
# the facade's default also brings the pyramid back to the host (mvImagePyramid): one call for all levels vs one per level
def pyr_all(): ex.pyramid(0)
def pyr_each():
    for l in range(8): ex.level_image(l)
ta, tb = [], []
for _ in range(100):
    full(); t0 = time.perf_counter(); pyr_all(); ta.append(time.perf_counter() - t0)
    full(); t0 = time.perf_counter(); pyr_each(); tb.append(time.perf_counter() - t0)
print("pyramid to host: orbx_pyramid_fetch %.3f ms; eight orbx_level_image calls %.3f ms" % (1e3 * float(np.median(ta)), 1e3 * float(np.median(tb))))
