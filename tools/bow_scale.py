"""8(f).1 at ORBvoc scale: time of the batched ComputeBoW bucket kernel (k = 10, L = 6, levelsup = 4) for 512 frames x 1200 real descriptors."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
m = pkg.ORBmatcher(0.7)
voc = pkg.ORBVocabulary(m, synth.gen_vocabulary(10, 6, seed=7))
ex = pkg.ORBextractor(1200, max_size=(752, 480), max_batch=16)
res = ex.extract_batch([synth.gen_image(752, 480, 50 + i) for i in range(16)], [(0, 0)] * 16)
d16 = np.concatenate([r[2][:1200] for r in res])
desc = np.concatenate([d16] * 32)                                   # 512 frames' worth of real descriptors
n = len(desc)
dd = pkg.DeviceBuffer(desc.nbytes); dd.upload(desc); dn = pkg.DeviceBuffer(4 * n)
for levelsup in (4, 0):
    for _ in range(3):
        assert m.L.orbm_bow_nodes_batch_async(m.h, voc.h, dd.ptr, n, levelsup, dn.ptr) == 0
    m.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        m.L.orbm_bow_nodes_batch_async(m.h, voc.h, dd.ptr, n, levelsup, dn.ptr)
    m.sync()
    dt = (time.perf_counter() - t0) / 10
    print("levelsup %d: %d descriptors in %.3f ms = %.2f us per 1200-descriptor frame" % (levelsup, n, dt * 1e3, dt * 1e6 / (n / 1200)))
