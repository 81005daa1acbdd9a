"""Latency of the single-pair stereo / triangulation entry points (752x480, 1200 features), median of 100."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb-slam3_amd"); synth = importlib.import_module("orb-slam3_amd.synth")
l, r = synth.gen_stereo_pair(752, 480, 100)
exl = pkg.ORBextractor(1200, max_size=(752, 480)); exr = pkg.ORBextractor(1200, max_size=(752, 480))
for e in (exl, exr): e.L.orbx_set_stage_timing(e.h, 0)
_, kl, dl = exl(l, (0, 0)); _, kr, dr = exr(r, (0, 0))
M = pkg.ORBmatcher(0.6)
mbf = 47.90639384423901; mb = mbf / 435.2046959714599
F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
fv = lambda d: pkg.feature_vector_csr(d[:, 0].astype(np.int64) & 63)
fl, fr = fv(dl), fv(dr)
sf = exl.GetScaleFactors(); s2 = exl.GetScaleSigmaSquares()
def med(f, n=100):
    for _ in range(5): f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))
print("extract left + right                 : %.3f ms" % med(lambda: (exl(l, (0, 0)), exr(r, (0, 0)))))
print("ComputeStereoMatches (single pair)   : %.3f ms" % med(lambda: M.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)))
z = np.zeros(len(kl), bool); zr = np.zeros(len(kr), bool)
print("SearchForTriangulation (single pair) : %.3f ms" % med(lambda: M.SearchForTriangulation(k1=kl, d1=dl, has_mp1=z, ur1=np.full(len(kl), -1.0), fv1=fl, k2=kr, d2=dr, has_mp2=zr,
      ur2=np.full(len(kr), -1.0), fv2=fr, F12=F12, ep=(900.0, 240.0), sf2=sf, sigma2_2=s2, only_stereo=False, coarse=False, check_ori=False)))
