import importlib, os, sys, time
sys.path.insert(0, "/root/repo"); 
import numpy as np
pkg = importlib.import_module("orb-slam3_amd"); synth = importlib.import_module("orb-slam3_amd.synth")
l, r = synth.gen_stereo_pair(752, 480, 100)
exl = pkg.ORBextractor(1200, max_size=(752, 480)); exr = pkg.ORBextractor(1200, max_size=(752, 480))
m = pkg.ORBmatcher(0.6)
SF, SG = exl.GetScaleFactors(), exl.GetScaleSigmaSquares()
mbf = 47.90639384423901; mb = mbf / 435.2046959714599
F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
fv = lambda d: pkg.feature_vector_csr(d[:, 0].astype(np.int64) & 63)
def T(f, n=30):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
_, kl, dl = exl(l, (0, 0)); _, kr, dr = exr(r, (0, 0))
print("extract one frame     %.3f ms" % T(lambda: exl(l, (0, 0))))
print("stereo matches        %.3f ms" % T(lambda: m.ComputeStereoMatches(exl, exr, kl, dl, kr, dr, mb, mbf)))
f1, f2 = fv(dl), fv(dr)
z = np.zeros(len(kl), np.uint8); z2 = np.zeros(len(kr), np.uint8)
print("triangulation search  %.3f ms" % T(lambda: m.SearchForTriangulation(k1=kl, d1=dl, has_mp1=z, ur1=None, fv1=f1, k2=kr, d2=dr, has_mp2=z2, ur2=None, fv2=f2, F12=F12, ep=(900.0, 240.0), sf2=SF, sigma2_2=SG)))
print("feature_vector_csr    %.3f ms (host python)" % T(lambda: fv(dl)))
fvw = pkg.FrameView(kl, dl, 752, 480)
print("grid build            %.3f ms" % T(lambda: m.grid_build(fvw)))
u = kl["x"].copy(); v = kl["y"].copy(); nq = len(kl)
print("proj-frame search     %.3f ms" % T(lambda: m.SearchByProjectionFrame(fvw, np.zeros(fvw.n, np.uint8), SF, np.ones(nq, np.uint8), u, v, np.zeros(nq, np.float32), kl["octave"], kl["angle"], dl, np.zeros(nq, np.uint8), 15.0)))
