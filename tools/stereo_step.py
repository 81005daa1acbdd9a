#!/usr/bin/env python3
"""BASELINE config 3 as a step: per stereo pair (752x480, 1200 features per image) two extractions + Frame::ComputeStereoMatches
(M15) + one SearchForTriangulation (M10) against the previous step's left features (synthetic FeatureVector: first 6 bits of the
descriptor; fixed small-motion F12).  Prints GPU steps/s and the same steps through the CPU oracle.
This is the single-pair (latency) API, not the batched bench path."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orbref
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NCPU = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pairs = [synth.gen_stereo_pair(752, 480, 100 + t) for t in range(NSTEP)]
mbf = 47.90639384423901; mb = mbf / 435.2046959714599                     # Examples/Stereo/EuRoC.yaml:9,28
F12 = np.array([[1e-7, -3e-6, 1.1e-3], [2.5e-6, 2e-7, -0.0231], [-1.3e-3, 0.0229, 0.35]], np.float32)
fv = lambda d: pkg.feature_vector_csr(d[:, 0].astype(np.int64) & 63)


def run(ex_l, ex_r, M, extract, n):
    prev = None; res = []
    t0 = time.perf_counter()
    for t in range(n):
        l, r = pairs[t]
        (kl, dl), (kr, dr) = extract(ex_l, l), extract(ex_r, r)
        ns, ur, dp = M.ComputeStereoMatches(ex_l, ex_r, kl, dl, kr, dr, mb, mbf)
        nt = -1
        if prev is not None:
            pk, pd, pur = prev
            nt, m12 = M.SearchForTriangulation(k1=pk, d1=pd, has_mp1=np.zeros(len(pk), np.uint8), ur1=pur, fv1=fv(pd), k2=kl, d2=dl,
                                               has_mp2=np.zeros(len(kl), np.uint8), ur2=ur, fv2=fv(dl), F12=F12, ep=(900.0, 240.0), sf2=SF, sigma2_2=SG)
        prev = (kl, dl, ur); res.append((len(kl), len(kr), ns, nt))
    return (time.perf_counter() - t0) / n, res


exl = pkg.ORBextractor(1200, max_size=(752, 480)); exr = pkg.ORBextractor(1200, max_size=(752, 480))
SF, SG = exl.GetScaleFactors(), exl.GetScaleSigmaSquares()
gm = pkg.ORBmatcher(0.6)
gpu_extract = lambda ex, im: ex(im, (0, 0))[1:]
run(exl, exr, gm, gpu_extract, 3)
tg, rg = run(exl, exr, gm, gpu_extract, NSTEP)
ol, orr = orbref.Extractor(1200), orbref.Extractor(1200)
om = orbref._oracle_matcher_class()()
cpu_extract = lambda ex, im: (lambda o: (o[1], o[2]))(ex(im, (0, 0)))
tc, rc = run(ol, orr, om, cpu_extract, NCPU)
assert rg[:NCPU] == rc, (rg[:NCPU], rc)
print("stereo step (2 extractions + ComputeStereoMatches + SearchForTriangulation): GPU %.2f ms = %.0f steps/s; CPU oracle %.1f ms = %.2f steps/s (1 thread); identical match counts on the first %d steps: %s" % (tg * 1e3, 1 / tg, tc * 1e3, 1 / tc, NCPU, rc[-1]))
