#!/usr/bin/env python3
"""Writes tests/golden/*.npz: inputs + expected outputs for small cases.

The reference ships no golden vectors and cannot be compiled here, so these vectors are produced by
the CPU ORACLE (oracle/liborbref.so), not by the reference: they pin the oracle against silent drift and give the
GPU path a fixed target that needs no oracle build.  PARITY UNPINNED still applies.  Regenerate with:
    python tools/gen_golden.py
"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orbref
synth = importlib.import_module("orb-slam3_amd.synth")
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# (a) extractor: 320x240, 4 levels, 300 features, both lapping conventions
img = synth.gen_image(320, 240, 77)
ex = orbref.Extractor(300, 1.2, 4, 20, 7)
n, kps, desc, mono = ex(img, (0, 1000))
cands = [ex.level_candidates(l) for l in range(4)]
np.savez_compressed(os.path.join(OUT, "extract_320x240_nf300_l4.npz"), image=img, kps=kps, desc=desc, mono=np.int32(mono),
                    lap=np.array([0, 1000], np.int32), params=np.array([300, 4, 20, 7], np.int32), scale=np.float32(1.2),
                    cand0=cands[0], cand1=cands[1], cand2=cands[2], cand3=cands[3],
                    level3=ex.level_image(3), blur0=ex.level_image(0, blurred=True))
n2, kps2, desc2, mono2 = ex(img, (100, 200))
np.savez_compressed(os.path.join(OUT, "extract_320x240_lap100_200.npz"), kps=kps2, desc=desc2, mono=np.int32(mono2),
                    lap=np.array([100, 200], np.int32))
# (b) matcher: 2-NN on real descriptors + the Hamming matrix corner
idx, dist = orbref.knn2(desc[:120], desc2[::-1][:150])
np.savez_compressed(os.path.join(OUT, "knn2_120x150.npz"), q=desc[:120], t=desc2[::-1][:150], idx=idx, dist=dist)
print("wrote", sorted(os.listdir(OUT)), "n =", n, n2)
