#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP extractor against the CPU oracle (run on the GPU box)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orbref
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")


def compare(w, h, nfeat, seed, lap, kind="textured"):
    img = synth.gen_image(w, h, seed, kind)
    ref = orbref.Extractor(nfeat)
    n_ref, kps_ref, desc_ref, mono_ref = ref(img, lap)
    ex = pkg.ORBextractor(nfeat, max_size=(w, h), max_batch=2)
    mono, kps, desc = ex(img, lap)
    ok = True
    for l in range(8):
        a = ex.level_image(l); b = ref.level_image(l)
        same = a.shape == b.shape and np.array_equal(a, b)
        ca = ex.level_candidates(l); cb = ref.level_candidates(l)
        csame = ca.shape == cb.shape and np.array_equal(ca, cb)
        sa = ex.level_selected(l); sb, _ = ref.level_keypoints(l)
        ssame = sa.shape == sb.shape and np.array_equal(sa, sb)
        bl = ref.level_image(l, blurred=True)
        bsame = bl is None or np.array_equal(ex.level_image(l, blurred=True), bl)
        print("  L%d img=%s cand=%s (%d/%d) sel=%s (%d/%d) blur=%s" % (l, same, csame, len(ca), len(cb), ssame, len(sa), len(sb), bsame))
        if not ssame and len(sa) and len(sb):
            k = min(len(sa), len(sb))
            bad = np.nonzero((sa[:k] != sb[:k]).any(1))[0]
            print("     first sel mismatch at", bad[:5], "set-equal:", set(map(tuple, sa)) == set(map(tuple, sb)))
        if not csame and len(ca) and len(cb):
            k = min(len(ca), len(cb))
            bad = np.nonzero((ca[:k] != cb[:k]).any(1))[0]
            print("     first cand mismatch", bad[:5], ca[bad[:3]].tolist() if len(bad) else "", cb[bad[:3]].tolist() if len(bad) else "")
        ok &= same and csame and ssame and bsame
    kr = kps_ref.view(kps.dtype)
    same_n = len(kps) == n_ref and mono == mono_ref
    same_k = same_n and np.array_equal(kps, kr)
    same_d = same_n and np.array_equal(desc, desc_ref)
    if same_n and not same_k:
        for f in kps.dtype.names:
            print("     field", f, "equal:", np.array_equal(kps[f], kr[f]), "maxdiff", np.abs(kps[f].astype(np.float64) - kr[f]).max())
    if same_n and not same_d:
        bits = np.unpackbits(desc ^ desc_ref, axis=1).sum(1)
        print("     descriptor rows differing:", int((bits > 0).sum()), "max bits", int(bits.max()))
    print("%dx%d nf=%d seed=%d %s lap=%s: n=%d/%d mono=%d/%d kps=%s desc=%s stages=%s  gpu=%s" % (
        w, h, nfeat, seed, kind, lap, len(kps), n_ref, mono, mono_ref, same_k, same_d, ok, ex.timings()))
    return ok and same_k and same_d


if __name__ == "__main__":
    allok = True
    allok &= compare(752, 480, 1000, 1, (0, 1000))
    allok &= compare(752, 480, 1200, 100, (0, 0))
    allok &= compare(512, 512, 1500, 200, (0, 511))
    allok &= compare(752, 480, 1000, 9, (0, 0), "lowcontrast")
    allok &= compare(752, 480, 1000, 0, (0, 0), "constant")
    allok &= compare(1920, 1080, 4000, 300, (0, 0))
    print("ALL OK" if allok else "MISMATCH")
    sys.exit(0 if allok else 1)
