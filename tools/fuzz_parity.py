#!/usr/bin/env python3
"""Randomised parity sweep: HIP extractor vs CPU oracle over many seeds / sizes / parameters (run on the GPU box).
    tools/fuzz_parity.py [cases] [batch]     batch > 0: every case is a BATCH of 1..batch different frames through extract_batch (the
batched kernel paths: several frames per wave in the resize, per-frame tables), each frame compared with the oracle.
Exits non-zero if any case differs."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import orbref
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
maxbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(2026)
t0 = time.time()
bad = 0
nkp = 0
for case in range(ncases):
    nlevels = int(rng.integers(3, 9))
    sf = float(rng.choice([1.2, 1.2, 1.2, 1.15, 1.25, 1.3]))
    top = sf ** (nlevels - 1)
    wmin = int(np.ceil(70 * top)) + 8
    w = int(rng.integers(wmin, max(wmin + 1, 900))); h = int(rng.integers(max(wmin * 2 // 3, int(np.ceil(70 * top)) + 8), max(wmin, 700)))
    if w / h > 3.2 or h / w > 1.9:       # keep nIni in the supported range and >= 1
        continue
    nf = int(rng.integers(100, 3000))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1))
    kind = str(rng.choice(["textured", "textured", "sparse", "lowcontrast"]))
    lap = (int(rng.integers(0, w)), int(rng.integers(0, w + 200)))
    img = synth.gen_image(w, h, int(rng.integers(1, 10**6)), kind)
    nb = int(rng.integers(1, maxbatch + 1)) if maxbatch > 0 else 0
    try:
        ex = pkg.ORBextractor(nf, sf, nlevels, ini, mn, max_size=(w, h), max_batch=max(nb, 1))
    except pkg.OrbError as e:
        print("case %d skipped (%s)" % (case, e)); continue
    ref = orbref.Extractor(nf, sf, nlevels, ini, mn)
    if nb:
        imgs = [img] + [synth.gen_image(w, h, int(rng.integers(1, 10**6)), kind) for _ in range(nb - 1)]
        if ref(img, lap)[0] < 0:
            print("case %d oracle rejected size %dx%d" % (case, w, h)); ex.close(); continue
        res = ex.extract_batch(imgs, [lap] * nb)
        okb = True
        for im, (mono, kps, desc) in zip(imgs, res):
            n_ref, kps_ref, desc_ref, mono_ref = ref(im, lap)
            okb = okb and len(kps) == n_ref and mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
            nkp += max(n_ref, 0)
        if not okb:
            bad += 1
            print("MISMATCH case %d (batch of %d): %dx%d nf=%d sf=%.2f L=%d th=%d/%d %s lap=%s" % (case, nb, w, h, nf, sf, nlevels, ini, mn, kind, lap))
        ex.close()
        if case % 20 == 0:
            print("case %d ok so far, %d keypoints compared, %.0fs" % (case, nkp, time.time() - t0), flush=True)
        continue
    n_ref, kps_ref, desc_ref, mono_ref = ref(img, lap)
    if n_ref < 0:
        print("case %d oracle rejected size %dx%d" % (case, w, h)); continue
    mono, kps, desc = ex(img, lap)
    ok = len(kps) == n_ref and mono == mono_ref and kps.tobytes() == kps_ref.tobytes() and np.array_equal(desc, desc_ref)
    nkp += n_ref
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d nf=%d sf=%.2f L=%d th=%d/%d %s lap=%s n=%d/%d" % (case, w, h, nf, sf, nlevels, ini, mn, kind, lap, len(kps), n_ref))
        if len(kps) == n_ref:
            for f in kps.dtype.names:
                d = np.nonzero(kps[f] != kps_ref[f])[0]
                if len(d): print("   field", f, "differs at", d[:5], kps[f][d[:3]], kps_ref[f][d[:3]])
            bits = np.unpackbits(desc ^ desc_ref, axis=1).sum(1)
            print("   descriptor rows differing:", int((bits > 0).sum()), "max bits", int(bits.max()) if len(bits) else 0)
    ex.close()
    if case % 20 == 0:
        print("case %d ok so far, %d keypoints compared, %.0fs" % (case, nkp, time.time() - t0), flush=True)
print("done: %d cases, %d keypoints, %d mismatching cases, %.0fs" % (ncases, nkp, bad, time.time() - t0))
sys.exit(1 if bad else 0)
