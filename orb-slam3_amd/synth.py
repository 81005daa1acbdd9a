"""Seeded synthetic inputs (no datasets in the image: SURVEY F7, section 8(d)).

gen_image(w, h, seed): 8-octave value noise (mean 110, sigma ~45) + w*h/600 axis-aligned
rectangles (3..40 px, contrast +-30..120) + w*h/1500 small blobs + N(0,2) pixel noise, clipped to u8.
Pure numpy, deterministic for a given (w, h, seed, numpy version's PCG64 stream).
kind = "textured" (above: SURVEY 8(d)'s generator; ~6 % of the level-0 pixels are FAST-20 corners, 10 % over the pyramid),
"sparse" (camera-like corner density: ~1-2 % at level 0 -- smooth shading, a tenth of the rectangles, half the pixel noise),
"lowcontrast" (corners mostly in the [minTh, iniTh) band: the per-cell threshold retry), "constant" (no corner at all).
"""
import numpy as np


def _upsample(g, w, h):
    gh, gw = g.shape
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, gh - 1); x1 = np.minimum(x0 + 1, gw - 1)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x1]; c = g[y1][:, x0]; d = g[y1][:, x1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def gen_image(w, h, seed, kind="textured"):
    rng = np.random.default_rng(seed)
    if kind == "constant":
        return np.full((h, w), 128, np.uint8)
    img = np.zeros((h, w), np.float64)
    amp = 1.0
    for o in range(8):
        gw = max(2, (w >> (7 - o)) + 2); gh = max(2, (h >> (7 - o)) + 2)
        img += amp * _upsample(rng.standard_normal((gh, gw)), w, h)
        amp *= 0.6
    img = (img - img.mean()) / (img.std() + 1e-9)
    if kind == "lowcontrast":
        # faint structure: most FAST cells hold corners only in the [minTh, iniTh) band -> exercises the
        # per-cell ini->min threshold fallback (ORBextractor.cc:1118-1125)
        img = 128 + 6.0 * img
        for _ in range(max(1, w * h // 900)):
            rw = int(rng.integers(3, 30)); rh = int(rng.integers(3, 30))
            x = int(rng.integers(0, max(1, w - rw))); y = int(rng.integers(0, max(1, h - rh)))
            img[y:y + rh, x:x + rw] += float(rng.integers(9, 17)) * (1 if rng.random() < 0.5 else -1)
        img += rng.normal(0, 0.7, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if kind == "sparse":
        # smooth shading (the fine octaves of the value noise damped) under a few objects: corners sit where rectangle edges meet,
        # not all over the texture -- the corner density of camera imagery rather than of SURVEY 8(d)'s stress pattern
        rng2 = np.random.default_rng(seed + 104729)
        smooth = np.zeros((h, w), np.float64)
        amp = 1.0
        for o in range(5):
            gw = max(2, (w >> (7 - o)) + 2); gh = max(2, (h >> (7 - o)) + 2)
            smooth += amp * _upsample(rng2.standard_normal((gh, gw)), w, h)
            amp *= 0.5
        smooth = (smooth - smooth.mean()) / (smooth.std() + 1e-9)
        img = 115 + 30 * smooth + 3.0 * img
        for _ in range(max(1, w * h // 2500)):
            rw = int(rng.integers(6, 61)); rh = int(rng.integers(6, 61))
            x = int(rng.integers(0, max(1, w - rw))); y = int(rng.integers(0, max(1, h - rh)))
            c = float(rng.integers(30, 101)) * (1 if rng.random() < 0.5 else -1)
            img[y:y + rh, x:x + rw] += c
        for _ in range(max(1, w * h // 4000)):
            r = int(rng.integers(1, 4))
            x = int(rng.integers(r, w - r)); y = int(rng.integers(r, h - r))
            c = float(rng.integers(40, 121)) * (1 if rng.random() < 0.5 else -1)
            yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
            img[y - r:y + r + 1, x - r:x + r + 1] += c * (xx * xx + yy * yy <= r * r)
        img += rng.normal(0, 1.0, img.shape)
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)
    img = 110 + 45 * img
    for _ in range(max(1, w * h // 600)):
        rw = int(rng.integers(3, 41)); rh = int(rng.integers(3, 41))
        x = int(rng.integers(0, max(1, w - rw))); y = int(rng.integers(0, max(1, h - rh)))
        c = float(rng.integers(30, 121)) * (1 if rng.random() < 0.5 else -1)
        img[y:y + rh, x:x + rw] += c
    for _ in range(max(1, w * h // 1500)):
        r = int(rng.integers(1, 4))
        x = int(rng.integers(r, w - r)); y = int(rng.integers(r, h - r))
        c = float(rng.integers(40, 121)) * (1 if rng.random() < 0.5 else -1)
        yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
        img[y - r:y + r + 1, x - r:x + r + 1] += c * (xx * xx + yy * yy <= r * r)
    img += rng.normal(0, 2.0, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def gen_stereo_pair(w, h, seed, dmin=2, dmax=40):
    """Right image = left resampled with a smooth positive disparity field + independent noise."""
    left = gen_image(w, h, seed)
    rng = np.random.default_rng(seed + 7919)
    d = _upsample(rng.uniform(dmin, dmax, (4, 5)), w, h)
    xs = np.arange(w)[None, :] + d                 # right(x) = left(x + d)  => uL - uR = d > 0
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1)
    f = xs - np.floor(xs)
    rows = np.arange(h)[:, None]
    right = left[rows, x0] * (1 - f) + left[rows, x1] * f
    right += rng.normal(0, 2.0, right.shape)
    return left, np.clip(np.rint(right), 0, 255).astype(np.uint8)


def gen_vocabulary(k=10, L=6, seed=7, tie_every=997):
    """A seeded complete k-ary vocabulary tree of depth L as arrays (ORBvoc.txt -- k = 10, L = 6, Frame.cc:905-918 uses it with
    levelsup = 4 -- is a missing blob in the reference snapshot): node 0 = root, breadth-first ids (siblings consecutive, as DBoW2's
    HKmeansStep numbers them), random 256-bit node descriptors, idf-like weights, a few stopped words (weight 0).  Every `tie_every`-th
    inner node gets two children with the SAME descriptor: the descent must take the earlier one (strict `d < best_d`,
    TemplatedVocabulary.h:1239-1250).  Returns the dict the vocabulary classes take."""
    rng = np.random.default_rng(seed)
    nnodes = (k ** (L + 1) - 1) // (k - 1)
    parent = np.zeros(nnodes, np.int32)
    ids = np.arange(1, nnodes, dtype=np.int64)
    parent[1:] = (ids - 1) // k
    first_leaf = (k ** L - 1) // (k - 1)
    is_leaf = np.zeros(nnodes, np.uint8); is_leaf[first_leaf:] = 1
    desc = rng.integers(0, 256, (nnodes, 32), dtype=np.uint8)
    inner = np.arange(0, first_leaf, tie_every)
    desc[inner * k + 4] = desc[inner * k + 2]                  # children 2 and 4 (1-based ids k*p + 1 ..): equal distance to everything
    weight = rng.uniform(0.1, 9.0, nnodes)
    weight[first_leaf + rng.integers(0, nnodes - first_leaf, max(1, (nnodes - first_leaf) // 200))] = 0.0
    return dict(k=k, L=L, parent=parent, is_leaf=is_leaf, desc=desc, weight=weight)
