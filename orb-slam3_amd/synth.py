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
