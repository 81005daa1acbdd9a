"""Debug aid: candidates of k_fast4 (default) against k_fast3 (ORBX_FAST_V3=1) on one frame; prints the first differences per level."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("orb-slam3_amd")
synth = importlib.import_module("orb-slam3_amd.synth")
w, h = int(sys.argv[1]) if len(sys.argv) > 1 else 752, int(sys.argv[2]) if len(sys.argv) > 2 else 480
img = synth.gen_image(w, h, 1)
def run():
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_size=(w, h), max_batch=1)
    ex(img, (0, 1000))
    return [ex.level_candidates(l) for l in range(8)]
new = run()
os.environ["ORBX_FAST_V3"] = "1"
old = run()
for l in range(8):
    a, b = new[l], old[l]
    sa = set(map(tuple, a.tolist())); sb = set(map(tuple, b.tolist()))
    extra = sorted(sa - sb, key=lambda t: (t[1], t[0])); missing = sorted(sb - sa, key=lambda t: (t[1], t[0]))
    print("level", l, "new", len(a), "old", len(b), "extra", len(extra), "missing", len(missing), "same order", np.array_equal(a, b))
    print("  extra", extra[:12]); print("  missing", missing[:12])
