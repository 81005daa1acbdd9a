import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("orb-slam3_amd")
pkg.LIB_PATH = os.environ.get("ORB_LIB", pkg.LIB_PATH)
synth = importlib.import_module("orb-slam3_amd.synth")
B = 128
ex = pkg.ORBextractor(1000, max_size=(752, 480), max_batch=B)
imgs = [synth.gen_image(752, 480, 1 + i % 16) for i in range(B)]
for _ in range(3):
    try:
        ex.extract_batch(imgs)
    except Exception as e:
        pass
print(os.path.basename(pkg.LIB_PATH), {k: round(v, 3) for k, v in ex.timings().items()})
