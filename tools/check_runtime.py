"""Checks that torch and liborbslam3_amd.so share ONE libamdhip64 in-process (run on the GPU box)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print("torch cuda:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.ones(4, device="cuda")
pkg = importlib.import_module("orb-slam3_amd")
pkg.lib()
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l})
print("\n".join(libs))
synth = importlib.import_module("orb-slam3_amd.synth")
ex = pkg.ORBextractor(1000)
mono, kps, desc = ex(synth.gen_image(752, 480, 1), (0, 1000))
print("extracted", len(kps), "with torch loaded; torch tensor sum", float(x.sum()))
