"""End-to-end latency of the single-frame entry points on one MI355X (host arrays in and out): median of 200 calls each."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb-slam3_amd"); synth = importlib.import_module("orb-slam3_amd.synth")
l, r = synth.gen_stereo_pair(752, 480, 100)
exl = pkg.ORBextractor(1200, max_size=(752, 480)); exr = pkg.ORBextractor(1200, max_size=(752, 480))
_, kl, dl = exl(l, (0, 0)); _, kr, dr = exr(r, (0, 0))
m = pkg.ORBmatcher(0.7)
sf = exl.GetScaleFactors()
rng = np.random.default_rng(0)
n = len(kl)
u = (kl["x"] - 12 + rng.normal(0, 3, n)).astype(np.float32); v = (kl["y"] + rng.normal(0, 1, n)).astype(np.float32)
view = pkg.FrameView(kr, dr, 752, 480, backend=m)
args = dict(cur_blocked=np.zeros(len(kr), bool), scale_factors=sf, valid=np.ones(n, bool), u=u, v=v, invzc=np.full(n, 0.3, np.float32),
            octave=kl["octave"], angle=kl["angle"], qdesc=dl, mp_obs=np.ones(n, bool), th=15, forward=False, backward=False, mbf=47.9, check_ori=True)
pargs = dict(blocked=np.zeros(len(kr), bool), scale_factors=sf, in_view=np.ones(n, bool), px=u, py=v, pxr=u - 5, view_cos=np.full(n, 0.999, np.float32),
             level=kl["octave"], qdesc=dl, mp_obs=np.ones(n, bool), th=3.0, nnratio=0.8)
res = pkg.ResidentFrame(m, view)


def med(f, reps=200):
    for _ in range(10):
        f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


print("features per frame: %d / %d" % (len(kl), len(kr)))
print("SearchByProjection(Frame,Frame)   host frame each call : %.3f ms" % med(lambda: m.SearchByProjectionFrame(view, **args)))
print("SearchByProjection(Frame,Frame)   resident frame        : %.3f ms" % med(lambda: m.SearchByProjectionFrameResident(res, **args)))
# the same call straight through the C ABI with the argument arrays prepared once (what a C++ caller pays)
import ctypes as C
L = pkg.lib()
keep = [np.ascontiguousarray(a, t) for a, t in ((args["cur_blocked"], np.uint8), (sf, np.float32), (args["valid"], np.uint8), (u, np.float32), (v, np.float32),
        (args["invzc"], np.float32), (kl["octave"], np.int32), (kl["angle"], np.float32), (dl, np.uint8), (args["mp_obs"], np.uint8))]
match = np.zeros(len(kr), np.int32)
ptrs = [a.ctypes.data_as(C.c_void_p) for a in keep]
mp = match.ctypes.data_as(C.c_void_p)
print("  ... C ABI only, arguments prepared once              : %.3f ms" % med(lambda: L.orbm_search_by_projection_frame_resident(m.h, res.h, ptrs[0], ptrs[1], n, *ptrs[2:], 15.0, 0, 0, 47.9, 1, mp)))
print("SearchByProjection(Frame,points)  host frame each call : %.3f ms" % med(lambda: m.SearchByProjectionPoints(view, **pargs)))
print("SearchByProjection(Frame,points)  resident frame        : %.3f ms" % med(lambda: m.SearchByProjectionPointsResident(res, **pargs)))
print("resident frame creation (upload + device grid build)   : %.3f ms" % med(lambda: pkg.ResidentFrame(m, view).close(), 50))
print("grid build (host arrays)                               : %.3f ms" % med(lambda: pkg.FrameView(kr, dr, 752, 480, backend=m), 50))
print("one extraction (host image in, host results out)       : %.3f ms" % med(lambda: exl(l, (0, 0)), 50))
exl.L.orbx_set_stage_timing(exl.h, 0)                        # what the C++ facade does: no per-stage events -> single-graph replay
print("  ... per-stage timing off (graph replay)               : %.3f ms" % med(lambda: exl(l, (0, 0)), 50))
