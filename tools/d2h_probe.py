"""Host stalls of hipMemcpyAsync (device -> pinned) behind queued kernels: per-call host time over 40 back-to-back steps."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
hip = C.CDLL("libamdhip64.so")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1500000
mode = sys.argv[3] if len(sys.argv) > 3 else "wait"
args = bench.parse_args(["--config", cfg, "--cpu-sample", "0"])
wl = bench.OrbWorkload(args, 0, 0)
wl.download = False
wl.prime()
L = wl.L
st = C.c_void_p(L.orbx_stream(wl.ex.h))
s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1)
evs = [C.c_void_p() for _ in range(64)]
for e in evs: hip.hipEventCreateWithFlags(C.byref(e), 2)
dev = C.c_void_p(); hip.hipMalloc(C.byref(dev), 64 << 20)
pin = C.c_void_p(); hip.hipHostMalloc(C.byref(pin), 64 << 20, 0)
wl.sync()
ts = []
t00 = time.perf_counter()
for k in range(40):
    wl.step()
    ev = evs[k % 64] if mode != "oneev" else evs[0]
    if mode != "nowait":
        hip.hipEventRecord(ev, st); hip.hipStreamWaitEvent(s, ev, 0)
    t0 = time.perf_counter()
    hip.hipMemcpyAsync(pin, dev, C.c_size_t(size), 2, s if mode != "samestream" else st)
    ts.append((time.perf_counter() - t0) * 1e3)
t1 = time.perf_counter()
wl.sync(); hip.hipStreamSynchronize(s)
t2 = time.perf_counter()
print("probe", cfg, size, mode, "enqueue loop %.2f ms, wall %.2f ms; slow calls:" % ((t1 - t00) * 1e3, (t2 - t00) * 1e3), [(i, round(t, 2)) for i, t in enumerate(ts) if t > 0.1])
