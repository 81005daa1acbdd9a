"""Device -> pinned-host copy rate on an otherwise idle GPU: one 64 MB copy on one stream, and the same bytes as 2 / 4 pieces on as many streams."""
import ctypes as C, time
import torch  # (loads the HIP runtime the product uses)
hip = C.CDLL("libamdhip64.so")
N = 64 << 20
dev = C.c_void_p(); assert hip.hipMalloc(C.byref(dev), N) == 0
pin = C.c_void_p(); assert hip.hipHostMalloc(C.byref(pin), N, 0) == 0
ss = []
for _ in range(4):
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; ss.append(s)
for k in (1, 2, 4):
    best = 1e9
    for rep in range(6):
        t0 = time.perf_counter()
        for i in range(k):
            off = N // k * i
            assert hip.hipMemcpyAsync(C.c_void_p(pin.value + off), C.c_void_p(dev.value + off), C.c_size_t(N // k), 2, ss[i]) == 0
        for i in range(k):
            hip.hipStreamSynchronize(ss[i])
        best = min(best, time.perf_counter() - t0)
    print("%d stream(s): %.1f GB/s" % (k, N / best / 1e9))
