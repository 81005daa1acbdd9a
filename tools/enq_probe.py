"""Where does the host spend its time per step?  Times the graph launch and the download enqueue separately."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
for cfg, extra in (("c2", []), ("c4", []), ("c4", ["--batch", "512"]), ("c2", ["--launch", "eager"])):
    args = bench.parse_args(["--config", cfg, "--cpu-sample", "0"] + extra)
    wl = bench.OrbWorkload(args, 0, 0)
    wl.prime()
    for _ in range(5):
        wl.step()
    wl.sync()
    L = wl.L
    tg = td = 0.0
    t0 = time.perf_counter()
    for _ in range(20):
        blk = wl.k % wl.nblk
        a = time.perf_counter()
        if wl.graph:
            L.orbx_graph_launch(wl.ex.h, wl.k % wl.nslots)
        else:
            wl._enqueue(blk)
        b = time.perf_counter()
        L.orbx_result_download_async(wl.ex.h, wl.h_blk[blk].ptr)
        c = time.perf_counter()
        tg += b - a; td += c - b
        wl.k += 1
    t1 = time.perf_counter()
    wl.sync()
    t2 = time.perf_counter()
    print(cfg, extra, "launch %.3f ms/step, download enqueue %.3f ms/step, loop %.3f, wall %.3f ms/step" % (tg / 20 * 1e3, td / 20 * 1e3, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
    del wl
