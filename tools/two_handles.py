"""Do two extractor handles in flight overlap the latency-bound tail of one batch (quadtree, descriptors, match) with the
issue-bound head of the other (pyramid + FAST)?  Two independent bench workloads of B/2 frames each, stepped alternately on
their own streams, against one workload of B frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch


def run(wls, steps):
    for w in wls:
        w.prime()
    for _ in range(5):
        for w in wls:
            w.step()
    for w in wls:
        w.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for w in wls:
            w.step()
    for w in wls:
        w.sync()
    dt = time.perf_counter() - t0
    return sum(w.B for w in wls) * steps / dt


for B in (512, 1024):
    a = bench.OrbWorkload(bench.parse_args(["--cpu-sample", "0", "--batch", str(B)]), 0, 0)
    a.download = False
    one = run([a], 20)
    del a
    w1 = bench.OrbWorkload(bench.parse_args(["--cpu-sample", "0", "--batch", str(B // 2)]), 0, 0)
    w2 = bench.OrbWorkload(bench.parse_args(["--cpu-sample", "0", "--batch", str(B // 2)]), 0, 0)
    w1.download = w2.download = False
    two = run([w1, w2], 20)
    del w1, w2
    print("frames per step %4d: one handle %.0f frames/s, two handles of %d in flight %.0f frames/s" % (B, one, B // 2, two))
