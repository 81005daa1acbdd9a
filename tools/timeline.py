#!/usr/bin/env python3
"""Prints the kernel timeline of the last complete step from a rocprofv3 kernel trace (start / end / duration in us, queue):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0
   python tools/timeline.py gpurun_out/tl"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('orbxk::', '').replace('orbmk::', '')[:28], r['Queue_Id'])
            for r in csv.DictReader(open(f)))
idx = [i for i, e in enumerate(ev) if 'knn2' in e[2] or 'k_track_claim' in e[2]]   # the last kernel of a step's match leg
a, b = idx[-3] + 1, idx[-2] + 1
t0 = ev[a][0]
for e in ev[a:b]:
    print("%9.1f %9.1f %8.1f  q%s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
