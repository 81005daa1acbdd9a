#!/usr/bin/env python3
"""Kernel timeline of the last single-frame call in a rocprofv3 kernel trace of tools/single_probe.py (us relative to its first kernel)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('orbxk::', '')[:28], r['Queue_Id'])
            for r in csv.DictReader(open(f)))
idx = [i for i, e in enumerate(ev) if 'k_fetch_one' in e[2]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = ev[a][0]
for e in ev[a:b]:
    print("%8.1f %8.1f %7.1f  q%s %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
