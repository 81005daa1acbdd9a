#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counters...>   (runs bench.py briefly under rocprofv3 --pmc, prints per-kernel averages)
# extra bench arguments: PMC_BENCH_ARGS="--config c5 --batch 16"
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$out
timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --batch 128 --launch eager $PMC_BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/$out/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in acc.items():
    if 'rocclr' in k: continue
    print(k, {c: round(x / n[k][c]) for c, x in v.items()})
PY
