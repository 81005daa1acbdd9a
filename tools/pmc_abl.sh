#!/bin/bash
# usage: tools/pmc_abl.sh <lib-or-empty> <counters...>  -- PMC counters of k_fast3 for an ablation build (tools/abl.py)
lib=$1; shift
cd /tmp && export TMPDIR=/tmp ORBX_SERIAL=1
[ -n "$lib" ] && export ORB_LIB=$GRAFT_REPO_ROOT/$lib
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcabl
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcabl -- python3 $GRAFT_REPO_ROOT/tools/abl.py > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/pmcabl/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in acc.items():
    if 'k_fast3' in k: print('$lib', {c: round(x / n[k][c]) for c, x in v.items()}, '(average per launch; three k_fast3 launches per step)')
PY
