#!/bin/bash
# FETCH_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip): prints the counter next to the true bytes per kernel
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/calib
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib -- $GRAFT_REPO_ROOT/tools/ubench/fetch_calib > $GRAFT_REPO_ROOT/gpurun_out/calib.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/calib.log | grep requested
python3 - <<PY
import csv, glob
f = glob.glob('$GRAFT_REPO_ROOT/gpurun_out/calib/*/*counter_collection.csv')[0]
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE' and r['Kernel_Name'].startswith('k_'):
        print(r['Kernel_Name'].split('(')[0], 'FETCH_SIZE', float(r['Counter_Value']), 'KB =', float(r['Counter_Value']) * 1024, 'B')
PY
