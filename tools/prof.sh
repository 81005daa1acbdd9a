#!/bin/bash
# usage: tools/prof.sh <name> [bench args]: rocprofv3 --kernel-trace --stats of a short bench run; per-kernel summary -> gpurun_out/<name>_kernel_stats.csv
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$name
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 --launch eager "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp $f $GRAFT_REPO_ROOT/gpurun_out/${name}_kernel_stats.csv; head -25 $f | cut -c1-160; else ls -R $GRAFT_REPO_ROOT/gpurun_out/$name | head; tail -5 $GRAFT_REPO_ROOT/gpurun_out/$name.log; fi
