#!/bin/bash
# Regenerates everything under profiles/r02_* that is measured on the GPU box (run through gpurun; results land in gpurun_out/refresh/,
# tools/refresh_collect.py copies them into profiles/).  usage: tools/refresh_profiles.sh bench | prof
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/refresh; mkdir -p $out
if [ "$1" = bench ]; then
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r02_bench_c2.json 2> $out/bench_c2.err; tail -c 300 $out/r02_bench_c2.json; echo
  for c in c3 c4 c5; do python bench.py --config $c --steps 20 --warmup 5 > $out/r02_bench_$c.json 2> $out/bench_$c.err; tail -c 200 $out/r02_bench_$c.json; echo; done
  python bench.py --config c4 --batch 512 --steps 20 --warmup 5 > $out/r02_bench_c4_b512.json 2> $out/bench_c4b.err; tail -c 200 $out/r02_bench_c4_b512.json; echo
  python tools/latency_breakdown.py > $out/r02_latency.txt 2>&1; tail -5 $out/r02_latency.txt
else
  for c in c2 c3 c4 c5; do bash tools/prof.sh refresh/prof_$c --config $c > $out/prof_$c.txt 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/refresh/prof_${c}_kernel_stats.csv $out/r02_kernel_stats_$c.csv 2>/dev/null; echo prof $c done; done
  bash tools/pmc.sh refresh/pmc_fetch FETCH_SIZE > $out/pmc_fetch.txt 2>&1 && echo fetch done
  bash tools/pmc.sh refresh/pmc_write WRITE_SIZE > $out/pmc_write.txt 2>&1 && echo write done
  PMC_BENCH_ARGS="--config c5 --batch 16" bash tools/pmc.sh refresh/pmc_fetch_c5 FETCH_SIZE > $out/pmc_fetch_c5.txt 2>&1 && echo fetch c5 done
  PMC_BENCH_ARGS="--config c5 --batch 16" bash tools/pmc.sh refresh/pmc_write_c5 WRITE_SIZE > $out/pmc_write_c5.txt 2>&1 && echo write c5 done
  bash tools/pmc.sh refresh/pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU > $out/r02_pmc_sq_per_launch.txt 2>&1 && echo sq done
fi
