#!/bin/bash
# Regenerates the round-3 figures under gpurun_out/refresh3/ (copy into profiles/ afterwards): bench lines, kernel stats, PMC passes.
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/refresh3; mkdir -p $out
if [ "$1" = pmc2 ]; then
  :
fi
if [ "$1" = bench ]; then
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r03_bench_c2.json 2> $out/bench_c2.err; tail -c 200 $out/r03_bench_c2.json; echo
  python bench.py --steps 20 --warmup 5 --match knn2 > $out/r03_bench_c2_knn2.json 2> $out/bench_c2k.err; tail -c 200 $out/r03_bench_c2_knn2.json; echo
  for t in sparse lowcontrast mixed; do python bench.py --steps 20 --warmup 5 --texture $t --distinct 48 > $out/r03_bench_c2_$t.json 2> $out/bench_c2_$t.err; tail -c 200 $out/r03_bench_c2_$t.json; echo; done
  python bench.py --steps 20 --warmup 5 --distinct 48 > $out/r03_bench_c2_dense.json 2> $out/bench_c2_dense.err; tail -c 200 $out/r03_bench_c2_dense.json; echo
  for c in c3 c4 c5; do python bench.py --config $c --steps 20 --warmup 5 > $out/r03_bench_$c.json 2> $out/bench_$c.err; tail -c 200 $out/r03_bench_$c.json; echo; done
  python bench.py --config c4 --batch 512 --steps 20 --warmup 5 > $out/r03_bench_c4_b512.json 2> $out/bench_c4b.err; tail -c 200 $out/r03_bench_c4_b512.json; echo
  # the pass with the GPU to itself: blur behind FAST (A/B library only; the default keeps it inside the pass because the step is faster that way)
  ORB_LIB=$GRAFT_REPO_ROOT/orb-slam3_amd/liborbslam3_amd_ab.so ORBX_BLUR_LATE=1 python bench.py --steps 20 --warmup 5 > $out/r03_bench_c2_blur_late.json 2> $out/bench_c2_bl.err; tail -c 200 $out/r03_bench_c2_blur_late.json; echo
  python tools/latency_breakdown.py > $out/r03_latency.txt 2>&1; tail -5 $out/r03_latency.txt
else
  for c in c2 c3 c4 c5; do bash tools/prof.sh refresh3/prof_$c --config $c > $out/prof_$c.txt 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/refresh3/prof_${c}_kernel_stats.csv $out/r03_kernel_stats_$c.csv 2>/dev/null; echo prof $c done; done
  bash tools/prof.sh refresh3/prof_c2k --match knn2 > $out/prof_c2k.txt 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/refresh3/prof_c2k_kernel_stats.csv $out/r03_kernel_stats_c2_knn2.csv 2>/dev/null; echo prof c2 knn2 done
  ORB_LIB=$GRAFT_REPO_ROOT/orb-slam3_amd/liborbslam3_amd_ab.so ORBX_BLUR_LATE=1 bash tools/prof.sh refresh3/prof_c2bl > $out/prof_c2bl.txt 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/refresh3/prof_c2bl_kernel_stats.csv $out/r03_kernel_stats_c2_blur_late.csv 2>/dev/null; echo prof c2 blur late done
  bash tools/prof.sh refresh3/prof_c2s --texture sparse --distinct 48 > $out/prof_c2s.txt 2>&1; cp $GRAFT_REPO_ROOT/gpurun_out/refresh3/prof_c2s_kernel_stats.csv $out/r03_kernel_stats_c2_sparse.csv 2>/dev/null; echo prof c2 sparse done
  bash tools/pmc.sh refresh3/pmc_fetch FETCH_SIZE > $out/pmc_fetch.txt 2>&1 && echo fetch c2 done
  bash tools/pmc.sh refresh3/pmc_write WRITE_SIZE > $out/pmc_write.txt 2>&1 && echo write c2 done
  bash tools/pmc.sh refresh3/pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU > $out/r03_pmc_sq_per_launch.txt 2>&1 && echo sq c2 done
  python tools/make_traffic_json.py gpurun_out/refresh3/pmc_fetch gpurun_out/refresh3/pmc_write 752 480 1000 128 && cp profiles/r03_traffic.json profiles/r03_pmc_fetch_counter_collection.csv profiles/r03_pmc_write_counter_collection.csv $out/
  python tools/make_valu_json.py gpurun_out/refresh3/pmc_sq 752 480 1000 128 && cp profiles/r03_valu.json $out/
  PMC_BENCH_ARGS="--config c5 --batch 16" bash tools/pmc.sh refresh3/pmc_fetch_c5 FETCH_SIZE > $out/pmc_fetch_c5.txt 2>&1 && echo fetch c5 done
  PMC_BENCH_ARGS="--config c5 --batch 16" bash tools/pmc.sh refresh3/pmc_write_c5 WRITE_SIZE > $out/pmc_write_c5.txt 2>&1 && echo write c5 done
  PMC_BENCH_ARGS="--config c5 --batch 16" bash tools/pmc.sh refresh3/pmc_sq_c5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU > $out/r03_pmc_sq_per_launch_c5.txt 2>&1 && echo sq c5 done
  python tools/make_traffic_json.py gpurun_out/refresh3/pmc_fetch_c5 gpurun_out/refresh3/pmc_write_c5 1920 1080 4000 16 c5 17023027 12838642 && cp profiles/r03_traffic_c5.json profiles/r03_pmc_*_c5_counter_collection.csv $out/
  python tools/make_valu_json.py gpurun_out/refresh3/pmc_sq_c5 1920 1080 4000 16 c5 && cp profiles/r03_valu_c5.json $out/
fi
