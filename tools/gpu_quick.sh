#!/bin/bash
# quick GPU iteration: extractor parity tests + short bench, prints the stage breakdown
python -m pytest tests/test_gpu_extract.py -m gpu -x -q > gpurun_out/pytest_quick.log 2>&1; tail -3 gpurun_out/pytest_quick.log
python bench.py --steps 20 --warmup 5 --cpu-sample 0 "$@" > gpurun_out/bench_quick.log 2>gpurun_out/bench_quick.err
tail -1 gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', round(d['value']), 'res', round(d['value_device_resident']), 'frac', round(d['roofline']['frac'],4), 'enq', round(d['host_enqueue_ms_per_step'],3), 'loop', round(d['host_loop_ms_per_step'],3), 'gpuwall', round(d['gpu_wall_ms_per_step'],3), 'ok', d['host_copy_matches_device'], {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -20 gpurun_out/bench_quick.err
