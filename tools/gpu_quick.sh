#!/bin/bash
# quick GPU iteration: extractor parity tests + short bench, prints the stage breakdown
python -m pytest tests/test_gpu_extract.py -m gpu -x -q > gpurun_out/pytest_quick.log 2>&1; tail -3 gpurun_out/pytest_quick.log
python bench.py --steps 20 --warmup 3 --cpu-sample 0 "$@" > gpurun_out/bench_quick.log 2>&1
tail -1 gpurun_out/bench_quick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps', round(d['value']), 'frac', round(d['roofline']['frac'],4), {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})" || tail -20 gpurun_out/bench_quick.log
