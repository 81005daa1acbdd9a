#!/bin/bash
# stage times with every kernel alone on the GPU (ORBX_SERIAL=1, eager): k_fast4 against k_fast3 (extra env via "$@")
for v in "" "ORBX_FAST_V3=1"; do
  env ORBX_SERIAL=1 $v "$@" python bench.py --steps 10 --warmup 3 --cpu-sample 0 --launch eager > gpurun_out/serial.log 2>gpurun_out/serial.err
  tail -1 gpurun_out/serial.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'fps', round(d['value']), {k: round(v,3) for k,v in d['stage_ms_per_step'].items()})"
done
