#!/bin/bash
# stage times with every kernel alone on the GPU (ORBX_SERIAL=1, eager): variants given as env assignments, e.g. tools/serial_ab.sh "" ORBX_FAST_V4=1 ORBX_FAST_V3=1
for v in "$@"; do
  env ORBX_SERIAL=1 $v python bench.py --steps 10 --warmup 3 --cpu-sample 0 --launch eager > gpurun_out/serial.log 2>gpurun_out/serial.err
  tail -1 gpurun_out/serial.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', 'fps', round(d['value']), {k: round(v,3) for k,v in d['stage_ms_per_step'].items() if k in ('fast','pyramid_fast_span','total')})" || tail -3 gpurun_out/serial.err
done
