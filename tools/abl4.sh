#!/bin/bash
# serial-mode FAST stage time of ablated builds (tools/ab/f4_*.so) against the current library
for v in "" NOQUICK NOAPPEND NOSCORE NONMS "$@"; do
  env ORBX_SERIAL=1 ${v:+ORB_LIB=$PWD/tools/ab/f4_$v.so} python bench.py --steps 10 --warmup 3 --cpu-sample 0 --launch eager 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-full}', d['stage_ms_per_step']['fast'])"
done
