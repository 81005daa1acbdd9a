#!/bin/bash
# A/B of another library / environment against the current build on the C5, C3, C4 bench lines: tools/ab_cfg.sh <lib-or-""> [ENV=VAL ...]
lib=$1; shift
for c in c5 c3 c4; do
  python bench.py --config $c --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/x_$c.log 2>gpurun_out/x_$c.err
  env ${lib:+ORB_LIB=$lib} "$@" python bench.py --config $c --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/y_$c.log 2>gpurun_out/y_$c.err
  python - <<PY
import json
for f in ("gpurun_out/x_$c.log", "gpurun_out/y_$c.log"):
    d = json.loads(open(f).read().strip().splitlines()[-1]); s = d["stage_ms_per_step"]
    print("$c", f[11], round(d["value"]), round(d["value_device_resident"]), round(s["orient_desc"], 4), round(s["blur"], 4), round(s["total"], 4))
PY
done
