#!/bin/bash
# experiment: blur directly behind the resize chain (beside FAST) for both blur kernels
p='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), {k: round(v,3) for k,v in d["stage_ms_per_step"].items()})'
for env in "" "ORBX_BLUR_EARLY=1" "ORBX_BLUR_MFMA=1" "ORBX_BLUR_MFMA=1 ORBX_BLUR_EARLY=1"; do
  echo "== $env"
  env $env python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$p"
done
