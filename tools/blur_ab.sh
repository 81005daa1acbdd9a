#!/bin/bash
# blur A/B: matrix-core k_blur3 vs VALU k_blur2, stages serialized (ORBX_SERIAL) so that the stage times are standalone
p='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), {k: round(v,3) for k,v in d["stage_ms_per_step"].items()})'
ORBX_SERIAL=1 ORBX_BLUR_MFMA=1 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$p"
ORBX_SERIAL=1 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$p"
ORBX_BLUR_MFMA=1 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$p"
python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | tail -1 | python -c "$p"
