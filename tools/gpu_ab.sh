#!/bin/bash
# same-box A/B of plan options: bash tools/gpu_ab.sh TAG "3=0" "3=1" ...
OUT=gpurun_out/${1:-r02k}; mkdir -p $OUT; shift
timeout 600 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "variants or fixture" 2>&1 | tail -3
for rep in 1 2; do for o in "$@"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --plan-opt $o > $OUT/ab_${o}_$rep.json 2> $OUT/ab.err
  python - <<PY
import json
d=json.load(open("$OUT/ab_${o}_$rep.json"))
print("opt $o rep $rep", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done; done
