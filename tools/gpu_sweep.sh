#!/bin/bash
# offsets-per-item sweep of the attention kernel: bash tools/gpu_sweep.sh TAG 2 3 4 6
OUT=gpurun_out/${1:-r02q}; mkdir -p $OUT; shift
timeout 600 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "variants or fixture or pair_path or medium" 2>&1 | tail -3
for c in "$@"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --spair-chunk $c > $OUT/sw_$c.json 2> $OUT/sw.err
  python - <<PY
import json
d=json.load(open("$OUT/sw_$c.json"))
print("achunk $c", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done
