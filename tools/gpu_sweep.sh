#!/bin/bash
# offsets-per-item sweep of the attention kernel: bash tools/gpu_sweep.sh TAG WORKLOAD 6 10 16
OUT=gpurun_out/${1:-r02q}; mkdir -p $OUT; W=$2; shift; shift
timeout 600 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "fixture or pair_path or medium" 2>&1 | tail -3
for c in "$@"; do
  timeout 400 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-full-round --breakdown --spair-chunk $c > $OUT/sw_$c.json 2> $OUT/sw.err
  python - <<PY
import json
d=json.load(open("$OUT/sw_$c.json"))
print("$W achunk $c", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done
