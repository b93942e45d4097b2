#!/bin/bash
# A/B of plan options on the default workload:  gpurun -- 'bash tools/gpu_opt.sh r02O "2=14" "2=12"'
OUT=gpurun_out/$1; mkdir -p $OUT; shift
for o in "$@"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --plan-opt $o > $OUT/bench_$o.json 2> $OUT/bench_$o.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$o.json"))
print("$o", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done
