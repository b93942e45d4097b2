#!/bin/bash
# attention-kernel probe: new DPM tests, offsets-per-item sweep of k_edge_attn, test durations
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "dpm or fused or graph" --durations=8 2>&1 | tail -25 | tee $OUT/pytest_dpm.log
for c in 1 2 3 4 6; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --spair-chunk $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_c$c.json"))
print("achunk $c", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done
timeout 1500 python -m pytest tests -m gpu -q --durations=30 2>&1 | tail -50 | tee $OUT/pytest_durations.log
