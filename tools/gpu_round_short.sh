#!/bin/bash
# parity suite + per-workload bench lines (breakdown)
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee $OUT/pytest.log
for w in ${2:-qm9 geom384}; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  tail -2 $OUT/bench_$w.err | cut -c1-300
  python - <<PY
import json
d=json.load(open("$OUT/bench_$w.json"))
print("$w", round(d["ms_per_step"],3), round(d["value"],2), d["kernel_ms"], "graph", d["hip_graph_replay"], "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
done
