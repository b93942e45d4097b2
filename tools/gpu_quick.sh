#!/bin/bash
# quick check: a parity subset + bench lines
OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "${K:-fixture or per_block or pair_path or asymmetric or medium or invariants}" 2>&1 | tail -4 | tee $OUT/pytest.log
for w in ${2:-qm9 cond}; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$w.json"))
print("$w", round(d["ms_per_step"],3), round(d["value"],2), d["kernel_ms"], "graph", d["hip_graph_replay"], "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
done
