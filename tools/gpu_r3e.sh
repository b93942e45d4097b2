#!/bin/bash
# round-3 call E: edge-kernel TU with vgpr-form MFMAs + packed fp32 vector code: correctness subset + per-class timing
OUT=gpurun_out/${1:-r03E}; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "fixture or oracle or uniform or pair_path or per_block or pinned or variants or invariants or medium" 2>&1 | tail -15 > $OUT/pytest.txt; tail -3 $OUT/pytest.txt
run() { # tag, workload
  timeout 400 python bench.py --workload $2 --steps 40 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$1.json"))
    print("$1", round(d["ms_per_step"],3), d["kernel_ms"], "frac", round(d["roofline"]["frac"],3), "whole", round(d["roofline"]["whole_step_frac"],3), "graph", round(d["hip_graph_replay"].get("ms_per_step",0),3))
except Exception as e:
    print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
PY
}
run qm9_1 qm9; run qm9_2 qm9
for w in geom cond geom384; do run $w $w; done
