#!/bin/bash
# round-3 call D: stream-interleaved sub-batches (model.n_streams): test + A/B
OUT=gpurun_out/${1:-r03D}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "stream_interleaved or pinned" 2>&1 | tail -15 > $OUT/pytest.txt; tail -3 $OUT/pytest.txt
run() { # tag, workload, streams
  timeout 400 python bench.py --workload $2 --streams $3 --steps 40 --warmup 5 --no-cpu-baseline --no-full-round > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$1.json"))
    print("$1", round(d["ms_per_step"],3), "upd launch", round(d["roofline"]["avg_launch_ms"],4), "whole", round(d["roofline"]["whole_step_frac"],3), "graph", d["hip_graph_replay"])
except Exception as e:
    print("$1 failed", e); print(open("$OUT/$1.err").read()[-1500:])
PY
}
for rep in 1 2; do for k in 1 2 3; do run qm9_s${k}_$rep qm9 $k; done; done
run qm9_s4 qm9 4
for w in geom cond geom384; do for k in 1 2; do run ${w}_s$k $w $k; done; done
