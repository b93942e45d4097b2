#!/bin/bash
# bench lines of the other BASELINE workloads (breakdown of every kernel class) + hybrid DPM-solver rounds (config 5)
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT
for w in qm9 geom cond geom384; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$w.json"))
print("$w", round(d["ms_per_step"],3), round(d["value"],2), d["kernel_ms"], "graph", d["hip_graph_replay"], "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
done
timeout 300 python tools/full_round.py cond 313 50 fast 2>&1 | tail -1 | tee $OUT/full_round_cond_dpm.json
HIP_GRAPH=1 timeout 300 python tools/full_round.py cond 313 50 fast 2>&1 | tail -1 | tee $OUT/full_round_cond_dpm_graph.json
timeout 300 python tools/full_round.py cond 1250 50 fast 2>&1 | tail -1 | tee $OUT/full_round_cond1250_dpm.json
HIP_GRAPH=1 timeout 300 python tools/full_round.py cond 1250 50 fast 2>&1 | tail -1 | tee $OUT/full_round_cond1250_dpm_graph.json
