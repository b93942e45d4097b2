#!/bin/bash
# final check of a round: GPU parity suite, smoke(), the default bench line, the nf = 384 bench line
OUT=gpurun_out/${1:-r02z}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; wc -l $OUT/bench.json; cut -c1-400 $OUT/bench.json
timeout 600 python bench.py --workload geom384 --steps 20 --warmup 5 --no-cpu-baseline --breakdown > $OUT/bench_geom384.json 2> $OUT/bench_geom384.err
python - <<PY
import json
d=json.load(open("$OUT/bench_geom384.json"))
print("geom384", round(d["ms_per_step"],3), round(d["value"],2), d["kernel_ms"], "graph", d["hip_graph_replay"], "roof", round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
