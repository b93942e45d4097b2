#!/bin/bash
# A/B of plan options on one box: gpurun -- 'bash tools/gpu_ab.sh qm9 "8=0" "8=1"'   (each argument: a set of --plan-opt values, "-" = defaults)
W=${1:-qm9}; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for rep in 1 2; do
for opts in "$@"; do
  args=""; if [ "$opts" != "-" ]; then for o in $opts; do args="$args --plan-opt $o"; done; fi
  timeout 600 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-full-round $args > $OUT/last.json 2> $OUT/last.err
  python - <<PY
import json
d = json.load(open("$OUT/last.json")); c = d['roofline']['classes']
print("$W [$opts] ms/step=%.3f graph=%.3f" % (d['ms_per_step'], (d.get('hip_graph_replay') or {}).get('ms_per_step', 0)), {k: round(v['ms_per_step'], 3) for k, v in c.items()})
PY
done
done
