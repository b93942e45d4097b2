#!/bin/bash
# experiment: does a launch's per-item time depend on how many SIMDs run it? (weight stream from L2 shared by every wave)
OUT=gpurun_out/${1:-r04_contention}; mkdir -p $OUT
for b in 100 200 400 900 1818 2500; do
  timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/b$b.json 2> $OUT/b$b.err
  python - <<PY
import json
d = json.load(open("$OUT/b$b.json"))
c = d['config']; k = d['kernel_ms']
print("B=%d nodes=%d edges=%d ms/step=%.3f" % ($b, c['nodes_per_step'], c['directed_edges_per_step'], d['ms_per_step']), {x: round(v, 4) for x, v in k.items()})
PY
done
