#!/bin/bash
OUT=gpurun_out/${1:-r04_contention2}; mkdir -p $OUT
for b in 450 900 1800; do
  timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --plan-opt 2=1 --plan-opt 7=0 > $OUT/b$b.json 2> $OUT/b$b.err
  python - <<PY
import json
d = json.load(open("$OUT/b$b.json"))
c = d['config']; k = d['kernel_ms']
print("B=%d nodes=%d strips~%d ms/step=%.3f" % ($b, c['nodes_per_step'], (c['nodes_per_step']+31)//32, d['ms_per_step']), {x: round(v, 4) for x, v in k.items()})
PY
done
cd /tmp && export TMPDIR=/tmp
for b in 450 900 1800; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps 5 --warmup 3 --no-cpu-baseline --no-full-round --plan-opt 2=1 --plan-opt 7=0 > /dev/null 2>&1
  f=$(ls $GRAFT_REPO_ROOT/$OUT/prof$b/*/*kernel_stats.csv | head -1); echo "B=$b"; grep -E "k_node_post|k_node_ab|k_node_gram|k_edge_update_sym|k_edge_attn" $f | cut -c1-200 | head -8
done
