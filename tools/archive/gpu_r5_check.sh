#!/bin/bash
# round-5 check on one box: (optionally) the GPU suite, then a short bench line per workload with the class timers
#   gpurun -- 'bash tools/gpu_r5_check.sh <tag> [tests: all|fast|none] [workloads...]'
TAG=${1:-r05check}; TESTS=${2:-all}; shift; shift
WL=${@:-qm9 geom geom384 cond}
OUT=gpurun_out/$TAG; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
if [ "$TESTS" = "all" ]; then
  timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
elif [ "$TESTS" = "fast" ]; then
  timeout 1800 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "fixture or pinned or pair_path or full_size" 2>&1 | tail -30 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
fi
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
for w in $WL; do
  timeout 900 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --no-full-round > $OUT/bench_${w}.json 2> $OUT/bench_$w.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_${w}.json"))
    print("$w ms/step=%.3f graph=%.3f whole=%.3f" % (d['ms_per_step'], (d.get('hip_graph_replay') or {}).get('ms_per_step', 0), d['roofline']['whole_step_frac']), {k: round(v['ms_per_step'], 3) for k, v in d['roofline']['classes'].items()})
except Exception as e:
    print("$w failed:", e); print(open("$OUT/bench_$w.err").read()[-2000:])
PY
done
