#!/bin/bash
# the other three workloads: bench line without class brackets, then the profile passes (kernel stats + FETCH / WRITE)
for w in ${1:-geom cond geom384}; do
  mkdir -p gpurun_out/r05P
  timeout 900 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r05P/bench_$w.json 2> gpurun_out/r05P/bench_$w.err
  python -c "
import json; d = json.load(open('gpurun_out/r05P/bench_$w.json')); print('$w', d['ms_per_step'], d['roofline']['launch_class'], d['roofline']['frac'], d['roofline']['whole_step_frac'])"
  bash tools/gpu_profile.sh $w r05P > gpurun_out/r05P/profile_$w.log 2>&1; tail -3 gpurun_out/r05P/profile_$w.log
done
