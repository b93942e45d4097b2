#!/bin/bash
# bench lines of a round without the test suite: smoke(), the default line, the other workloads, DPM-solver rounds
OUT=gpurun_out/${1:-r02L}; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-300 $OUT/bench.json
timeout 600 python bench.py --workload cond --batch 313 --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_cond_b313.json 2> $OUT/bench_cond_b313.err
bash tools/gpu_workloads.sh ${1:-r02L}
