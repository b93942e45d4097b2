#!/bin/bash
# Headline bench + rocprofv3 stats + PMC passes + GEOM / cond benches (no pytest).  Usage: gpurun -- 'bash tools/gpu_round2.sh r01d'
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; cat $OUT/bench.json
# per-kernel-class table: every class bracketed with HIP events (adds ~0.4 ms/step of event packets)
timeout 900 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > $OUT/bench_breakdown.json 2> $OUT/bench_breakdown.err; cat $OUT/bench_breakdown.json
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- \
    python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f" | cut -c1-220 > $OUT/kernel_stats_top.csv && cat $OUT/kernel_stats_top.csv | cut -c1-160
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
timeout 600 python bench.py --workload geom --steps 6 --warmup 2 --breakdown --no-cpu-baseline > $OUT/bench_geom.json 2> $OUT/bench_geom.err; cat $OUT/bench_geom.json
timeout 900 python bench.py --workload geom384 --steps 4 --warmup 2 --breakdown --no-cpu-baseline > $OUT/bench_geom384.json 2> $OUT/bench_geom384.err; cat $OUT/bench_geom384.json
timeout 600 python bench.py --workload cond --steps 10 --warmup 3 --breakdown --no-cpu-baseline > $OUT/bench_cond.json 2> $OUT/bench_cond.err; cat $OUT/bench_cond.json
bash tools/gpu_pmc.sh $TAG 2>&1 | tail -60
