#!/bin/bash
# rocprofv3 kernel stats of the training step (tools/train_bench.py)
W=${1:-qm9}; TAG=${2:-r04T}; OUT=$PWD/gpurun_out/$TAG; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $ROOTD/tools/train_bench.py --workload $W --steps 3 --warmup 1 ${3:+--batch $3} > $OUT/train_bench_$W.json 2> $OUT/prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/train_kernel_stats_$W.csv && head -45 "$f" | cut -c1-170
find $OUT/prof -name "*kernel_trace.csv" -delete
cat $OUT/train_bench_$W.json | cut -c1-400
