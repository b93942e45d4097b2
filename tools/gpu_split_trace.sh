#!/bin/bash
# kernel table of a short split-bf16 A/B run (which pair-update kernels ran, how long each took)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-qm9}
rm -rf $R/gpurun_out/split_trace_$WL
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split_trace_$WL -o t -- python $R/tools/split_ab.py --workload $WL --steps 6 > $R/gpurun_out/split_trace_$WL.log 2>&1
f=$(find $R/gpurun_out/split_trace_$WL -name "*kernel_stats.csv" | head -1)
head -14 "$f" | cut -c1-220
find $R/gpurun_out/split_trace_$WL -name "*.db" -delete; find $R/gpurun_out/split_trace_$WL -name "*kernel_trace.csv" -delete
