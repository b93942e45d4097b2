#!/bin/bash
# per-shape GEMM time inside one real training forward + backward (product log x rocprofv3 kernel trace):  gpurun -- 'bash tools/gpu_gemm_trace.sh'
export TMPDIR=/tmp; OUT=$PWD/gpurun_out/gt; mkdir -p $OUT; R=$PWD; rm -f /tmp/g.log
( cd /tmp && JODO_TRAIN_GEMM_LOG=/tmp/g.log timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o t -- python $R/tools/train_gemm_shapes.py > $OUT/shapes.txt 2> $OUT/err )
f=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
n=$(head -1 $OUT/shapes.txt | awk '{print $1}'); n=$(grep -o "^[0-9]* products" $OUT/shapes.txt | head -1 | awk '{print $1}')
python3 tools/train_gemm_trace.py /tmp/g.log "$f" ${n:-617} | tee $OUT/gemm_trace.txt
rm -rf $OUT/p
