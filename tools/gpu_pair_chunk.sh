#!/bin/bash
# Pair-update items of several offsets (plan parameter pair_chunk) with and without the cross-offset row prefetch (-DJODO_X_UPD_NEXT=1 library):
#   gpurun -- 'bash tools/gpu_pair_chunk.sh qm9 "1 2 3 4 7" - jodo_amd/csrc/libjodo_hip_next.so'      ("-" = the product library)
W=${1:-qm9}; CH=${2:-"1 2 4"}; shift; shift
OUT=gpurun_out/pair_chunk; mkdir -p $OUT
for lib in "$@"; do
for k in $CH; do
  if [ "$lib" == "-" ]; then unset JODO_HIP_LIB; else export JODO_HIP_LIB=$PWD/$lib; fi
  timeout 600 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --no-split-leg --pair-chunk $k > $OUT/last.json 2> $OUT/last.err
  python - <<PY | tee -a $OUT/summary_$W.txt
import json
d = json.load(open("$OUT/last.json")); c = d['roofline']['classes']
print("$W [$lib] pair_chunk=$k ms/step=%.3f pair update %.3f ms/step" % (d['ms_per_step'], c['edge_update']['ms_per_step']))
PY
done
done
