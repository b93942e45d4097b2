#!/bin/bash
# A/B of experiment LIBRARIES (csrc/Makefile LIB=... EXTRA=...) on one box, same bench command, two repetitions each:
#   gpurun -- 'bash tools/gpu_ab_lib.sh qm9 - jodo_amd/csrc/libjodo_hip_x1.so'      ("-" = the product library)
W=${1:-qm9}; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" == "-" ]; then unset JODO_HIP_LIB; else export JODO_HIP_LIB=$PWD/$lib; fi
  timeout 600 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/last.json 2> $OUT/last.err
  python - <<PY
import json
d = json.load(open("$OUT/last.json")); c = d['roofline']['classes']
print("$W [$lib] ms/step=%.3f" % d['ms_per_step'], {k: round(v['ms_per_step'], 3) for k, v in c.items()})
PY
done
done
