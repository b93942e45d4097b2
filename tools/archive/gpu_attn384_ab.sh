#!/bin/bash
# A/B of experiment builds of the library on one box (csrc/Makefile LIB=... BUILD=... EXTRA="-DJODO_X_... -mllvm -amdgpu-mfma-vgpr-form"):
#   gpurun -- 'bash tools/gpu_attn384_ab.sh geom384 "" pg6 pg12 pref noldss'      ("" = the product library)
W=${1:-geom384}; shift
OUT=gpurun_out/xab; mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  lib=""; [ -n "$v" ] && lib=$PWD/jodo_amd/csrc/libjodo_hip_$v.so
  JODO_HIP_LIB=$lib timeout 600 python bench.py --workload $W --steps 20 --warmup 4 --no-cpu-baseline --no-full-round > $OUT/last.json 2> $OUT/last.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/last.json")); c = d['roofline']['classes']
    print("$W [${v:-product}] ms/step=%.3f" % d['ms_per_step'], {k: round(x['ms_per_step'], 3) for k, x in c.items()})
except Exception as e:
    print("$W [${v:-product}] failed", e); print(open("$OUT/last.err").read()[-600:])
PY
done
done
