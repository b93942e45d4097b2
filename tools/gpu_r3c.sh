#!/bin/bash
# round-3 call C: pinning test + A/B of (unpinned | pinned) x (default build | -amdgpu-mfma-vgpr-form build)
OUT=gpurun_out/${1:-r03C}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_dgt_gpu.py -m gpu -q -x -k "pinned or graph or fixture" 2>&1 | tail -4 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
run() { # tag, lib, extra args
  JODO_HIP_LIB=$2 timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-full-round --breakdown $3 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
d=json.load(open("$OUT/$1.json"))
print("$1", round(d["ms_per_step"],3), d["kernel_ms"], "graph", round(d["hip_graph_replay"].get("ms_per_step",0),3))
PY
}
for rep in 1 2; do
  run nopin_$rep "" "--no-pin"
  run pin_$rep "" ""
  run vgpr_pin_$rep $PWD/jodo_amd/csrc/libjodo_hip_vgpr.so ""
done
for w in geom cond geom384; do
  JODO_HIP_LIB= timeout 400 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --no-full-round --breakdown > $OUT/${w}_pin.json 2> $OUT/${w}.err
  JODO_HIP_LIB=$PWD/jodo_amd/csrc/libjodo_hip_vgpr.so timeout 400 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --no-full-round --breakdown > $OUT/${w}_vgpr.json 2>> $OUT/${w}.err
  python - <<PY
import json
for t in ("pin","vgpr"):
    d=json.load(open("$OUT/${w}_%s.json" % t))
    print("$w", t, round(d["ms_per_step"],3), d["kernel_ms"], "frac", round(d["roofline"]["frac"],3), round(d["roofline"]["whole_step_frac"],3))
PY
done
