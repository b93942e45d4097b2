#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r02f}; mkdir -p $OUT; ROOTD=$PWD
timeout 600 python -m pytest tests/test_dgt_gpu.py -m gpu -q -k "dpm" 2>&1 | tail -5 | tee $OUT/pytest_dpm.log
for c in 8 10 15; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown --spair-chunk $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_c$c.json"))
print("achunk $c", round(d["ms_per_step"],3), d["kernel_ms"])
PY
done
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o pmc -- \
     python $ROOTD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-round --spair-chunk 6 > $OUT/$n.json 2> $OUT/$n.err
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $ROOTD/tools/pmc_summary.py "$f" | grep "jd::" | head -8 > $OUT/$n.summary.txt && cat $OUT/$n.summary.txt
  rm -rf $OUT/$n
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
