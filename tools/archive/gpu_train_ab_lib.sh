#!/bin/bash
# A/B of experiment LIBRARIES on the training step, one box:  gpurun -- 'bash tools/gpu_train_ab_lib.sh 128 "" jodo_amd/csrc/libjodo_hip_x.so'
# ("" = the product library); TRAIN_OPTS="--train-opt 3=0" adds train_bench options to every row
B=${1:-128}; shift
OPTS=${TRAIN_OPTS:-}
for rep in 1 2; do
for lib in "$@"; do
  JODO_HIP_LIB=$lib timeout 600 python tools/train_bench.py --batch $B --steps 10 --warmup 3 $OPTS 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('B=$B lib=[$lib] $OPTS loader %.2f ms fresh %.2f ms fixed %.2f ms fwd %.2f bwd %.2f' % (d['loader_batches']['s_per_step']*1e3, d['fresh_batches']['s_per_step']*1e3, d['fixed_batch']['s_per_step']*1e3, d['forward_ms'], d['backward_ms']))"
done
done
