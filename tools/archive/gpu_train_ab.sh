#!/bin/bash
# A/B of training-step variants on one box:  gpurun -- 'bash tools/gpu_train_ab.sh 128 "" "--save-always" "--train-opt 0=0 --train-opt 1=0"'
B=${1:-128}; shift
for rep in 1 2; do
for v in "$@"; do
  timeout 600 python tools/train_bench.py --batch $B --steps 10 --warmup 3 $v 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('B=$B [$v] loader %.2f ms fresh %.2f ms fixed %.2f ms fwd %.2f bwd %.2f' % (d['loader_batches']['s_per_step']*1e3, d['fresh_batches']['s_per_step']*1e3, d['fixed_batch']['s_per_step']*1e3, d['forward_ms'], d['backward_ms']))"
done
done
