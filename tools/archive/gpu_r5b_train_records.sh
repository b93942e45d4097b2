#!/bin/bash
# round-5 (second half) training records on one box
OUT=gpurun_out/${1:-r05b_train}; mkdir -p $OUT
timeout 900 python tools/train_bench.py --steps 20 --warmup 3 --cpu > $OUT/train_bench_qm9_b128.json 2> $OUT/err.txt; cut -c1-400 $OUT/train_bench_qm9_b128.json
timeout 900 python tools/train_bench.py --batch 512 --steps 10 --warmup 3 > $OUT/train_bench_qm9_b512.json 2>> $OUT/err.txt; cut -c1-300 $OUT/train_bench_qm9_b512.json
timeout 900 python tools/train_bench.py --batch 2048 --steps 6 --warmup 2 > $OUT/train_bench_qm9_b2048.json 2>> $OUT/err.txt; cut -c1-300 $OUT/train_bench_qm9_b2048.json
timeout 900 python tools/train_bench.py --workload geom --steps 10 --warmup 3 > $OUT/train_bench_geom.json 2>> $OUT/err.txt; cut -c1-300 $OUT/train_bench_geom.json
for v in "" "--train-opt 3=0"; do JODO_OPTIM_FLAT=1 timeout 600 python tools/train_bench.py --steps 20 --warmup 3 $v 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('[$v] loader %.2f fresh %.2f fixed %.2f fwd %.2f bwd %.2f' % (d['loader_batches']['s_per_step']*1e3, d['fresh_batches']['s_per_step']*1e3, d['fixed_batch']['s_per_step']*1e3, d['forward_ms'], d['backward_ms']))" | tee -a $OUT/ab.txt; done
JODO_OPTIM_FLAT=0 timeout 600 python tools/train_bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('[JODO_OPTIM_FLAT=0] loader %.2f fresh %.2f fixed %.2f fwd %.2f bwd %.2f' % (d['loader_batches']['s_per_step']*1e3, d['fresh_batches']['s_per_step']*1e3, d['fixed_batch']['s_per_step']*1e3, d['forward_ms'], d['backward_ms']))" | tee -a $OUT/ab.txt
JODO_TRAIN_RUNAHEAD=0 timeout 600 python tools/train_bench.py --steps 20 --warmup 3 2>/dev/null | python -c "
import json, sys; d = json.loads(sys.stdin.read()); print('[JODO_TRAIN_RUNAHEAD=0] loader %.2f fresh %.2f fixed %.2f fwd %.2f bwd %.2f' % (d['loader_batches']['s_per_step']*1e3, d['fresh_batches']['s_per_step']*1e3, d['fixed_batch']['s_per_step']*1e3, d['forward_ms'], d['backward_ms']))" | tee -a $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/prof -o tfb -- python /root/repo/tools/train_fwd_prof.py 5 1 bwd > /root/repo/$OUT/prof.log 2>&1
find /root/repo/$OUT/prof -name "*kernel_stats.csv" -exec cp {} /root/repo/$OUT/train_fwd_bwd_kernel_stats.csv \;
rm -rf /root/repo/$OUT/prof
