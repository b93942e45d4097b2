#!/bin/bash
# round-6 records on one box: GPU suite, default bench line (+ rocprofv3 kernel table of that command), the other workloads, torchrun
# 1-rank leg, config 5's DPM rounds, split-bf16 A/B, one training bench
OUT=gpurun_out/${1:-r06final}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_qm9.json 2> $OUT/bench_qm9.err; cut -c1-260 $OUT/bench_qm9.json
for w in geom geom384 cond "geom --seed 80"; do
  tag=$(echo $w | tr -d ' -'); tag=${tag/seed/_seed}
  timeout 900 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_${tag}.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_${tag}.json"))
sp = d.get('split_bf16_opt_in') or {}
print("$w ms/step=%.3f graph=%.3f whole=%.3f frac=%.3f split=%s" % (d['ms_per_step'], (d.get('hip_graph_replay') or {}).get('ms_per_step', 0), d['roofline']['whole_step_frac'], d['roofline']['frac'], sp.get('ms_per_step')), {k: round(v['ms_per_step'], 3) for k, v in d['roofline']['classes'].items()})
PY
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun_raw.txt 2> $OUT/bench_torchrun.err
grep '^{' $OUT/bench_torchrun_raw.txt | tail -1 > $OUT/bench_qm9_torchrun_1rank.json; python -c "
import json; d = json.load(open('$OUT/bench_qm9_torchrun_1rank.json')); print('torchrun 1 rank:', d['ms_per_step'], d['sharded_round'])"
rm -f $OUT/full_round_cond_dpm.jsonl
for b in 313 1250; do for g in 0 1; do HIP_GRAPH=$g timeout 300 python tools/full_round.py cond $b 50 fast 2>/dev/null | grep '^{' | tail -1 >> $OUT/full_round_cond_dpm.jsonl; done; done
cut -c1-200 $OUT/full_round_cond_dpm.jsonl
for w in qm9 geom; do timeout 300 python tools/split_ab.py --workload $w --steps 30 2>&1 | grep -v amdgpu.ids > $OUT/split_ab_$w.txt; tail -1 $OUT/split_ab_$w.txt; done
timeout 600 python tools/train_bench.py --steps 10 --warmup 3 > $OUT/train_bench_qm9.json 2> $OUT/train_bench.err; cut -c1-400 $OUT/train_bench_qm9.json
# the kernel table of the DEFAULT bench command
( cd /tmp && export TMPDIR=/tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_default -o trace -- python $OLDPWD/bench.py > $OLDPWD/$OUT/bench_qm9_under_rocprof.json 2> $OLDPWD/$OUT/prof_default.err )
f=$(find $OUT/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/rocprofv3_kernel_stats_qm9.csv && head -8 "$f" | cut -c1-160
rm -rf $OUT/prof_default
