#!/bin/bash
# round-5 records on one box: GPU suite, default bench line, the other workloads (+ a GEOM batch with molecules above an attention
# group), torchrun 1-rank leg, config 5's 50-NFE DPM rounds, training bench with its CPU leg
OUT=gpurun_out/${1:-r05final}; mkdir -p $OUT
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
print("$w ms/step=%.3f graph=%.3f whole=%.3f kernel=%s frac=%.3f max_n=%d" % (d['ms_per_step'], (d.get('hip_graph_replay') or {}).get('ms_per_step', 0), d['roofline']['whole_step_frac'], d['roofline']['launch_class'], d['roofline']['frac'], d['config']['max_n']), {k: round(v['ms_per_step'], 3) for k, v in d['roofline']['classes'].items()})
PY
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun_raw.txt 2> $OUT/bench_torchrun.err
grep '^{' $OUT/bench_torchrun_raw.txt | tail -1 > $OUT/bench_qm9_torchrun_1rank.json; python -c "
import json; d = json.load(open('$OUT/bench_qm9_torchrun_1rank.json')); print('torchrun 1 rank:', d['ms_per_step'], d['sharded_round'])"
rm -f $OUT/full_round_cond_dpm.jsonl
for b in 313 1250; do for g in 0 1; do HIP_GRAPH=$g timeout 300 python tools/full_round.py cond $b 50 fast 2>/dev/null | grep '^{' | tail -1 >> $OUT/full_round_cond_dpm.jsonl; done; done
cut -c1-200 $OUT/full_round_cond_dpm.jsonl
timeout 600 python tools/train_bench.py --steps 10 --warmup 3 --cpu > $OUT/train_bench_qm9.json 2> $OUT/train_bench.err; cut -c1-700 $OUT/train_bench_qm9.json
timeout 600 python tools/train_bench.py --workload geom --steps 10 --warmup 3 > $OUT/train_bench_geom.json 2>> $OUT/train_bench.err; cut -c1-300 $OUT/train_bench_geom.json
