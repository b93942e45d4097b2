#!/bin/bash
# round-4 records on one box: GPU suite, default bench line, the other workloads, torchrun 1-rank leg, training bench with its CPU leg
OUT=gpurun_out/${1:-r04final}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_qm9.json 2> $OUT/bench_qm9.err; cut -c1-260 $OUT/bench_qm9.json
for w in geom geom384 cond; do
  timeout 900 python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_${w}.json 2> $OUT/bench_$w.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_${w}.json"))
print("$w ms/step=%.3f graph=%.3f whole=%.3f dom=%s frac=%.3f" % (d['ms_per_step'], (d.get('hip_graph_replay') or {}).get('ms_per_step', 0), d['roofline']['whole_step_frac'], d['roofline']['launch_class'], d['roofline']['frac']), {k: round(v['ms_per_step'], 3) for k, v in d['roofline']['classes'].items()})
PY
done
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun_raw.txt 2> $OUT/bench_torchrun.err
grep '^{' $OUT/bench_torchrun_raw.txt | tail -1 > $OUT/bench_qm9_torchrun_1rank.json; python -c "
import json; d = json.load(open('$OUT/bench_qm9_torchrun_1rank.json')); print('torchrun 1 rank:', d['ms_per_step'], d['sharded_round'])"
timeout 600 python tools/train_bench.py --steps 10 --warmup 3 --cpu > $OUT/train_bench_qm9.json 2> $OUT/train_bench.err; cut -c1-300 $OUT/train_bench_qm9.json; python -c "
import json; d = json.load(open('$OUT/train_bench_qm9.json')); print(d['cpu'], d['gpu_over_cpu_forward_backward'])"
