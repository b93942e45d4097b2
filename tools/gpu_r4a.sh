#!/bin/bash
# round 4, first pass: the float64-yardstick parity tests, the adversarial-weights stress of the rotated statistics, the RCCL leg
# at world size 1, then the rest of the GPU suite and a bench line (what the centred Gram tiles cost)
OUT=gpurun_out/${1:-r04a}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests/test_dgt_gpu.py -m gpu -q -k "adversarial or rotated" -s 2>&1 | tail -25 > $OUT/pytest_rot.txt; tail -12 $OUT/pytest_rot.txt
timeout 900 python -m pytest tests/test_callers_gpu.py -m gpu -q -k "rccl" 2>&1 | tail -15 > $OUT/pytest_rccl.txt; tail -6 $OUT/pytest_rccl.txt
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_callers_gpu.py::test_rccl_leg_at_world_size_one -k "not adversarial and not rotated" 2>&1 | tail -30 > $OUT/pytest_rest.txt; tail -8 $OUT/pytest_rest.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_qm9.json 2> $OUT/bench_qm9.err; tail -3 $OUT/bench_qm9.err; cut -c1-400 $OUT/bench_qm9.json
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_torchrun_1rank.json 2> $OUT/bench_torchrun.err; tail -2 $OUT/bench_torchrun.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_torchrun_1rank.json"))
    print("torchrun 1 rank:", d["ms_per_step"], d.get("sharded_round"))
except Exception as e:
    print("torchrun leg:", e)
PY
