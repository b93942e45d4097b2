#!/bin/bash
# bench lines of a round without the test suite: smoke(), the default line, the other workloads, DPM-solver rounds
OUT=gpurun_out/${1:-r02L}; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-300 $OUT/bench.json
timeout 600 python bench.py --workload cond --batch 313 --steps 20 --warmup 5 --no-cpu-baseline --no-full-round --breakdown > $OUT/bench_cond_b313.json 2> $OUT/bench_cond_b313.err
bash tools/gpu_workloads.sh ${1:-r02L}
# the launch form the driver uses for N > 1, with one rank (RCCL init, barrier and max-over-ranks timing paths)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-round > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; tail -1 $OUT/bench_torchrun1.err; cut -c1-200 $OUT/bench_torchrun1.json
