#!/bin/bash
# round-3 validation pass: whole GPU suite, smoke(), default bench line, launcher forms of the bench
OUT=gpurun_out/${1:-r03V}; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-300 $OUT/bench.json
# the driver's N > 1 launch form with one rank, and the self-launch path (on a 1-GPU box it must stop with a clear message)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-round > $OUT/bench_torchrun_1rank.json 2> $OUT/bench_torchrun.err; cut -c1-200 $OUT/bench_torchrun_1rank.json
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-full-round > $OUT/bench_selflaunch_2.json 2> $OUT/bench_selflaunch_2.err; grep -h "GPU(s) visible\|Error\|error" $OUT/bench_selflaunch_2.err | head -3
