#!/bin/bash
OUT=gpurun_out/${1:-r04d}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_dgt_gpu.py -m gpu -q -k "train or gradients or backward or gemm or rejected or step_fn" 2>&1 | tail -12 > $OUT/pytest_train.txt; tail -6 $OUT/pytest_train.txt
timeout 600 python tools/train_bench.py --steps 10 --warmup 3 > $OUT/train_bench_qm9.json 2> $OUT/train_bench_qm9.err; tail -2 $OUT/train_bench_qm9.err; cat $OUT/train_bench_qm9.json
timeout 600 python tools/train_bench.py --workload geom --batch 32 --steps 5 --warmup 2 > $OUT/train_bench_geom.json 2> $OUT/train_bench_geom.err; tail -2 $OUT/train_bench_geom.err; cat $OUT/train_bench_geom.json
