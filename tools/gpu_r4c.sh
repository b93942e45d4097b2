#!/bin/bash
# round 4: training tests, whole GPU suite, default bench line (roofline on the dominant class, CPU legs)
OUT=gpurun_out/${1:-r04c}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q -s 2>&1 | tail -40 > $OUT/pytest_train.txt; tail -12 $OUT/pytest_train.txt
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_train_gpu.py 2>&1 | tail -30 > $OUT/pytest_rest.txt; tail -8 $OUT/pytest_rest.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench_qm9.json 2> $OUT/bench_qm9.err; tail -3 $OUT/bench_qm9.err; cut -c1-300 $OUT/bench_qm9.json
