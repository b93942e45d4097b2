#!/bin/bash
# round 4, second pass: the training path on the device (GEMM, loss.backward() against the reference's gradients, step function),
# the parity tests that failed in the first pass under their documented multiples, then the rest of the suite
OUT=gpurun_out/${1:-r04b}; mkdir -p $OUT
rm -f gpurun_out/parity_errors.jsonl
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q -x 2>&1 | tail -40 > $OUT/pytest_train.txt; tail -25 $OUT/pytest_train.txt
timeout 1500 python -m pytest tests/test_dgt_gpu.py -m gpu -q -k "adversarial or rotated or uniform_and_per" 2>&1 | tail -25 > $OUT/pytest_rot.txt; tail -12 $OUT/pytest_rot.txt
cp gpurun_out/parity_errors.jsonl $OUT/ 2>/dev/null
