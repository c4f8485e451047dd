#!/bin/bash
# round 3: training tests + benchmark + kernel trace after the coarser im2col
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q -rA > gpurun_out/pytest_r3j.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3j.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3j.log | head -20
bash scripts/gpu_train_bench.sh 384 noref prof 2>&1 | tail -32 | cut -c1-200
