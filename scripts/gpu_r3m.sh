#!/bin/bash
# round 3: the fused training head (tests first, then the whole training file, then the benchmark with a kernel trace)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -rA -k "fused_head" -s > gpurun_out/pytest_r3m_head.log 2>&1
grep -E "passed|failed|rel d" gpurun_out/pytest_r3m_head.log | tail -8 | cut -c1-400
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3m_head.log | head -20
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_p3.py -m gpu -q -rA -k "not fused_head" > gpurun_out/pytest_r3m.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3m.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3m.log | head -20
bash scripts/gpu_train_bench.sh 384 noref prof 2>&1 | tail -32 | cut -c1-200
