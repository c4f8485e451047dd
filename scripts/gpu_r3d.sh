#!/bin/bash
# round 3, fourth GPU call: training tests again (tiled im2col, float64 arbiter), training benchmark + kernel trace at
# C = 384, then the rocprofv3 evidence of the default benchmark (kernel trace + PMC traffic passes) tagged r03.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -d .ref_scratch/reference ] && export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q -rA > gpurun_out/pytest_r3d.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3d.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3d.log | head -20
bash scripts/gpu_train_bench.sh 384 noref prof 2>&1 | tail -22
bash scripts/gpu_profile.sh r03 2>&1 | tail -60
