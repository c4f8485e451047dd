#!/bin/bash
# round 3: training tests after the alignment kernels / synchronisation-free cosine maps and cycle sampling, then the training
# benchmark (C = 384, torch's own RNG) with and without the per-operator profile of iterations 3..6.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 1500 python -m pytest tests/test_gpu_train.py -m gpu -q -rA > gpurun_out/pytest_r3i.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3i.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3i.log | head -20
timeout 900 python scripts/train_bench.py --side hip --width 384 --frames 90 --iters 14 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_hip_384_v2.json 2> gpurun_out/train_bench_hip_384_v2.err
cut -c1-420 gpurun_out/train_bench_hip_384_v2.json; tail -2 gpurun_out/train_bench_hip_384_v2.err
DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_384_v2.txt timeout 900 python scripts/train_bench.py --side hip --width 384 --frames 90 --iters 8 --data-dir /tmp/dtk_train_data_384 > /dev/null 2> gpurun_out/train_bench_torchprof_v2.err
head -30 gpurun_out/train_torchprof_384_v2.txt | cut -c1-180
