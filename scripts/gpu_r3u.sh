#!/bin/bash
# round 3: correlation backward per workgroup -- fused-track tests and the training benchmark
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -k "fused or trainer or training_step" > gpurun_out/pytest_r3u.log 2>&1; tail -2 gpurun_out/pytest_r3u.log
timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 40 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_device_384.json 2> gpurun_out/train_bench_device_384.err
cut -c1-420 gpurun_out/train_bench_device_384.json
