#!/bin/bash
# round 3: implicit-GEMM convolutions in the training step (tests, then the benchmark and a profile)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rA -s -k "conv_mfma or fused or trainer or training_step or blurpool or regularisers" > gpurun_out/pytest_r3p.log 2>&1
grep -E "passed|failed|rel err" gpurun_out/pytest_r3p.log | tail -14 | cut -c1-300
grep -E "^(FAILED|ERROR)|^E  +" gpurun_out/pytest_r3p.log | head -30 | cut -c1-300
D=/tmp/dtk_train_data_384
timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 30 --data-dir $D > gpurun_out/train_bench_device_384.json 2> gpurun_out/train_bench_device_384.err
cut -c1-500 gpurun_out/train_bench_device_384.json; tail -5 gpurun_out/train_bench_device_384.err | cut -c1-300
DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_device_384.txt timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 10 --data-dir $D > gpurun_out/train_bench_device_384_prof.json 2> gpurun_out/train_bench_device_384_prof.err
tail -3 gpurun_out/train_bench_device_384_prof.err | cut -c1-300
