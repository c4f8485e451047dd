#!/bin/bash
# round 3: joint contrastive evaluation + single-buffer sampling backward -- tests and the training benchmark at both widths
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -k "fused or trainer or training_step or unmodified" > gpurun_out/pytest_r3w.log 2>&1; tail -2 gpurun_out/pytest_r3w.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_r3w.log | head
for W in 384 1024; do
  timeout 900 python scripts/train_bench.py --side hip --trainer device --width $W --frames 90 --iters 40 --data-dir /tmp/dtk_train_data_$W > gpurun_out/train_bench_device_$W.json 2> gpurun_out/train_bench_device_$W.err
  cut -c1-380 gpurun_out/train_bench_device_$W.json | cut -c180-380
done
