#!/bin/bash
# round 3, GPU call: per-operator host time of the training loop (torch.profiler), C = 384, 8 iterations
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
DTK_TRAIN_TORCHPROF_STACK=1 DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_384.txt timeout 900 python scripts/train_bench.py --side hip --width 384 --frames 90 --iters 8 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_torchprof.json 2> gpurun_out/train_bench_torchprof.err
tail -2 gpurun_out/train_bench_torchprof.err; cut -c1-250 gpurun_out/train_bench_torchprof.json
head -44 gpurun_out/train_torchprof_384.txt | cut -c1-180; grep -n "by source location" -A 50 gpurun_out/train_torchprof_384.txt | cut -c1-330 | head -70
