#!/bin/bash
# round 3: host/device profile of the device-side trainer at the reference's width C = 1024
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_device_1024.txt timeout 900 python scripts/train_bench.py --side hip --trainer device --width 1024 --frames 90 --iters 10 --data-dir /tmp/dtk_train_data_1024 > gpurun_out/train_bench_device_1024_prof.json 2> gpurun_out/train_bench_device_1024_prof.err
tail -2 gpurun_out/train_bench_device_1024_prof.err | cut -c1-200; cut -c1-200 gpurun_out/train_bench_device_1024_prof.json
