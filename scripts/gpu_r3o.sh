#!/bin/bash
# round 3: device-side trainer (tests, benchmark against the reference's loop on the same models, host/device profile)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -rA -x -s -k "fused or trainer" > gpurun_out/pytest_r3o.log 2>&1
grep -E "passed|failed|rel |device-side|golden \(|exact route" gpurun_out/pytest_r3o.log | tail -14 | cut -c1-420
grep -E "^(FAILED|ERROR)|^E  +" gpurun_out/pytest_r3o.log | head -30 | cut -c1-300
D=/tmp/dtk_train_data_384
timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 30 --data-dir $D > gpurun_out/train_bench_device_384.json 2> gpurun_out/train_bench_device_384.err
cut -c1-700 gpurun_out/train_bench_device_384.json; tail -5 gpurun_out/train_bench_device_384.err | cut -c1-300
DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_device_384.txt timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 10 --data-dir $D > gpurun_out/train_bench_device_384_prof.json 2> gpurun_out/train_bench_device_384_prof.err
tail -3 gpurun_out/train_bench_device_384_prof.err | cut -c1-300
