#!/bin/bash
# round 3: host + device profile of the training iteration with the fused head, by operator and by source location
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
DTK_TRAIN_TORCHPROF_STACK=1 DTK_TRAIN_TORCHPROF=$R/gpurun_out/train_torchprof_384_v3.txt timeout 900 python scripts/train_bench.py --side hip --width 384 --frames 90 --iters 10 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_torchprof_v3.json 2> gpurun_out/train_bench_torchprof_v3.err
tail -2 gpurun_out/train_bench_torchprof_v3.err; cut -c1-300 gpurun_out/train_bench_torchprof_v3.json
