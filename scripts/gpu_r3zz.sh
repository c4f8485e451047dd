#!/bin/bash
# round 3: training at the reference's width C = 1024 in both operand modes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
for mode in split fp16; do
  DTK_TRAIN_CONV_OPERANDS=$mode timeout 900 python scripts/train_bench.py --side hip --trainer device --width 1024 --frames 90 --iters 40 --data-dir /tmp/dtk_train_data_1024 > gpurun_out/train_bench_device_1024_$mode.json 2> gpurun_out/train_bench_device_1024_$mode.err
  python -c "
import json; t = json.load(open('gpurun_out/train_bench_device_1024_$mode.json')); print('$mode', t['s_per_iteration_median'], t['final_losses']['total'])"
done
