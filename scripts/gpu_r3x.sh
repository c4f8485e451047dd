#!/bin/bash
# round 3: which calls of a training iteration still synchronise with the device (torch's sync debug mode)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
DTK_TRAIN_SYNC_DEBUG=1 timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 6 --data-dir /tmp/dtk_train_data_384 --keep-stderr gpurun_out/train_sync_debug.err > gpurun_out/train_sync_debug.json 2> gpurun_out/train_sync_debug2.err
grep -c "synchroniz" gpurun_out/train_sync_debug.err
grep -B1 -A3 "synchroniz" gpurun_out/train_sync_debug.err | grep -E "^\s+File|\.py:[0-9]+" | sort | uniq -c | sort -rn | head -30
