#!/bin/bash
# round 3, seventh GPU call: where does the HOST spend a training iteration (cProfile of the un-modified train.py on this
# implementation, C = 384), and the SQ-counter pass of the default benchmark (profiles/r03_pmc_sq.md).
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
DTK_TRAIN_CPROFILE=$R/gpurun_out/train_cprofile_384.txt timeout 900 python scripts/train_bench.py --side hip --width 384 --frames 90 --iters 10 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_cprofile.json 2> gpurun_out/train_bench_cprofile.err
tail -2 gpurun_out/train_bench_cprofile.err; cut -c1-300 gpurun_out/train_bench_cprofile.json
grep -E "dino_tracker_amd|dino_tracker.py|train_ops|tracker.py|dataset.py" gpurun_out/train_cprofile_384.txt | head -60
cd /tmp && rm -rf /tmp/prof_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES -d /tmp/prof_sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/r03_sq.err
db=$(find /tmp/prof_sq -name "*.db" | head -1)
python $R/scripts/pmc_sq.py $db > $R/gpurun_out/r03_pmc_sq.md 2>> $R/gpurun_out/r03_sq.err
head -24 $R/gpurun_out/r03_pmc_sq.md | cut -c1-200
