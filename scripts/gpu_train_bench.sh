#!/bin/bash
# One gpurun call: BASELINE.json config 5 (test-time training, train.py un-modified) at full size on the device and on the
# box's host cores, plus a rocprofv3 kernel trace of the device run.  Needs scripts/stage_reference.sh first.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
W=${1:-1024}
D=/tmp/dtk_train_data_$W
timeout 1200 python scripts/train_bench.py --side hip --width $W --frames 90 --iters 14 --data-dir $D > gpurun_out/train_bench_hip_$W.json 2> gpurun_out/train_bench_hip_$W.err
cat gpurun_out/train_bench_hip_$W.json; tail -3 gpurun_out/train_bench_hip_$W.err
if [ "$2" = "ref" ]; then
  timeout 1500 python scripts/train_bench.py --side reference --width $W --frames 90 --iters 3 --data-dir $D > gpurun_out/train_bench_ref_$W.json 2> gpurun_out/train_bench_ref_$W.err
  cat gpurun_out/train_bench_ref_$W.json; tail -3 gpurun_out/train_bench_ref_$W.err
fi
if [ "$3" = "prof" ]; then
  cd /tmp && timeout 900 python $OLDPWD/scripts/train_bench.py --side hip --width $W --frames 90 --iters 6 --data-dir $D --profile-dir /tmp/train_prof > $OLDPWD/gpurun_out/train_prof_$W.json 2> $OLDPWD/gpurun_out/train_prof_$W.err
  cd $OLDPWD
  DB=$(ls /tmp/train_prof/*.db /tmp/train_prof/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python scripts/rocpd_summary.py $DB > gpurun_out/train_kernel_trace_$W.md && head -30 gpurun_out/train_kernel_trace_$W.md
  ls /tmp/train_prof | head
fi
