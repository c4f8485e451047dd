#!/bin/bash
# round 3: full GPU suite with the reference staged (incl. the un-modified-script tests), smoke, training at both widths with
# both trainers, kernel trace of the device-side trainer
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_full_ref.log 2>&1; tail -3 gpurun_out/pytest_gpu_full_ref.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_full_ref.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for W in 384 1024; do
  D=/tmp/dtk_train_data_$W
  timeout 900 python scripts/train_bench.py --side hip --trainer device --width $W --frames 90 --iters 30 --data-dir $D > gpurun_out/train_bench_device_$W.json 2> gpurun_out/train_bench_device_$W.err
  cut -c1-420 gpurun_out/train_bench_device_$W.json; tail -2 gpurun_out/train_bench_device_$W.err | cut -c1-200
done
timeout 900 python scripts/train_bench.py --side hip --trainer reference --width 384 --frames 90 --iters 14 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_reftrainer_384.json 2> gpurun_out/train_bench_reftrainer_384.err
cut -c1-420 gpurun_out/train_bench_reftrainer_384.json
cd /tmp && rm -rf /tmp/train_prof && timeout 900 python $R/scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 8 --data-dir /tmp/dtk_train_data_384 --profile-dir /tmp/train_prof > $R/gpurun_out/train_prof_device_384.json 2> $R/gpurun_out/train_prof_device_384.err
cd $R
DB=$(ls /tmp/train_prof/*.db /tmp/train_prof/*/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python scripts/rocpd_summary.py $DB > gpurun_out/train_kernel_trace_device_384.md && head -24 gpurun_out/train_kernel_trace_device_384.md | cut -c1-150
