#!/bin/bash
# round 3, near-final GPU call: the driver's sequence (pytest -m gpu -x, smoke, default bench) + training at the reference's width
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_driver_like.log 2>&1; tail -3 gpurun_out/pytest_gpu_driver_like.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-330 gpurun_out/bench_default.json; tail -2 gpurun_out/bench_default.err
if [ -d .ref_scratch/reference ]; then
  export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
  timeout 900 python scripts/train_bench.py --side hip --width 1024 --frames 90 --iters 12 --data-dir /tmp/dtk_train_data_1024 > gpurun_out/train_bench_hip_1024.json 2> gpurun_out/train_bench_hip_1024.err
  cut -c1-400 gpurun_out/train_bench_hip_1024.json; tail -2 gpurun_out/train_bench_hip_1024.err
fi
