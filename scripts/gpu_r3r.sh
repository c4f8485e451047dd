#!/bin/bash
# round 3, final: the driver's sequence on a box without the reference (pytest -m gpu -x, smoke, default bench line) and the
# kernel trace of the bench
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_driver_like.log 2>&1; tail -3 gpurun_out/pytest_gpu_driver_like.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-400 gpurun_out/bench_default.json; tail -2 gpurun_out/bench_default.err
cd /tmp && rm -rf /tmp/prof_b && rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/bench_traced.json 2> $R/gpurun_out/bench_traced.err
cd $R
db=$(find /tmp/prof_b -name "*.db" | head -1)
[ -n "$db" ] && python scripts/rocpd_summary.py $db > gpurun_out/r03_bench_kernel_trace_final.md && head -16 gpurun_out/r03_bench_kernel_trace_final.md | cut -c1-140
