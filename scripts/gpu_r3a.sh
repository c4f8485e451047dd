#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (no -x: every failure with its measured values), the benchmark line,
# the end-to-end error from the video for both operand types, a kernel trace of the benchmark for the gap table.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rA --durations=10 > gpurun_out/pytest_gpu_full.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu_full.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_full.log | head -40
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
timeout 600 python scripts/e2e_error.py 476 854 8 8 bf16 > gpurun_out/e2e_bf16.log 2>&1; tail -22 gpurun_out/e2e_bf16.log
cd /tmp && rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03a_kt_bench.json 2> $R/gpurun_out/r03a_kt.err
db=$(find /tmp/prof_kt -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db --gaps --gaps-window patch_embed 3 15 > $R/gpurun_out/r03a_kernel_trace_table.md 2>> $R/gpurun_out/r03a_kt.err
grep -n -i "gap ends" -A 8 $R/gpurun_out/r03a_kernel_trace_table.md
