#!/bin/bash
# round 3, last: the whole GPU suite with the reference staged (un-modified scripts incl. the golden training parity), then the
# driver's own sequence parts that do not need it (smoke, default bench line)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_full_ref.log 2>&1; tail -2 gpurun_out/pytest_gpu_full_ref.log
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_full_ref.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-330 gpurun_out/bench_default.json; tail -2 gpurun_out/bench_default.err
