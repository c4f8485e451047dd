#!/bin/bash
# round 3, sixth GPU call: what the driver runs at round end (the whole -m gpu suite with -x, smoke(), the default bench line),
# plus a kernel trace of ONE training step of this implementation alone (Tracker.forward train mode + backward, no loss code of
# the reference) to count the launches that are ours.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_driver_like.log 2>&1; tail -4 gpurun_out/pytest_gpu_driver_like.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-700 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
cd /tmp && rm -rf /tmp/prof_ts
rocprofv3 --kernel-trace --stats -d /tmp/prof_ts -o ts -- python -m pytest $R/tests/test_gpu_train.py -q -m gpu -k device_matches_host > $R/gpurun_out/train_step_trace.log 2>&1
db=$(find /tmp/prof_ts -name "*.db" | head -1)
python $R/scripts/rocpd_summary.py $db > $R/gpurun_out/train_step_kernel_trace.md 2>> $R/gpurun_out/train_step_trace.log
head -40 $R/gpurun_out/train_step_kernel_trace.md | cut -c1-150
python - <<PY
import re
rows = [l.split('|') for l in open('$R/gpurun_out/train_step_kernel_trace.md') if l.startswith('| ') and not l.startswith('| kernel')]
calls = sum(int(r[2]) for r in rows if r[2].strip().isdigit())
print('total kernel launches in one training step of this implementation (+ test scaffolding):', calls)
PY
