#!/bin/bash
# round 3, third GPU call (with a scratch copy of the reference next to the snapshot): training tests incl. the new MFMA
# convolutions and the un-modified train.py, the reference-script tests + their reference-free twins, BASELINE config 2 as a
# bench line, the training benchmark at C = 384 with a kernel trace.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -d .ref_scratch/reference ] && export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 2400 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_scripts.py -m gpu -q -rA > gpurun_out/pytest_r3c.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3c.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert|Runtime)" gpurun_out/pytest_r3c.log | head -40
grep -E "rel err y|end to end" gpurun_out/pytest_r3c.log | head -12
timeout 900 python bench.py --frames 50 --queries 256 --steps 5 --warmup 2 > gpurun_out/bench_config2.json 2> gpurun_out/bench_config2.err
cat gpurun_out/bench_config2.json | cut -c1-900; tail -3 gpurun_out/bench_config2.err
bash scripts/gpu_train_bench.sh 384 noref prof 2>&1 | tail -45
