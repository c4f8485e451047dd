#!/bin/bash
# One gpurun call: GPU test suite (+ the reference's scripts when a scratch copy was staged), attention micro-benchmark,
# end-to-end error of the bf16 ViT, benchmark line.  Outputs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -d .ref_scratch/reference ] && export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
WHAT=${1:-all}
if [[ $WHAT == all || $WHAT == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
  tail -15 gpurun_out/pytest_gpu.log
fi
if [[ $WHAT == all || $WHAT == *attn* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I dino_tracker_amd/csrc \
      scripts/ubench/attn_bench.hip -o /tmp/attn_bench 2> gpurun_out/attn_build.log && timeout 600 /tmp/attn_bench > gpurun_out/attn_bench.log 2>&1
  cat gpurun_out/attn_bench.log
fi
if [[ $WHAT == all || $WHAT == *e2e* ]]; then
  timeout 900 python scripts/e2e_error.py 238 322 6 4 > gpurun_out/e2e_small.log 2>&1; tail -25 gpurun_out/e2e_small.log
  timeout 900 python scripts/e2e_error.py 476 854 3 3 > gpurun_out/e2e_full.log 2>&1; tail -25 gpurun_out/e2e_full.log
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
fi
