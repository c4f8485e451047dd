#!/bin/bash
# round 3, second GPU call: the re-toleranced P1 tests, the end-to-end test with arg-max margins, the N4 NMS tests, and the
# attention micro-benchmark (v2 = library kernel, v3 = pipelined across key tiles) for both operand types.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_p1.py tests/test_gpu_reference_scripts.py tests/test_gpu_bench_path.py -m gpu -q -rA \
    -k "p1 or end_to_end or best_buddies or bb_nms" > gpurun_out/pytest_r3b.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_r3b.log | tail -3
grep -E "^(FAILED|ERROR)|^E  +(Assert|assert)" gpurun_out/pytest_r3b.log | head -40
grep -E "outlier frame|ViT-L|fp16 rel|end to end" gpurun_out/pytest_r3b.log | head -20
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I dino_tracker_amd/csrc \
    scripts/ubench/attn_bench.hip -o /tmp/attn_bench 2> gpurun_out/attn_build.log || { tail -20 gpurun_out/attn_build.log; exit 1; }
timeout 900 /tmp/attn_bench 30 8108 abl > gpurun_out/attn_bench_r3b.log 2>&1
cat gpurun_out/attn_bench_r3b.log
