#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DATTN_NO_V3 -I dino_tracker_amd/csrc \
    scripts/ubench/attn_bench.hip -o /tmp/attn_bench 2> gpurun_out/attn_build.log || { cat gpurun_out/attn_build.log; exit 1; }
timeout 600 /tmp/attn_bench 30 8108 $1 > gpurun_out/attn_bench.log 2>&1
cat gpurun_out/attn_bench.log
