#!/bin/bash
# attention micro-benchmark: timing of all variants + ablations, then one PMC pass (SQ counters) of the three main variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I dino_tracker_amd/csrc \
    scripts/ubench/attn_bench.hip -o /tmp/attn_bench 2> gpurun_out/attn_build.log || { cat gpurun_out/attn_build.log; exit 1; }
timeout 600 /tmp/attn_bench 30 8108 abl > gpurun_out/attn_bench.log 2>&1
cat gpurun_out/attn_bench.log
R=$PWD
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$tag -o pmc -- /tmp/attn_bench 30 8108 > $R/gpurun_out/pmc_attn_$tag.log 2>&1
  db=$(find /tmp/pmc_$tag -name "*.db" | head -1)
  python $R/scripts/pmc_summary.py $db attention > $R/gpurun_out/pmc_attn_$tag.txt 2>&1
  cat $R/gpurun_out/pmc_attn_$tag.txt
done
