#!/bin/bash
# A/B of the tracker kernels in a DEV build (DTK_DEBUG switches compiled in): corr_peaks with / without the group prefilter
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
make -C dino_tracker_amd/csrc clean > /dev/null
make -C dino_tracker_amd/csrc -j16 DEV=1 > gpurun_out/dev_build.log 2>&1 || { tail -20 gpurun_out/dev_build.log; exit 1; }
for dbg in 0 524288 8192; do
  DTK_DEBUG=$dbg timeout 600 python scripts/prof_peaks.py 30 2700000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/track_ab.log
done
