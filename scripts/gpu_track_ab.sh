#!/bin/bash
# A/B of the tracker kernels in a DEV build (DTK_DEBUG switches compiled in), then the production build + P3 tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp dino_tracker_amd/csrc/libdtk.so /tmp/libdtk_prod.so
make -C dino_tracker_amd/csrc clean > /dev/null
make -C dino_tracker_amd/csrc -j16 DEV=1 > gpurun_out/dev_build.log 2>&1 || { tail -20 gpurun_out/dev_build.log; exit 1; }
rm -f gpurun_out/track_ab.log
for dbg in 0 524288 8192; do
  DTK_DEBUG=$dbg timeout 600 python scripts/prof_peaks.py 30 2700000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/track_ab.log
done
cp /tmp/libdtk_prod.so dino_tracker_amd/csrc/libdtk.so
timeout 900 python -m pytest tests/test_gpu_p3.py tests/test_gpu_bench_path.py -m gpu -q -x 2>&1 | tail -5
