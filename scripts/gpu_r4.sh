#!/bin/bash
# Round-4 GPU work, one gpurun call per invocation: scripts/gpu_r4.sh <step> [<step> ...]; outputs under gpurun_out/.
#   attn        attention micro-benchmark (v2 library kernel vs v4 one-wave-per-SIMD kernel + ablations)
#   e2e_p2      end-to-end error from the video with Delta-DINO convolution operands split vs fp16 (one oracle pass)
#   bench       bench.py default;  bench_p2fp16  the same with DTK_P2_OPERANDS=fp16
#   tests       the -m gpu suite;  tests:<expr>  pytest -k <expr>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -d .ref_scratch/reference ] && export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
for WHAT in "$@"; do
  echo "=== $WHAT"
  case $WHAT in
    attn)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DATTN_NO_V3 -I dino_tracker_amd/csrc \
          -I scripts/ubench scripts/ubench/attn_bench.hip -o /tmp/attn_bench 2> gpurun_out/attn_build.log || { cat gpurun_out/attn_build.log; continue; }
      timeout 600 /tmp/attn_bench 30 8108 1 > gpurun_out/attn_bench.log 2>&1; cat gpurun_out/attn_bench.log ;;
    slot_rate)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/ubench/slot_rate.hip -o /tmp/slot_rate 2> gpurun_out/slot_build.log || { cat gpurun_out/slot_build.log; continue; }
      timeout 300 /tmp/slot_rate > gpurun_out/slot_rate.log 2>&1; cat gpurun_out/slot_rate.log ;;
    e2e_p2)
      timeout 1500 python scripts/e2e_error.py 476 854 8 8 fp16 split,fp16 > gpurun_out/e2e_p2.log 2>&1
      python - <<'PY'
import json
for r in json.load(open("gpurun_out/e2e_error_476x854x8_fp16.json")):
    print(r["config"]); print("  refined rel err", r["feature_rel_err_refined"], "P1", r["feature_rel_err_P1"])
    print("  decidable", r["px_err_decidable_points"]); print("  all", r["px_err_vs_oracle_on_same_video"])
    print("  beyond 1e-3:", r["points_beyond_1e-3px"], "arbitration failures", r["arbitration_failures"], r["arbitrated"][:4])
    print("  occ mismatch", r["occ_mismatch_same_video"], "without tie queries", r["occ_mismatch_same_video_queries_without_a_tie"], "same features max px", r["px_err_vs_oracle_on_same_features"])
PY
      tail -3 gpurun_out/e2e_p2.log ;;
    bench)
      timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err ;;
    bench_p2fp16)
      DTK_P2_OPERANDS=fp16 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_p2fp16.json 2> gpurun_out/bench_p2fp16.err; cat gpurun_out/bench_p2fp16.json; tail -3 gpurun_out/bench_p2fp16.err ;;
    bench_full)
      timeout 1200 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; cat gpurun_out/bench_full.json; tail -3 gpurun_out/bench_full.err ;;
    peaks_abl)
      # corr_peaks ablations need the development switches: DEV build on the box (the shipped library has none)
      make -C dino_tracker_amd/csrc -B DEV=1 -j16 > gpurun_out/make_dev.log 2>&1 || { tail -5 gpurun_out/make_dev.log; continue; }
      for d in 0 8192 16384 32768 24576 40960 49152 57344 65536 73728; do DTK_DEBUG=$d python scripts/prof_peaks.py 30 2097152 2>&1 | tail -1; done | tee gpurun_out/peaks_abl.log ;;
    twin)
      timeout 2400 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "twin or reference_order" 2>&1 | tail -40 > gpurun_out/pytest_twin.log; tail -30 gpurun_out/pytest_twin.log ;;
    tests)
      timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; tail -25 gpurun_out/pytest_gpu.log ;;
    tests:*)
      timeout 2400 python -m pytest tests -m gpu -q -x -k "${WHAT#tests:}" 2>&1 | tail -40 > gpurun_out/pytest_gpu_k.log; tail -25 gpurun_out/pytest_gpu_k.log ;;
    *) echo "unknown step $WHAT" ;;
  esac
done
