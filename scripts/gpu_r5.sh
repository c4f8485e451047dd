#!/bin/bash
# Round-5 GPU work, one gpurun call per invocation: scripts/gpu_r5.sh <step> [<step> ...]; outputs under gpurun_out/.
#   new_tests     the tests this round added (full-size parity, fp16 range in every frame, > 2 GB split planes, weights file)
#   tests         the whole -m gpu suite;   tests:<expr>  pytest -k <expr>
#   bench         bench.py with the driver's flags (--steps 20 --warmup 5);  bench_quick  --steps 5 --warmup 2 --no-cpu-baseline
#   profile       scripts/gpu_profile.sh r05: kernel trace + stats, PMC traffic passes, SQ pass -> gpurun_out/r05_*
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for WHAT in "$@"; do
  echo "=== $WHAT"
  case $WHAT in
    new_tests)
      timeout 1500 python -m pytest -m gpu -x -q -s tests/test_gpu_fullsize.py tests/test_gpu_p3.py::test_split_planes_beyond_2gb \
          "tests/test_gpu_p1.py::test_fp16_saturation_is_reported_and_bf16_is_the_way_out" \
          "tests/test_gpu_p1.py::test_saturation_in_a_late_frame_is_caught_and_healed" \
          "tests/test_gpu_p1.py::test_weights_from_an_upstream_named_checkpoint_file" \
          "tests/test_gpu_p1.py::test_outlier_tokens_stay_in_fp16_range" 2>&1 | tail -40 | tee gpurun_out/new_tests.log ;;
    tests)
      timeout 2400 python -m pytest -m gpu -x -q tests 2>&1 | tail -25 | tee gpurun_out/tests.log ;;
    tests:*)
      timeout 2400 python -m pytest -m gpu -x -q -s tests -k "${WHAT#tests:}" 2>&1 | tail -40 | tee gpurun_out/tests_k.log ;;
    bench)
      timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
    bench_quick)
      timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-videos30 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; cat gpurun_out/bench_quick.json; tail -5 gpurun_out/bench_quick.err ;;
    bench_w1024)
      timeout 900 python bench.py --width 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-power --no-videos30 > gpurun_out/bench_w1024.json 2> gpurun_out/bench_w1024.err; cat gpurun_out/bench_w1024.json; tail -5 gpurun_out/bench_w1024.err ;;
    bench_config2)   # BASELINE.json configs[1]: 854x480x50, 256 queries
      timeout 900 python bench.py --frames 50 --queries 256 --steps 5 --warmup 2 --no-videos30 --parity-video-frames 50 > gpurun_out/bench_config2.json 2> gpurun_out/bench_config2.err; cat gpurun_out/bench_config2.json; tail -3 gpurun_out/bench_config2.err ;;
    bench_w768)
      timeout 900 python bench.py --width 768 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-power --no-videos30 > gpurun_out/bench_w768.json 2> gpurun_out/bench_w768.err; cat gpurun_out/bench_w768.json; tail -3 gpurun_out/bench_w768.err ;;
    attn_ab)   # same-box A / B of the attention stage: scripts/ubench/libdtk_prev.so (a copy of the previous build) vs the tree's library
      timeout 600 python scripts/attn_ab.py scripts/ubench/libdtk_prev.so dino_tracker_amd/csrc/libdtk.so 2>&1 | tee gpurun_out/attn_ab.log ;;
    bench_ab)   # same-box A / B of the whole step: the tree's library, then scripts/ubench/libdtk_prev.so (a copy of the previous build), then the tree's again
      F="--steps 5 --warmup 2 --no-cpu-baseline --no-clock-power --no-videos30 --parity-queries 0"
      L=dino_tracker_amd/csrc/libdtk.so
      timeout 600 python bench.py $F > gpurun_out/bench_ab_new1.json 2> gpurun_out/bench_ab.err
      cp $L /tmp/libdtk_new.so && cp scripts/ubench/libdtk_prev.so $L
      timeout 600 python bench.py $F > gpurun_out/bench_ab_prev.json 2>> gpurun_out/bench_ab.err
      cp /tmp/libdtk_new.so $L
      timeout 600 python bench.py $F > gpurun_out/bench_ab_new2.json 2>> gpurun_out/bench_ab.err
      python - <<'PY'
import json
r = {k: json.load(open(f"gpurun_out/bench_ab_{k}.json")) for k in ("new1", "prev", "new2")}
print("ms per step:", {k: v["ms_per_step"] for k, v in r.items()})
keys = sorted(set().union(*[v["roofline"]["kernel_ms"] for v in r.values()]))
for kk in keys:
    print(f"  {kk:18s}", "  ".join(f"{k} {r[k]['roofline']['kernel_ms'].get(kk, float('nan')):8.3f}" for k in r))
PY
      ;;
    attn_v5)   # the round-5 experiment kernel (two waves per SIMD) against the library's attention4, same process
      timeout 600 python scripts/attn_ab.py dino_tracker_amd/csrc/libdtk.so dino_tracker_amd/csrc/libdtk.so:0x200 dino_tracker_amd/csrc/libdtk.so:0x600 dino_tracker_amd/csrc/libdtk.so:0x100 dino_tracker_amd/csrc/libdtk.so:0xa00 dino_tracker_amd/csrc/libdtk.so:0xe00 dino_tracker_amd/csrc/libdtk.so:0x1200 dino_tracker_amd/csrc/libdtk.so:0x1600 2>&1 | tee gpurun_out/attn_v5.log ;;
    files:*)
      timeout 2400 python -m pytest -m gpu -x -q ${WHAT#files:} 2>&1 | tail -25 | tee gpurun_out/tests_files.log ;;
    profile)
      bash scripts/gpu_profile.sh r05 ;;
    *) echo "unknown step $WHAT" ;;
  esac
done
