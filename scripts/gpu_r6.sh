#!/bin/bash
# Round-6 GPU work, one gpurun call per invocation: scripts/gpu_r6.sh <step> [<step> ...]; outputs under gpurun_out/.
#   precision     the round-6 precision tests (split operands, range escalation, auto calibration, e2e under outlier weights) + P1 suite
#   e2e_outlier   scripts/e2e_error.py from the video under outlier / ls1 / bench weights, fast / split / bf16 -> gpurun_out/e2e_error_*
#   tests         the whole -m gpu suite;   tests:<expr>  pytest -k <expr>;   files:<paths>
#   bench         bench.py with the driver's flags (--steps 20 --warmup 5);  bench_quick  --steps 5 --warmup 2 --no-cpu-baseline
#   profile       scripts/gpu_profile.sh r06: kernel trace + stats, PMC traffic passes, SQ pass -> gpurun_out/r06_*
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for WHAT in "$@"; do
  echo "=== $WHAT"
  case $WHAT in
    precision)
      timeout 2400 python -m pytest -m gpu -x -q -s tests/test_gpu_precision.py 2>&1 | tail -150 | tee gpurun_out/precision_tests.log
      timeout 1500 python -m pytest -m gpu -x -q tests/test_gpu_p1.py 2>&1 | tail -15 | tee gpurun_out/p1_tests.log ;;
    e2e_outlier)
      timeout 1500 python scripts/e2e_error.py 476 854 16 16 fp16 default cuda outlier fast,split,bf16 > gpurun_out/e2e_outlier.log 2>&1; tail -80 gpurun_out/e2e_outlier.log
      timeout 1200 python scripts/e2e_error.py 476 854 16 16 fp16 default cuda ls1 fast,split > gpurun_out/e2e_ls1.log 2>&1; tail -50 gpurun_out/e2e_ls1.log ;;
    e2e_full)
      timeout 2400 python scripts/e2e_error.py 476 854 90 32 fp16 default cuda ${E2E_WEIGHTS:-outlier} ${E2E_PREC:-fast,split} > gpurun_out/e2e_full.log 2>&1; tail -80 gpurun_out/e2e_full.log ;;
    tests)
      timeout 3000 python -m pytest -m gpu -x -q tests 2>&1 | tail -25 | tee gpurun_out/tests.log ;;
    tests:*)
      timeout 2400 python -m pytest -m gpu -x -q -s tests -k "${WHAT#tests:}" 2>&1 | tail -40 | tee gpurun_out/tests_k.log ;;
    files:*)
      timeout 2400 python -m pytest -m gpu -x -q -s ${WHAT#files:} 2>&1 | tail -150 | tee gpurun_out/tests_files.log ;;
    bench)
      timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
    bench_quick)
      timeout 900 python bench.py --steps 5 --warmup 2 --no-train --no-cpu-baseline --no-videos30 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; cat gpurun_out/bench_quick.json; tail -5 gpurun_out/bench_quick.err ;;
    bench_fast)   # kernel times only: no oracle legs
      timeout 600 python bench.py --steps 5 --warmup 2 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0 > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err; cat gpurun_out/bench_fast.json; tail -5 gpurun_out/bench_fast.err ;;
    bench_split)
      timeout 900 python bench.py --precision split --steps 3 --warmup 1 --no-train --no-cpu-baseline --no-clock-power --no-videos30 > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err; cat gpurun_out/bench_split.json; tail -5 gpurun_out/bench_split.err ;;
    bench_w1024)
      timeout 900 python bench.py --width 1024 --precision fast --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-clock-power --no-videos30 > gpurun_out/bench_w1024.json 2> gpurun_out/bench_w1024.err; cat gpurun_out/bench_w1024.json; tail -5 gpurun_out/bench_w1024.err ;;
    bench_w768)
      timeout 900 python bench.py --width 768 --precision fast --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-clock-power --no-videos30 > gpurun_out/bench_w768.json 2> gpurun_out/bench_w768.err; cat gpurun_out/bench_w768.json | cut -c1-600; tail -3 gpurun_out/bench_w768.err ;;
    bench_ab)   # same-box A / B of the whole step: the tree's library, then scripts/ubench/libdtk_prev.so (a copy of the previous build), then the tree's again
      F="--steps 5 --warmup 2 --no-train --no-cpu-baseline --no-clock-power --no-videos30 --parity-queries 0"
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
    wide_ab)   # the 256 x 256 GEMMs with / without the half-step fragment prefetch (DTK_VIT_GEMM_WIDE_V1=1: without), C = 1024 and 768
      F="--precision fast --steps 2 --warmup 1 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0"
      for W in 1024 768; do
        timeout 600 python bench.py --width $W $F > gpurun_out/wide_ab_${W}_new.json 2> gpurun_out/wide_ab.err
        DTK_VIT_GEMM_WIDE_V1=1 timeout 600 python bench.py --width $W $F > gpurun_out/wide_ab_${W}_v1.json 2>> gpurun_out/wide_ab.err
      done
      python - <<'PY'
import json
for W in (1024, 768):
    r = {k: json.load(open(f"gpurun_out/wide_ab_{W}_{k}.json")) for k in ("new", "v1")}
    print(f"C = {W}: ms per step:", {k: v["ms_per_step"] for k, v in r.items()})
    keys = sorted(set().union(*[v["roofline"]["kernel_ms"] for v in r.values()]))
    for kk in keys:
        print(f"  {kk:18s}", "  ".join(f"{k} {r[k]['roofline']['kernel_ms'].get(kk, float('nan')):9.3f}" for k in r))
PY
      ;;
    wide_time)   # encoder kernels alone, ViT-L and ViT-B, this tree's wide GEMMs vs their round-5 form (same box)
      : > gpurun_out/wide_time.jsonl
      for M in ${WIDE_MODELS:-dinov2_vitl14 dinov2_vitb14}; do
        timeout 300 python scripts/dev/wide_abl.py $M 30 >> gpurun_out/wide_time.jsonl 2>> gpurun_out/wide_time.err
        DTK_VIT_GEMM_WIDE_V1=1 timeout 300 python scripts/dev/wide_abl.py $M 30 >> gpurun_out/wide_time.jsonl 2>> gpurun_out/wide_time.err
      done
      cat gpurun_out/wide_time.jsonl ;;
    split_ab)   # the split-operand step with this tree's GEMM epilogues vs the direct stores (same box)
      F="--precision split --steps 3 --warmup 1 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0"
      timeout 600 python bench.py $F > gpurun_out/split_ab_new.json 2> gpurun_out/split_ab.err
      DTK_VIT_GEMM_WIDE_V1=1 timeout 600 python bench.py $F > gpurun_out/split_ab_v1.json 2>> gpurun_out/split_ab.err
      python - <<'PY'
import json
r = {k: json.load(open(f"gpurun_out/split_ab_{k}.json")) for k in ("new", "v1")}
print("split step, ms:", {k: v["ms_per_step"] for k, v in r.items()})
for kk in sorted(set().union(*[v["roofline"]["kernel_ms"] for v in r.values()])):
    print(f"  {kk:22s}", "  ".join(f"{k} {r[k]['roofline']['kernel_ms'].get(kk, float('nan')):9.3f}" for k in r))
PY
      ;;
    ln_ab)   # fc2 + the next block's LayerNorm in one launch vs two (DTK_VIT_NO_LN_FUSION=1), whole step, same box, alternating
      F="--steps 5 --warmup 2 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0"
      timeout 600 python bench.py $F > gpurun_out/ln_ab_new1.json 2> gpurun_out/ln_ab.err
      DTK_VIT_NO_LN_FUSION=1 timeout 600 python bench.py $F > gpurun_out/ln_ab_off.json 2>> gpurun_out/ln_ab.err
      timeout 600 python bench.py $F > gpurun_out/ln_ab_new2.json 2>> gpurun_out/ln_ab.err
      python - <<'PY'
import json
r = {k: json.load(open(f"gpurun_out/ln_ab_{k}.json")) for k in ("new1", "off", "new2")}
print("ms per step:", {k: v["ms_per_step"] for k, v in r.items()})
for kk in sorted(set().union(*[v["roofline"]["kernel_ms"] for v in r.values()])):
    print(f"  {kk:18s}", "  ".join(f"{k} {r[k]['roofline']['kernel_ms'].get(kk, float('nan')):8.3f}" for k in r))
PY
      ;;
    wide_abl)   # DTK_DEV ablations of gemm_wide_kernel (scripts/ubench/libdtk_dev.so = a `make DEV=1` build): where a 256 x 256 tile's time goes
      L=dino_tracker_amd/csrc/libdtk.so
      cp $L /tmp/libdtk_keep.so && cp scripts/ubench/libdtk_dev.so $L
      : > gpurun_out/wide_abl.jsonl
      for D in 0 262144 524288 1048576 1310720; do
        DTK_DEBUG=$D timeout 300 python scripts/dev/wide_abl.py dinov2_vitl14 30 >> gpurun_out/wide_abl.jsonl 2>> gpurun_out/wide_abl.err
      done
      cp /tmp/libdtk_keep.so $L
      cat gpurun_out/wide_abl.jsonl ;;
    attn_ab)
      timeout 600 python scripts/attn_ab.py scripts/ubench/libdtk_prev.so dino_tracker_amd/csrc/libdtk.so 2>&1 | tee gpurun_out/attn_ab.log ;;
    profile)
      bash scripts/gpu_profile.sh r06 ;;
    sq1024)   # SQ-counter pass + kernel trace of the C = 1024 step (ViT-L/14)
      R=$PWD; rm -rf /tmp/prof_sq /tmp/prof_kt
      A="--width 1024 --precision fast --steps 1 --warmup 0 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0"
      ( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
          -d /tmp/prof_sq -o sq -- python $R/bench.py $A > /dev/null 2> $R/gpurun_out/r06_sq1024.err )
      python scripts/pmc_sq.py $(find /tmp/prof_sq -name "*.db" | head -1) > gpurun_out/r06_pmc_sq_width1024.md 2>> gpurun_out/r06_sq1024.err; head -30 gpurun_out/r06_pmc_sq_width1024.md
      ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py $A > /dev/null 2>> $R/gpurun_out/r06_sq1024.err )
      python scripts/rocpd_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > gpurun_out/r06_kernel_trace_width1024.md 2>> gpurun_out/r06_sq1024.err; head -24 gpurun_out/r06_kernel_trace_width1024.md ;;
    sq)   # the SQ-counter pass alone (scripts/pmc_sq.py: rows per template instantiation since round 6)
      R=$PWD; rm -rf /tmp/prof_sq; ( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES \
          -d /tmp/prof_sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-train --no-cpu-baseline --no-clock-power --no-live-traffic --no-videos30 --parity-queries 0 > /dev/null 2> $R/gpurun_out/r06_sq.err )
      python scripts/pmc_sq.py $(find /tmp/prof_sq -name "*.db" | head -1) > gpurun_out/r06_pmc_sq.md 2>> gpurun_out/r06_sq.err; head -30 gpurun_out/r06_pmc_sq.md ;;
    cmd:*)
      timeout 2400 bash -c "${WHAT#cmd:}" 2>&1 | tail -60 | tee gpurun_out/cmd.log ;;
    *) echo "unknown step $WHAT" ;;
  esac
done
