#!/bin/bash
# round 3: plain-fp16 operand mode of the training convolutions -- test, micro-benchmark and training benchmark in both modes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_p2.py -m gpu -q -s -k "conv or p2 or training_step" > gpurun_out/pytest_r3z.log 2>&1; tail -2 gpurun_out/pytest_r3z.log; grep -E "rel err \(y|^(FAILED|ERROR)" gpurun_out/pytest_r3z.log | cut -c1-300
for mode in split fp16; do
  DTK_TRAIN_CONV_OPERANDS=$mode python scripts/wgrad_bench.py 2>/dev/null > gpurun_out/wgrad_bench_$mode.json
  DTK_TRAIN_CONV_OPERANDS=$mode timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 40 --data-dir /tmp/dtk_train_data_384 > gpurun_out/train_bench_device_384_$mode.json 2> gpurun_out/train_bench_device_384_$mode.err
done
python - <<PY
import json
for mode in ("split", "fp16"):
    d = json.load(open("gpurun_out/wgrad_bench_%s.json" % mode))
    print(mode, {k: "fwd %.2f dgrad %.2f wgrad %.2f ms" % (v["forward_ms"], v["dgrad_ms"], v["wgrad_ms"]) for k, v in d["layers"].items()})
    t = json.load(open("gpurun_out/train_bench_device_384_%s.json" % mode))
    print(mode, "s/iteration", t["s_per_iteration_median"], "final total loss", t["final_losses"]["total"])
PY
