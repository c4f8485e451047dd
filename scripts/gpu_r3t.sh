#!/bin/bash
# round 3: XCD-aware grids of the training convolution kernels -- tests, micro-benchmark, HBM-side traffic
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_p2.py -m gpu -q -k "conv_mfma or training_step or p2 or delta" > gpurun_out/pytest_r3t.log 2>&1; tail -2 gpurun_out/pytest_r3t.log
python scripts/wgrad_bench.py 2>/dev/null > gpurun_out/wgrad_bench.json
python - <<PY
import json
d = json.load(open("$R/gpurun_out/wgrad_bench.json"))
for k, v in d["layers"].items():
    print(k, "fwd %.2f dgrad %.2f wgrad %.2f ms; wgrad %.0f TF" % (v["forward_ms"], v["dgrad_ms"], v["wgrad_ms"], v["wgrad_tflops"]))
PY
cd /tmp && rm -rf /tmp/prof_rd /tmp/prof_wr
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_rd -o rd -- python $R/scripts/wgrad_bench.py > /dev/null 2> $R/gpurun_out/wgrad_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_wr -o wr -- python $R/scripts/wgrad_bench.py > /dev/null 2> $R/gpurun_out/wgrad_wr.err
cd $R
rd=$(find /tmp/prof_rd -name "*.db" | head -1); wr=$(find /tmp/prof_wr -name "*.db" | head -1)
python scripts/pmc_traffic.py $rd $wr conv_wgrad_split_kernel,conv5x5_split_kernel,nchw_to_split_kernel,nhwc_to_nchw_kernel,conv_wgrad_reduce_kernel "python scripts/wgrad_bench.py (the three 5x5 layers at 8 frames of 854x476: forward, data gradient, weight gradient; launches averaged over the layers)" > gpurun_out/r03_pmc_traffic_train_convs_xcd.json 2>> gpurun_out/wgrad_rd.err
python - <<PY
import json
d = json.load(open("$R/gpurun_out/r03_pmc_traffic_train_convs_xcd.json"))
for k, v in d["kernels"].items():
    print(k, "fetch %.0f MB write %.0f MB per launch" % (v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
PY
