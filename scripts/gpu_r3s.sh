#!/bin/bash
# round 3: 300 iterations of the device-side trainer (stability: losses, memory), HBM-side traffic of the training convolution
# kernels (PMC passes over scripts/wgrad_bench.py)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
export DTK_REFERENCE_ROOT=$PWD/.ref_scratch/reference
timeout 900 python scripts/train_bench.py --side hip --trainer device --width 384 --frames 90 --iters 300 --data-dir /tmp/dtk_train_data_384 --keep-losses gpurun_out/train_long_losses.json > gpurun_out/train_bench_device_384_long.json 2> gpurun_out/train_bench_device_384_long.err
cut -c1-300 gpurun_out/train_bench_device_384_long.json; tail -2 gpurun_out/train_bench_device_384_long.err | cut -c1-200
python - <<PY
import json
d = json.load(open("$R/gpurun_out/train_long_losses.json"))
L = d["losses"]; names = d["names"]
import statistics
for lo, hi in ((0, 20), (140, 160), (280, 300)):
    print(f"iterations {lo}..{hi}:", {n: round(statistics.mean(r[i] for r in L[lo:hi]), 6) for i, n in enumerate(names)})
PY
python scripts/wgrad_bench.py 2>/dev/null | cut -c1-1200 > gpurun_out/wgrad_bench.json; cut -c1-400 gpurun_out/wgrad_bench.json
cd /tmp && rm -rf /tmp/prof_rd /tmp/prof_wr
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_rd -o rd -- python $R/scripts/wgrad_bench.py > /dev/null 2> $R/gpurun_out/wgrad_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_wr -o wr -- python $R/scripts/wgrad_bench.py > /dev/null 2> $R/gpurun_out/wgrad_wr.err
cd $R
rd=$(find /tmp/prof_rd -name "*.db" | head -1); wr=$(find /tmp/prof_wr -name "*.db" | head -1)
python scripts/pmc_traffic.py $rd $wr conv_wgrad_split_kernel,conv5x5_split_kernel,nchw_to_split_kernel,nhwc_to_nchw_kernel,conv_wgrad_reduce_kernel "python scripts/wgrad_bench.py (the three 5x5 layers at 8 frames of 854x476: forward, data gradient, weight gradient; launches averaged over the layers)" > gpurun_out/r03_pmc_traffic_train_convs.json 2>> gpurun_out/wgrad_rd.err
cat gpurun_out/r03_pmc_traffic_train_convs.json | head -50
