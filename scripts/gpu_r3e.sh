#!/bin/bash
# round 3, fifth GPU call: working-set experiments (frames per ViT pass, sources per tracker round: do smaller batches keep
# the inter-kernel traffic inside the 256 MB Infinity Cache?), then the PMC traffic passes again (kernel-name parsing fixed).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "" "--vit-frame-batch 15" "--vit-frame-batch 10" "--vit-frame-batch 6" "--track-round 262144" "--track-round 131072"; do
  tag=$(echo "default $cfg" | tr -d ' -')
  DTK_BENCH_KERNELS=14 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline $cfg > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err
  python - "$cfg" gpurun_out/sweep_$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
km = d["roofline"]["kernel_ms"]
print(f"{sys.argv[1] or 'default':26s} ms/step {d['ms_per_step']:8.2f}  " + "  ".join(f"{k.replace('vit_', '')}:{v:.1f}" for k, v in km.items()))
PY
done
bash scripts/gpu_profile.sh r03 2>&1 | grep -A 12 '"kernels"' | head -30
