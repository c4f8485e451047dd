"""Per-kernel times of the ViT-L / ViT-B encoder alone (the 256 x 256 GEMMs of D = 1024 / 768: gemm_wide_kernel), for the DTK_DEV
ablations of that kernel (DTK_DEBUG bits << 16: 4 no stores, 8 no main loop, 16 every stage from k = 0; results invalid):
    python scripts/dev/wide_abl.py [model] [frames]
Prints one JSON line: ms per launch of every encoder kernel."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dino_tracker_amd import ops, synth  # noqa: E402
from dino_tracker_amd.extractor import VitExtractor  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitl14"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sd = synth.make_vit_weights(name, seed=6, layerscale=0.1)
ex = VitExtractor(name, stride=7, device="cuda:0", state_dict=sd, on_overflow="raise")
video = synth.synth_video(frames, 476, 854, seed=82).cuda()
layer = {"dinov2_vitl14": 15, "dinov2_vitb14": 11, "dinov2_vits14": 11}[name]
try:
    ex.encode(video, layer=layer)
except RuntimeError as err:    # (an ablated pass may leave the fp16 range: the timing is what counts)
    print("warm-up:", str(err)[:80], file=sys.stderr)
torch.cuda.synchronize()
ops.profile_enable(True)
for _ in range(2):
    try:
        ex.encode(video, layer=layer, defer_check=True)
    except RuntimeError:
        pass
prof = ops.profile_collect()
ops.profile_enable(False)
print(json.dumps({"model": name, "frames": frames, "DTK_DEBUG": os.environ.get("DTK_DEBUG", "0"), "wide_v1": ex.gemm_wide_v1,
                  "ms_per_launch": {k: round(ms / max(n, 1), 4) for k, (ms, n) in sorted(prof.items())}}))
