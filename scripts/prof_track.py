"""Per-kernel timing of the tracker stage on synthetic features (development aid)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from dino_tracker_amd import ops, synth
from gpu_util import make_inference, make_tracker
T, N, C = int(sys.argv[1]), int(sys.argv[2]), 384
H, W = 476, 854
feats = synth.synth_features(T, C, 67, 121, seed=0)
nx = int(round(N ** 0.5))
queries = synth.grid_queries(nx, N // nx, H, W, 0).cuda()
trk = make_tracker(torch.zeros(T, 3, H, W), feats, synth.synth_head_weights(3), method=1)
mi = make_inference(trk, H, W, T)
mi.infer(queries)
ops.profile_enable(True)
mi.infer(queries)
prof = ops.profile_collect()
ops.profile_enable(False)
maps = N * T + int(mi.last_counts[0]) * T
print("DTK_DEBUG", os.environ.get("DTK_DEBUG"), "maps", maps, {k: (round(v[0], 2), v[1]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:7]})

import ctypes
from dino_tracker_amd._lib import lib
out = (ctypes.c_ulonglong * 4)()
lib().dtk_debug_counters(out)
t, g, c, b = list(out)
if t:
    print(f"refine_corr tiles {t} groups/tile {g/t:.2f} cells/group {c/max(g,1):.1f} blocks/tile {b/t:.2f}")
