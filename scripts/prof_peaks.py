"""Fixed-workload timing of the fused correlation + selection kernel (development aid): M sources spread over T frames."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from dino_tracker_amd import ops, synth
from gpu_util import make_tracker
T, M, C = int(sys.argv[1]), int(sys.argv[2]), 384
H, W = 476, 854
feats = synth.synth_features(T, C, 67, 121, seed=0)
trk = make_tracker(torch.zeros(T, 3, H, W), feats, synth.synth_head_weights(3), method=1)
fe = trk.features(True)
g = torch.Generator().manual_seed(0)
tgt = torch.sort(torch.randint(0, T, (M,), generator=g)).values.to(torch.int32).cuda()
cells = torch.randint(0, 67 * 121, (M,), generator=g).cuda()
src_f = torch.randint(0, T, (M,), generator=g).cuda()
emb = fe[0].view(T, 67 * 121, C)[src_f, cells].contiguous()
out = torch.empty(M, 2, device="cuda")
for it in range(2):
    if it == 1:
        ops.profile_enable(True)
    trk.track_sources(fe, emb, None, tgt, None, out, M)
    torch.cuda.synchronize()
prof = ops.profile_collect()
ops.profile_enable(False)
ms = prof["corr_peaks"][0]
print("DTK_DEBUG", os.environ.get("DTK_DEBUG"), f"corr_peaks {ms:.3f} ms  {M * 8576 * 768 / ms / 1e9:.0f} TFLOP/s",
      {k: round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:4]})
