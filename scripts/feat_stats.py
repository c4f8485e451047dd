import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import torch
from dino_tracker_amd import ops, synth
from dino_tracker_amd.extractor import VitExtractor
from dino_tracker_amd.tracker import Tracker
from dino_tracker_amd.model_inference import ModelInference
from dino_tracker_amd.dataset import RangeNormalizer
T, H, W = 12, 476, 854
video = synth.synth_video(T, H, W, seed=2000).cuda()
for ls in (1.0, 0.3, 0.1, 0.03):
    ex = VitExtractor("dinov2_vits14", stride=7, device="cuda:0", state_dict=synth.make_vit_weights("dinov2_vits14", 2, layerscale=ls))
    f = ex.encode(video)  # T, HW, C
    fn = torch.nn.functional.normalize(f[0], dim=-1)
    idx = torch.randint(0, fn.shape[0], (2000,), device="cuda")
    cs = (fn[idx[:1000]] * fn[idx[1000:]]).sum(-1)
    nb = (fn[:-1] * fn[1:]).sum(-1)
    trk = Tracker(video=video, dino_features=f, device="cuda:0", track_method=ops.TRACK_MFMA)
    trk.tracker_head.load_state_dict(synth.synth_head_weights(3)); trk.to("cuda:0").eval()
    trk.refined_features = None
    trk._refined, trk._refined_norms = trk._packed_dino()   # (the raw volume's norms are made on demand since round 5: ADVICE r5)
    mi = ModelInference(trk, RangeNormalizer((W, H, T), device="cuda:0"), 0.7, 0.6)
    q = synth.grid_queries(8, 8, H, W, 0).cuda()
    ops.profile_enable(True)
    traj, occ = mi.infer(q)
    prof = ops.profile_collect(); ops.profile_enable(False)
    err = (traj[:, :, 0] - (q[:, None, 0] - 4.2 * torch.arange(T, device="cuda")[None])).abs().median().item()
    print(f"layerscale {ls}: cos(random pairs) mean {cs.mean():.3f} max {cs.max():.3f}; cos(neighbours) {nb.mean():.3f}; anchors/query {int(mi.last_counts[0])/64:.1f}; "
          f"median |x err| vs true motion {err:.2f}px; occ {occ.float().mean():.2f}; head_exact ms {prof.get('head_exact',(0,0))[0]:.1f} head16 ms {prof.get('head16', (0, 0))[0]:.1f}")
