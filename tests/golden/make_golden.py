"""Writes tests/golden/*.npz by running the UN-MODIFIED reference (/root/reference) on CPU through
oracle/ref_harness.py.  Run in the build container only:  python tests/golden/make_golden.py

Inputs are regenerated from seeds by dino_tracker_amd.synth (bit-identical on any machine), so the fixtures hold
only the reference's OUTPUTS plus the few intermediate tensors the parity tests need.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402
from dino_tracker_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (H, W, T, C, queries(nx,ny,t), head_seed, benign, feat_seed, delta: None|seed)
    "p3_small": dict(H=140, W=210, T=8, C=32, q=[(4, 3, 0), (2, 2, 5)], head_seed=3, benign=True, feat_seed=10, delta=None),
    "p3_small_wild": dict(H=140, W=210, T=8, C=32, q=[(4, 3, 2)], head_seed=5, benign=False, feat_seed=11, delta=None),
    "p3_full": dict(H=476, W=854, T=4, C=384, q=[(3, 2, 1)], head_seed=3, benign=True, feat_seed=12, delta=None),
    "p23_small": dict(H=140, W=210, T=5, C=32, q=[(3, 2, 0)], head_seed=3, benign=True, feat_seed=13, delta=4),
}


def build_inputs(cfg):
    ph, pw = 1 + (cfg["H"] - 14) // 7, 1 + (cfg["W"] - 14) // 7
    video = synth.synth_video(cfg["T"], cfg["H"], cfg["W"], seed=cfg["feat_seed"] + 100)
    dino = synth.synth_features(cfg["T"], cfg["C"], ph, pw, seed=cfg["feat_seed"])
    head = synth.synth_head_weights(cfg["head_seed"], cfg["benign"])
    queries = torch.cat([synth.grid_queries(nx, ny, cfg["H"], cfg["W"], t, margin=20.0) for nx, ny, t in cfg["q"]])
    delta = synth.synth_delta_dino_weights(cfg["C"], cfg["delta"]) if cfg["delta"] is not None else None
    return video, dino, head, queries, delta


def run_reference(cfg):
    R = ref_harness.load()
    video, dino, head, queries, delta = build_inputs(cfg)
    tmp = tempfile.mkdtemp()
    emb_path = os.path.join(tmp, "dino_embed_video.pt")
    torch.save(dino, emb_path)
    trk = R.tracker.Tracker(video=video, ckpt_path=tmp, dino_embed_path=emb_path, dino_patch_size=14, stride=7,
                            device="cpu")
    # the reference hard-codes a 1024-wide DeltaDINO (delta_dino.py:9); build it at the embedding width instead
    trk.delta_dino = R.delta_dino.DeltaDINO(channels=[3, 64, 128, 256, cfg["C"]], vit_stride=7)
    if delta is not None:
        trk.delta_dino.load_state_dict(delta)
    trk.tracker_head.load_state_dict(head)
    rn = R.dataset.RangeNormalizer(shapes=(cfg["W"], cfg["H"], cfg["T"]))
    mi = R.model_inference.ModelInference(trk, rn, anchor_cosine_similarity_threshold=0.7,
                                          cosine_similarity_threshold=0.6)
    with torch.no_grad():
        trajs = mi.compute_trajectories(queries, None)
        cs = mi.compute_trajectory_cos_sims(trajs, queries)
        anchors = mi.compute_anchor_trajectories(trajs, cs, None)
        occ = mi.compute_occlusion(trajs, cs, anchors)
        # single-call forward fixtures (Tracker.forward API): track query 0 into every frame
        inp = R.model_inference.generate_trajectory_input(queries[0], video)
        fwd = trk(inp)
        # TrackerHead alone on crafted maps
        hm = torch.zeros(4, 1, dino.shape[-2], dino.shape[-1])
        g = torch.Generator().manual_seed(99)
        hm[1] = torch.rand(1, dino.shape[-2], dino.shape[-1], generator=g)
        hm[2, 0, 0, 0] = 0.9
        hm[3, 0, -1, -1] = 0.7
        hm[3, 0, 3, 4] = 0.7
        head_out = trk.tracker_head(hm.clone())
    out = dict(traj=trajs[..., :2].numpy(), occ=occ.numpy(), cos_sims=cs.numpy(),
               n_anchors=np.array([anchors[i].shape[0] for i in range(len(anchors))]),
               anchor0=anchors[0].numpy(), fwd_q0=fwd.numpy(), head_maps=hm.numpy(), head_out=head_out.numpy(),
               queries=queries.numpy())
    if delta is not None:
        out["refined"] = trk.refined_features.detach().numpy()
    return out


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for name in names:
        res = run_reference(CASES[name])
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **res)
        print(name, {k: v.shape for k, v in res.items()}, "n_anchors", res["n_anchors"].tolist(),
              "occ frac", float(res["occ"].mean()), os.path.getsize(path) // 1024, "KiB")
