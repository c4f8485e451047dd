"""TEST INFRASTRUCTURE -- the on-disk inputs of the reference's train.py (dino_tracker.py:36-87) for a small synthetic
video, BASELINE.json config 5 at reduced size:

    <data>/video/*.jpg, <data>/masks/*.png           the first T frames of the reference's dataset/horsejump
    <data>/dino_embeddings/dino_embed_video.pt       seeded moving feature field (dino_tracker_amd.synth), C = 1024
    <data>/of_trajectories/{fg,bg}_trajectories.pt   [N, T, 2] px, NaN where untracked -- SYNTHETIC supervision following
                                                     the field's motion (the real ones need RAFT weights)
    <data>/dino_best_buddies/dino_best_buddies_filtered.pt   {"s_t": {source_coords, target_coords, r, cos_sims}}
    <data>/models/dino_tracker/{tracker_head,delta_dino}_1.pt  seeded start weights (iteration 1: dino_tracker.py:104-106
                                                     only loads a checkpoint whose iteration is > 0)
    <data>/train.yaml                                config/train.yaml with small batch sizes, every loss term on from
                                                     the first iteration, frame height 126 (the mask loader fixes the
                                                     width at 854, split_trajectories_to_fg_bg.py:39)
The reference hard-codes a 1024-wide DeltaDINO (delta_dino.py:9), hence C = 1024.
"""
import os
import shutil

import numpy as np
import torch
import yaml

from dino_tracker_amd import synth

# config/train.yaml of the reference (lines 1-64), restated so that the data directory can be built WITHOUT the checkout
# (tests/test_train_vs_reference.py::test_restated_train_yaml_matches_reference pins it against the file where it exists)
TRAIN_YAML = dict(
    checkpoint_interval=2500, video_resw=854, video_resh=476, fg_traj_ratio=0.5, keep_traj_in_cpu=False, train_batch_size=512,
    batch_n_frames=4, total_iterations=10000, lr_delta_dino=0.01, lr_cnn_refiner=0.01, apply_scheduler_every=40,
    scheduler_gamma=0.999, lambda_cyc=0.5, apply_cyc_after=5000, cyc_n_frames=4, cyc_batch_size_per_frame=256,
    cyc_fg_points_ratio=0.7, cyc_thresh=4, cyc_gamma=0.8, lambda_emb_norm=0.0001, lambda_angle=0.0001,
    lambda_cl_dino_bb=0.00025, lambda_cl_ref_bb=0.00005, cl_n_frames=4, cl_points_per_pair=256, cl_fg_points_ratio=0.7,
    cl_temp=0.1, cl_div_dino_bb=700, cl_div_ref_bb=900, apply_cl_ref_after=5000, bb_amb_sig_a=27, bb_amb_sig_b=-5.7, stride=7,
    dino_patch_size=14, anchor_cosine_similarity_threshold=0.7, cosine_similarity_threshold=0.6)

CFG = dict(T=8, C=1024, H=126, W=854, feat_seed=31, head_seed=3, delta_seed=9, start_iter=1, total_iterations=4,
           n_fg=160, n_bg=240, bb_per_pair=48)


def _tracks(g, n, T, W, H, vx, vy, inside):
    """Points moving with the content (-vx, -vy px per frame), NaN outside the frame and at random dropouts."""
    x0 = g.uniform(20, W - 20, n)
    y0 = g.uniform(10, H - 10, n)
    t = np.arange(T)[None, :]
    tr = np.stack([x0[:, None] - vx * t, y0[:, None] - vy * t], axis=2).astype(np.float32)
    bad = (tr[..., 0] < 0) | (tr[..., 0] > W - 1) | (tr[..., 1] < 0) | (tr[..., 1] > H - 1) | (g.uniform(size=(n, T)) < 0.25)
    tr[bad] = np.nan
    return torch.from_numpy(tr)


def _synthetic_video(dst, T, H, W):
    """Frames of dino_tracker_amd.synth.synth_video (content moving 4.2, 2.1 px per frame) and an elliptic foreground mask
    moving with it, as PNG files."""
    from PIL import Image
    video = synth.synth_video(T, H, W, seed=1)
    yy, xx = np.mgrid[0:H, 0:W]
    for t in range(T):
        Image.fromarray((video[t].permute(1, 2, 0).numpy() * 255).astype(np.uint8)).save(os.path.join(dst, "video", f"{t:05d}.png"))
        cx, cy = 0.6 * W - 4.2 * t, 0.55 * H - 2.1 * t
        m = (((xx - cx) / (0.22 * W)) ** 2 + ((yy - cy) / (0.3 * H)) ** 2) <= 1.0
        Image.fromarray((m * 255).astype(np.uint8)).save(os.path.join(dst, "masks", f"{t:05d}.png"))


def build(dst, ref_root, cfg=CFG, overrides=None, synthetic_video=False):
    """`overrides`: entries of train.yaml to replace (default: the small-batch settings of the parity fixture);
    `synthetic_video`: generated frames + masks instead of the reference's horsejump clip (any T, any size)."""
    T, C, H, W = cfg["T"], cfg["C"], cfg["H"], cfg["W"]
    assert ref_root or synthetic_video, "without the reference checkout only the synthetic video can be built"
    src = os.path.join(ref_root, "dataset", "horsejump") if ref_root else None
    for sub in ("video", "masks"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        if synthetic_video:
            continue
        for f in sorted(os.listdir(os.path.join(src, sub)))[:T]:
            shutil.copy(os.path.join(src, sub, f), os.path.join(dst, sub, f))
    if synthetic_video:
        _synthetic_video(dst, T, H, W)
    ph, pw = (H - 14) // 7 + 1, (W - 14) // 7 + 1
    os.makedirs(os.path.join(dst, "dino_embeddings"), exist_ok=True)
    torch.save(synth.synth_features(T, C, ph, pw, seed=cfg["feat_seed"]), os.path.join(dst, "dino_embeddings", "dino_embed_video.pt"))
    ck = os.path.join(dst, "models", "dino_tracker")
    os.makedirs(ck, exist_ok=True)
    it = cfg["start_iter"]
    torch.save(synth.synth_head_weights(cfg["head_seed"]), os.path.join(ck, f"tracker_head_{it}.pt"))
    torch.save(synth.synth_delta_dino_weights(C, cfg["delta_seed"]), os.path.join(ck, f"delta_dino_{it}.pt"))
    g = np.random.default_rng(7)
    vx, vy = 0.6 * 7, 0.3 * 7  # the feature field moves (0.6, 0.3) cells per frame
    os.makedirs(os.path.join(dst, "of_trajectories"), exist_ok=True)
    torch.save(_tracks(g, cfg["n_fg"], T, W, H, vx, vy, True), os.path.join(dst, "of_trajectories", "fg_trajectories.pt"))
    torch.save(_tracks(g, cfg["n_bg"], T, W, H, vx, vy, False), os.path.join(dst, "of_trajectories", "bg_trajectories.pt"))
    bb = {}
    K = cfg["bb_per_pair"]
    for s in range(T):
        for t in range(T):
            if s == t:
                continue
            sx = g.uniform(30, W - 30, K)
            sy = g.uniform(12, H - 12, K)
            tx = np.clip(sx - vx * (t - s) + g.normal(size=K), 7, W - 8)
            ty = np.clip(sy - vy * (t - s) + g.normal(size=K), 7, H - 8)
            bb[f"{s}_{t}"] = {
                "source_coords": torch.from_numpy(np.stack([sx, sy], 1).astype(np.float32)),
                "target_coords": torch.from_numpy(np.stack([tx, ty], 1).astype(np.float32)),
                "r": torch.from_numpy(g.uniform(0.5, 1.0, K).astype(np.float32)),
                "cos_sims": torch.from_numpy(g.uniform(0.3, 1.0, K).astype(np.float32)),
            }
    os.makedirs(os.path.join(dst, "dino_best_buddies"), exist_ok=True)
    torch.save(bb, os.path.join(dst, "dino_best_buddies", "dino_best_buddies_filtered.pt"))
    if ref_root:
        with open(os.path.join(ref_root, "config", "train.yaml")) as fh:
            conf = yaml.safe_load(fh.read())
    else:
        conf = dict(TRAIN_YAML)
    conf.update(video_resh=H, video_resw=W, total_iterations=cfg["total_iterations"], checkpoint_interval=100000,
                apply_cyc_after=0, apply_cl_ref_after=0)
    if overrides is None:
        overrides = dict(train_batch_size=48, batch_n_frames=3, cyc_n_frames=2, cyc_batch_size_per_frame=24, cl_n_frames=2,
                         cl_points_per_pair=24, checkpoint_interval=1000,
                         # larger weights than config/train.yaml so that every term moves the total visibly in three steps
                         lambda_cl_dino_bb=0.01, lambda_cl_ref_bb=0.01, lambda_emb_norm=0.01, lambda_angle=0.01)
    conf.update(overrides)
    path = os.path.join(dst, "train.yaml")
    with open(path, "w") as fh:
        yaml.safe_dump(conf, fh)
    return dst, path
