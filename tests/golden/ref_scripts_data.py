"""TEST INFRASTRUCTURE -- builds the on-disk layout the reference's scripts read (utils.py:10-29) for configs 1 and 3:

    <data>/video/*.jpg  <data>/masks/*.png  <data>/dino_embeddings/dino_embed_video.pt
    <data>/models/dino_tracker/{tracker_head,delta_dino}_100.pt      and a TAP-Vid-schema pickle (data/tapvid.py:5-41)

Frames and masks are the first T of the reference's own dataset/horsejump (so the reference checkout must be present:
the build container, or a scratch copy next to a GPU run); embeddings and weights are seeded (dino_tracker_amd.synth).
The un-modified reference hard-codes a 1024-wide DeltaDINO (models/networks/delta_dino.py:9), so C = 1024 here -- which
also is the width of its shipped configuration (ViT-L, config/preprocessing.yaml:10-13).
"""
import os
import pickle
import shutil

import numpy as np
import torch

from dino_tracker_amd import synth

CFG1 = dict(T=16, C=1024, interval=80, feat_seed=21, head_seed=3, delta_seed=8)   # config 1: horsejump 16 frames, grid queries
CFG3 = dict(T=12, C=1024, feat_seed=22, head_seed=3, delta_seed=8, video_idx=0,
            query_frames=(0, 5, 10), per_frame=6)                                     # configs 3-4: TAP-Vid-schema pickle


def build_data_dir(dst, ref_root, cfg):
    T, C = cfg["T"], cfg["C"]
    src = os.path.join(ref_root, "dataset", "horsejump")
    for sub in ("video", "masks"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        files = sorted(os.listdir(os.path.join(src, sub)))[:T]
        assert len(files) == T, (sub, len(files))
        for f in files:
            shutil.copy(os.path.join(src, sub, f), os.path.join(dst, sub, f))
    os.makedirs(os.path.join(dst, "dino_embeddings"), exist_ok=True)
    torch.save(synth.synth_features(T, C, 67, 121, seed=cfg["feat_seed"]), os.path.join(dst, "dino_embeddings", "dino_embed_video.pt"))
    ck = os.path.join(dst, "models", "dino_tracker")
    os.makedirs(ck, exist_ok=True)
    torch.save(synth.synth_head_weights(cfg["head_seed"]), os.path.join(ck, "tracker_head_100.pt"))
    torch.save(synth.synth_delta_dino_weights(C, cfg["delta_seed"]), os.path.join(ck, "delta_dino_100.pt"))
    return dst


def build_tapvid_pickle(path, cfg):
    """{"videos": [{"video_idx", "h", "w", "query_points": {f: [[x, y], ...]}, "target_points": {f: [N, T, 2]},
    "occluded": {f: [N, T]}}]} in a 256 x 256 raster like TAP-Vid-DAVIS (eval/metrics.py:168-202).  Ground truth = the
    motion of the synthetic feature field (0.6, 0.3) cells = (4.2, 2.1) px per frame at model resolution, with points that
    leave the frame marked occluded and annotation noise of mixed scale (0.3 .. 12 px) on the ground truth -- so that
    every one of the 13 metrics is a non-trivial number that moves with the threshold."""
    g = np.random.default_rng(5)
    T = cfg["T"]
    hh = ww = 256
    sx, sy = ww / 854.0, hh / 476.0
    qp, tp, oc = {}, {}, {}
    for f in cfg["query_frames"]:
        n = cfg["per_frame"]
        xy = np.stack([g.uniform(40, 216, n), g.uniform(40, 216, n)], axis=1)
        dt = (np.arange(T) - f)[None, :, None]
        tracks = xy[:, None, :] - dt * np.array([4.2 * sx, 2.1 * sy])[None, None, :]  # content moves by -v per frame
        noise = g.normal(size=tracks.shape) * g.choice([0.3, 0.8, 2.0, 5.0, 12.0], size=(n, T, 1))
        noise[:, f] = 0.0
        tracks = tracks + noise
        occ = (tracks[..., 0] < 0) | (tracks[..., 0] > ww - 1) | (tracks[..., 1] < 0) | (tracks[..., 1] > hh - 1)
        occ |= g.uniform(size=occ.shape) < 0.1
        occ[:, f] = False
        qp[f], tp[f], oc[f] = xy.tolist(), tracks.astype(np.float32), occ
    data = {"videos": [{"video_idx": cfg["video_idx"], "h": hh, "w": ww, "query_points": qp, "target_points": tp,
                        "occluded": oc}]}
    with open(path, "wb") as fh:
        pickle.dump(data, fh)
    return data


# ---- the same two configurations on a SYNTHETIC video (no reference checkout needed to rebuild the inputs) ----------------
# Frames are dino_tracker_amd.synth.synth_video quantised to 8 bits and written as 854 x 476 PNGs (lossless, already at the
# model resolution: the reference's LANCZOS resize to the same size returns the pixels unchanged), masks a centred rectangle.
CFG1S = dict(T=8, C=1024, interval=80, feat_seed=23, head_seed=3, delta_seed=8, video_seed=310)
CFG3S = dict(T=8, C=1024, feat_seed=24, head_seed=3, delta_seed=8, video_seed=311, video_idx=0, query_frames=(0, 3, 6),
             per_frame=6)


def build_synth_data_dir(dst, cfg):
    from PIL import Image
    T, C = cfg["T"], cfg["C"]
    video = (synth.synth_video(T, 476, 854, seed=cfg["video_seed"]) * 255.0).round().clamp(0, 255).to(torch.uint8)
    for sub in ("video", "masks"):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
    mask = np.zeros((476, 854), dtype=np.uint8)
    mask[120:360, 250:600] = 255
    for t in range(T):
        Image.fromarray(video[t].permute(1, 2, 0).numpy()).save(os.path.join(dst, "video", f"{t:05d}.png"))
        Image.fromarray(mask).save(os.path.join(dst, "masks", f"{t:05d}.png"))
    os.makedirs(os.path.join(dst, "dino_embeddings"), exist_ok=True)
    torch.save(synth.synth_features(T, C, 67, 121, seed=cfg["feat_seed"]), os.path.join(dst, "dino_embeddings", "dino_embed_video.pt"))
    ck = os.path.join(dst, "models", "dino_tracker")
    os.makedirs(ck, exist_ok=True)
    torch.save(synth.synth_head_weights(cfg["head_seed"]), os.path.join(ck, "tracker_head_100.pt"))
    torch.save(synth.synth_delta_dino_weights(C, cfg["delta_seed"]), os.path.join(ck, "delta_dino_100.pt"))
    return dst
