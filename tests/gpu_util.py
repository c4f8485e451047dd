"""Helpers for the -m gpu parity tests: build a HIP-backed Tracker/ModelInference from synthetic inputs."""
import os
import tempfile

import torch

from dino_tracker_amd import ops
from dino_tracker_amd.dataset import RangeNormalizer
from dino_tracker_amd.model_inference import ModelInference
from dino_tracker_amd.tracker import Tracker

DEV = "cuda:0"


def make_tracker(video, dino, head, delta=None, method=ops.TRACK_EXACT, cache=True, p2_operands="split"):
    """p2_operands: Delta-DINO convolution operands -- "split" (fp32-grade; what the 3e-5 feature comparisons need) or "fp16"
    (the library default) or None (whatever the default / $DTK_P2_OPERANDS says)."""
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "dino_embed_video.pt")
    torch.save(dino, path)
    trk = Tracker(video=video.to(DEV), ckpt_path=tmp, dino_embed_path=path, dino_patch_size=14, stride=7, device=DEV,
                  track_method=method)
    trk.tracker_head.load_state_dict(head)
    trk.tracker_head.to(DEV)
    if delta is not None:
        trk.delta_dino.load_state_dict(delta)
        trk.delta_dino.to(DEV)
        if p2_operands is not None:
            trk.delta_dino.conv_operands = p2_operands
    elif cache:
        trk.refined_features = dino.to(DEV)  # zero-initialised Delta-DINO == identity (delta_dino.py:33-35)
    return trk


def make_inference(trk, H, W, T, anchor_th=0.7, cos_th=0.6):
    rn = RangeNormalizer(shapes=(W, H, T), device=DEV)
    return ModelInference(trk, rn, anchor_cosine_similarity_threshold=anchor_th, cosine_similarity_threshold=cos_th)
