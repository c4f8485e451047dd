"""Top-level `utils` of the reference (utils.py): path layout and the per-video DINO feature driver."""
from __future__ import annotations

import os

import torch

from . import ops
from .extractor import VitExtractor


def add_config_paths(data_path, config):
    """utils.py:10-29 (unchanged contract: the on-disk layout the reference's scripts read and write)."""
    j = os.path.join
    config["video_folder"] = j(data_path, "video")
    config["trajectories_file"] = j(data_path, "of_trajectories", "trajectories.pt")
    config["unfiltered_trajectories_file"] = j(data_path, "of_trajectories", "trajectories_wo_direct_filter.pt")
    config["fg_trajectories_file"] = j(data_path, "of_trajectories", "fg_trajectories.pt")
    config["bg_trajectories_file"] = j(data_path, "of_trajectories", "bg_trajectories.pt")
    config["dino_embed_video_path"] = j(data_path, "dino_embeddings", "dino_embed_video.pt")
    config["dino_bb_dir"] = j(data_path, "dino_best_buddies")
    config["mask_dino_embed_video_path"] = j(data_path, "dino_embeddings", "dino_embed_video-layer=23.pt")
    config["masks_path"] = j(data_path, "masks")
    config["ckpt_folder"] = j(data_path, "models", "dino_tracker")
    config["trajectories_dir"] = j(data_path, "trajectories")
    config["occlusions_dir"] = j(data_path, "occlusions")
    config["grid_trajectories_dir"] = j(data_path, "grid_trajectories")
    config["grid_occlusions_dir"] = j(data_path, "grid_occlusions")
    config["model_vis_dir"] = j(data_path, "visualizations")
    return config


_FACET_ROW = {"queries": 0, "keys": 1, "values": 2}


@torch.no_grad()
def get_dino_features_video_packed(video, model_name="dinov2_vitb14", facet="tokens", stride=7, layer=None,
                                   device: str = "cuda:0", extractor: VitExtractor = None, **extractor_kwargs):
    """Device-resident variant: T x (ph*pw) x C token-major fp32 (what Tracker consumes), no D2H per frame."""
    ex = extractor if extractor is not None else VitExtractor(model_name=model_name, stride=stride, device=device,
                                                              **extractor_kwargs)
    if facet == "tokens":
        return ex.encode(video, layer=layer, normalize=True, want="feat")
    if facet not in _FACET_ROW:
        raise ValueError(f"facet {facet} not supported")  # utils.py:63
    d = ex.cfg["dim"]
    out = []
    for i in range(video.shape[0]):  # one frame at a time: the qkv record is 3x the token volume
        qkv = ex.encode(video[i:i + 1], layer=layer, normalize=True, want="qkv")
        out.append(qkv[0, 1:, _FACET_ROW[facet] * d:(_FACET_ROW[facet] + 1) * d])
    return torch.stack(out).contiguous()


@torch.no_grad()
def get_dino_features_video(video, model_name="dinov2_vitb14", facet="tokens", stride=7, layer=None,
                            device: str = "cuda:0", **extractor_kwargs):
    """utils.py:33-72: T x 3 x H x W frames in [0,1] -> T x C x ph x pw features on the CPU."""
    ex = extractor_kwargs.pop("extractor", None) or VitExtractor(model_name=model_name, stride=stride, device=device,
                                                                 **extractor_kwargs)
    feat = get_dino_features_video_packed(video, model_name, facet, stride, layer, device, extractor=ex)
    ph, pw = ex.get_height_patch_num(video[[0]].shape), ex.get_width_patch_num(video[[0]].shape)
    return ops.unpack_features(feat, ph, pw).cpu()


def bilinear_interpolate_video(video: torch.Tensor, points: torch.Tensor, h: int, w: int, t: int, normalize_h=False,
                               normalize_w=False, normalize_t=True):
    """utils.py:75-101 on the device (dtk_sample_grid): video 1 x C x T x H' x W', points B x 3 (x, y, t); the same
    normalisation switches; trilinear, border, align_corners -> 1 x C x 1 x B x 1."""
    samples = points.detach().clone().to(torch.float32)
    if normalize_w:
        samples[:, 0] = samples[:, 0] / (w - 1) * 2 - 1
    if normalize_h:
        samples[:, 1] = samples[:, 1] / (h - 1) * 2 - 1
    if normalize_t:
        if t > 1:
            samples[:, 2] = samples[:, 2] / (t - 1)
        samples[:, 2] = samples[:, 2] * 2 - 1
    _, C, T, hh, ww = video.shape
    feat, _ = ops.pack_features(video[0].permute(1, 0, 2, 3).to(torch.float32).contiguous())  # T C H' W' -> token-major
    out = ops.sample_grid(feat, hh, ww, samples.to(feat.device).contiguous())
    return out.t()[None, :, None, :, None]


def save_dino_embed_video(path: str, features: torch.Tensor, dtype: torch.dtype = torch.float32) -> None:
    """Write `dino_embeddings/dino_embed_video.pt` (utils.py:18; preprocessing/save_dino_embed_video.py:26): T x C x h x w.
    The reference stores fp32 (1.1 GB at T = 90, C = 384; 3 GB at C = 1024); `dtype=torch.bfloat16` halves the file -- the
    ViT produced the values from bf16 operands anyway -- and `Tracker.load_dino_embed_video` widens whatever it finds to the
    fp32 token-major volume.  `features` may also be the token-major [T, h*w, C] device volume with (h, w) given as a tuple
    in place of a tensor layout: pass `ops.unpack_features(feat, h, w)` for that."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(features.detach().to("cpu", dtype).contiguous(), path)
