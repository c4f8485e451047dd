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


@torch.no_grad()
def get_dino_features_video_packed(video, model_name="dinov2_vitb14", facet="tokens", stride=7, layer=None,
                                   device: str = "cuda:0", extractor: VitExtractor = None, **extractor_kwargs):
    """Device-resident variant: T x (ph*pw) x C token-major fp32 (what Tracker consumes), no D2H per frame."""
    if facet != "tokens":
        raise NotImplementedError(f"facet {facet!r}: only 'tokens' runs on the HIP encoder")
    ex = extractor if extractor is not None else VitExtractor(model_name=model_name, stride=stride, device=device,
                                                              **extractor_kwargs)
    return ex.encode(video, layer=layer, normalize=True, want="feat")


@torch.no_grad()
def get_dino_features_video(video, model_name="dinov2_vitb14", facet="tokens", stride=7, layer=None,
                            device: str = "cuda:0", **extractor_kwargs):
    """utils.py:33-72: T x 3 x H x W frames in [0,1] -> T x C x ph x pw features on the CPU."""
    feat = get_dino_features_video_packed(video, model_name, facet, stride, layer, device, **extractor_kwargs)
    patch = 14
    ph, pw = 1 + (video.shape[-2] - patch) // stride, 1 + (video.shape[-1] - patch) // stride
    return ops.unpack_features(feat, ph, pw).cpu()


def bilinear_interpolate_video(video: torch.Tensor, points: torch.Tensor, h: int, w: int, t: int, normalize_h=False,
                               normalize_w=False, normalize_t=True):
    """utils.py:75-101 signature; integral frame indices, runs dtk_sample_points.  video: 1 x C x T x H' x W',
    points B x 3 (x, y in [-1,1] of the token grid, t) -> 1 x C x 1 x B x 1."""
    if normalize_h or normalize_w:
        raise NotImplementedError("pixel-normalised sampling is not used by the tracker path")
    from ._lib import make_geom
    emb = video[0].permute(1, 0, 2, 3).contiguous()  # T C H W
    feat, _ = ops.pack_features(emb)
    T, C, hh, ww = emb.shape
    g = make_geom(T, C, 14 + 7 * (hh - 1), 14 + 7 * (ww - 1))
    xy = torch.stack([(points[:, 0] + 1) / 2 * (ww - 1) * 7 + 7, (points[:, 1] + 1) / 2 * (hh - 1) * 7 + 7], 1).contiguous()
    out = ops.sample_points(g, feat, xy.float(), points[:, 2].round().to(torch.int32).contiguous())
    return out.t()[None, :, None, :, None]
