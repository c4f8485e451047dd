"""Tracker -- the reference's models/tracker.py:17-325 API on top of the HIP device programs.

Differences in mechanism (not in results):
  * features live token-major on the device ([T][HW][C] fp32 + per-cell norms, computed once per video) instead of
    being gathered per call (tracker.py:316-317) and re-normalised per call (tracker.py:162);
  * one correlation map per (source, target frame) is computed instead of B x n maps of which B are kept
    (tracker.py:159-160);
  * DeltaDINO's width follows the embedding file (the reference hard-codes 1024, delta_dino.py:9).
"""
from __future__ import annotations

import gc
import os
from pathlib import Path
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import TrackStats, make_geom
from .dataset import RangeNormalizer
from .networks import DeltaDINO, TrackerHead


EPS = 1e-08  # models/tracker.py:14


def load_pre_trained_model(pre_trained_sd, target_model):
    """models/utils.py:71-76."""
    target_model.load_state_dict(dict(pre_trained_sd))
    return target_model


class Tracker(nn.Module):
    def __init__(self, video=None, ckpt_path="", dino_embed_path="", dino_patch_size=14, stride=7, device="cuda:0",
                 cyc_n_frames=4, cyc_batch_size_per_frame=256, cyc_fg_points_ratio=0.7, cyc_thresh=4,
                 track_method: Optional[int] = None, dino_features: Optional[torch.Tensor] = None):
        """Reference signature (models/tracker.py:18-30) plus two optional extensions: `track_method`
        (ops.TRACK_MFMA default / ops.TRACK_EXACT) and `dino_features`, a device-resident token-major
        [T, ph*pw, C] volume handed over by the extractor instead of the dino_embed_video.pt round trip."""
        super().__init__()
        self.stride = stride
        self.dino_patch_size = dino_patch_size
        self.device = device
        self.dino_embed_path = dino_embed_path
        self.ckpt_path = ckpt_path
        self.cyc_n_frames, self.cyc_batch_size_per_frame = cyc_n_frames, cyc_batch_size_per_frame
        self.cyc_fg_points_ratio, self.cyc_thresh = cyc_fg_points_ratio, cyc_thresh
        self.video = video
        self.track_method = ops.TRACK_MFMA if track_method is None else track_method
        self._dino_features_arg = dino_features

        self._dino = None          # token-major raw embeddings  [T][HW][C]
        self._dino_norms = None
        self._refined = None       # token-major refined features
        self._refined_norms = None
        self._refined_f16 = None
        self._dino_f16 = None
        self._refined_chw = None   # lazily unpacked view for callers reading `.refined_features`
        self._refined_key = None   # Delta-DINO parameter versions the cached refined volume was computed with
        self._workspace = None
        self.track_round_sources = 0          # dtk_track_opts.round_sources (0 = library default)
        self.cyc_sampling = os.environ.get("DTK_CYC_SAMPLING", "device")  # see _cycle_point_sets
        self.track_tier = ops.TIER_AUTO       # dtk_track_opts.tier
        self.last_track_stats = None          # dtk_track_stats of the most recent dtk_track call (dict)

        self._dino_chw = None
        self.load_dino_embed_video()
        t, c, h, w = self.video.shape
        emb_c = self._dino.shape[2]
        self.delta_dino = DeltaDINO(channels=[3, 64, 128, 256, emb_c], vit_stride=self.stride).to(device)
        self.tracker_head = TrackerHead(use_cnn_refiner=True, patch_size=dino_patch_size, step_h=stride, step_w=stride,
                                        video_h=h, video_w=w).to(device)
        self.range_normalizer = RangeNormalizer(shapes=(w, h, t), device=device)
        self.geom = make_geom(t, emb_c, h, w, dino_patch_size, stride, float(self.tracker_head.argmax_radius))
        if self._dino.shape[1] != self.geom.ph * self.geom.pw or self._dino.shape[0] != t:
            raise RuntimeError(f"dino embeddings {tuple(self._dino.shape)} (T, tokens, C) do not match video "
                               f"{tuple(self.video.shape)} at patch {dino_patch_size} stride {stride}")
        if emb_c % 32 != 0 and self.track_method == ops.TRACK_MFMA:
            self.track_method = ops.TRACK_EXACT  # the MFMA path needs C % 32 == 0

    # ---- embeddings ------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_dino_embed_video(self):
        """tracker.py:64-71: the T x C x h x w tensor written by preprocessing/save_dino_embed_video.py, kept
        token-major on the device; or the in-memory volume passed as `dino_features`."""
        if self._dino_features_arg is not None:
            self._dino = self._dino_features_arg.to(self.device, torch.float32).contiguous()
            self._dino_norms = None      # on first use (_packed_dino): only tracking on the RAW features reads them
            self._dino_features_arg = None
            return
        assert os.path.exists(self.dino_embed_path)
        chw = torch.load(self.dino_embed_path, map_location=self.device).to(torch.float32).contiguous()
        self._dino, self._dino_norms = ops.pack_features(chw)

    @torch.no_grad()
    def set_video(self, video: torch.Tensor, dino_features: torch.Tensor):
        """Re-target this tracker (weights kept) at another video of the same geometry: `video` [T,3,H,W] and its
        token-major DINO volume [T, ph*pw, C] straight from the extractor.  Drops every cached derived tensor."""
        if tuple(video.shape[1:]) != tuple(self.video.shape[1:]) or video.shape[0] != self.geom.T:
            raise RuntimeError("set_video: geometry differs; build a new Tracker")
        if tuple(dino_features.shape) != (self.geom.T, self.geom.ph * self.geom.pw, self.geom.C):
            raise RuntimeError(f"set_video: dino_features {tuple(dino_features.shape)} != (T, tokens, C)")
        self.video = video
        self._dino = dino_features.to(self.device, torch.float32).contiguous()
        self._dino_norms = None          # on first use (_packed_dino)
        self._dino_chw = self._dino_f16 = None
        self.refined_features = None

    @property
    def dino_embed_video(self):
        """T x C x h x w view for callers of the reference attribute (unpacked on first use)."""
        if self._dino_chw is None:
            self._dino_chw = ops.unpack_features(self._dino, self.geom.ph, self.geom.pw)
        return self._dino_chw

    @dino_embed_video.setter
    def dino_embed_video(self, emb):
        emb = emb.to(self.device, torch.float32).contiguous()
        self._dino, self._dino_norms = ops.pack_features(emb)
        self._dino_chw = None
        self._dino_f16 = None

    def _packed_dino(self):
        if self._dino_norms is None:     # the raw volume's norms are needed by use_raw_features tracking only: made on demand
            self._dino_norms = ops.feature_norms(self._dino)
        return self._dino, self._dino_norms

    def get_dino_embed_video(self, frames_set_t):
        return self.dino_embed_video[frames_set_t.to(self.device).long()]

    @property
    def refined_features(self):
        """T x C x h x w like the reference's attribute (None until cache_refined_embeddings())."""
        if self._refined is None:
            return None
        if self._refined_chw is None:
            self._refined_chw = ops.unpack_features(self._refined, self.geom.ph, self.geom.pw)
        return self._refined_chw

    @refined_features.setter
    def refined_features(self, value):
        self._refined_key = None  # an externally supplied volume is never invalidated by weight changes
        if value is None:
            self._refined = self._refined_norms = self._refined_f16 = self._refined_chw = None
        else:  # accept a T x C x h x w tensor (e.g. features refined elsewhere)
            self._refined, self._refined_norms = ops.pack_features(value.to(self.device, torch.float32).contiguous())
            self._refined_f16 = None
            self._refined_chw = None

    def normalize_points_for_sampling(self, points):
        """tracker.py:77-94 (kept for API parity; the kernels take pixel coordinates directly)."""
        t, c, h, w = self.video.shape
        half = self.dino_patch_size / 2
        last_h = ((h - self.dino_patch_size) // self.stride) * self.stride + half
        last_w = ((w - self.dino_patch_size) // self.stride) * self.stride + half
        key = (str(points.device), points.dtype)
        if getattr(self, "_norm_ab", None) is None or self._norm_ab[0] != key:  # two constants: built once, not per call
            a = torch.tensor([[2 / (last_w - half), 2 / (last_h - half), 1]], device=points.device, dtype=points.dtype)
            b = torch.tensor([[1 - last_w * 2 / (last_w - half), 1 - last_h * 2 / (last_h - half), 0]],
                             device=points.device, dtype=points.dtype)
            self._norm_ab = (key, a, b)
        _, a, b = self._norm_ab
        return a * points + b

    def sample_embeddings(self, embeddings, source_points):
        """tracker.py:96-111 -> utils.bilinear_interpolate_video: embeddings T' x C x h x w, source_points B x 3 with
        x, y in [-1, 1] (token-grid coordinates, see normalize_points_for_sampling) and t an index into `embeddings`
        (normalised by T' - 1 like the reference, utils.py:97-100) -> B x C."""
        t, c, h, w = embeddings.shape
        if torch.is_grad_enabled() and embeddings.requires_grad:  # training: gradients flow into the embeddings
            from . import train_ops
            return train_ops.sample_bilinear(embeddings, source_points.to(embeddings.device, torch.float32))
        if self._refined is not None and embeddings is self._refined_chw:
            feat = self._refined
        elif self._dino is not None and embeddings is self._dino_chw:
            feat = self._dino
        else:
            feat, _ = ops.pack_features(embeddings.to(self.device, torch.float32).contiguous())
        pts = source_points.to(self.device, torch.float32).clone()
        if t > 1:
            pts[:, 2] = pts[:, 2] / (t - 1)
        pts[:, 2] = pts[:, 2] * 2 - 1
        return ops.sample_grid(feat, h, w, pts.contiguous())

    # ---- Delta-DINO --------------------------------------------------------------------------------------
    def get_refined_embeddings(self, frames_set_t, return_raw_embeddings=False):
        """tracker.py:113-129 for an arbitrary frame subset (T' x C x h x w tensors, reference layout)."""
        from .delta_dino import refine_frames
        idx = frames_set_t.to(self.device).long()
        dino = self.dino_embed_video[idx].contiguous()
        if self.delta_dino.training:
            # test-time training: train-mode BatchNorm over batches of 8 frames (tracker.py:118-124), autograd graph
            # into the Delta-DINO parameters
            residual = torch.cat([self.delta_dino(self.video[idx[i:i + 8]], dino[i:i + 8]) for i in range(0, idx.numel(), 8)])
            refined = dino + residual
        else:
            refined = refine_frames(self.delta_dino, self.video[idx].contiguous(), dino, self.geom)
            residual = refined - dino
        if return_raw_embeddings:
            return refined, residual, dino
        return refined, residual

    @torch.no_grad()
    def cache_refined_embeddings(self, move_dino_to_cpu=False):
        """tracker.py:131-135: refined = dino + DeltaDINO(video) for all frames, kept token-major on the device."""
        from .delta_dino import refine_video_packed
        # (the norms of the refined volume come out of the kernel that writes it: no second pass over 1.1 GB)
        self._refined, self._refined_norms = refine_video_packed(self.delta_dino, self.video, self._dino, self.geom, want_norms=True)
        self._refined_f16 = None
        self._refined_chw = None
        self._refined_key = self._delta_key()
        # move_dino_to_cpu was a memory knob of the reference (two fp32 copies of the volume); the raw volume stays
        # on the device here because Tracker.forward(use_raw_features=True) reads it in place.

    @torch.no_grad()
    def set_refined_packed(self, refined_thwc: torch.Tensor):
        """Install a refined volume that is already token-major [T, ph*pw, C] on the device (e.g. all-gathered from the
        ranks that refined a slice of the frames each, sharding.query_parallel)."""
        if tuple(refined_thwc.shape) != (self.geom.T, self.geom.ph * self.geom.pw, self.geom.C):
            raise RuntimeError(f"set_refined_packed: {tuple(refined_thwc.shape)} != (T, tokens, C)")
        self._refined = refined_thwc.to(self.device, torch.float32).contiguous()
        self._refined_norms = ops.feature_norms(self._refined)
        self._refined_f16 = self._refined_chw = None
        self._refined_key = self._delta_key()

    def _delta_key(self):
        from .delta_dino import weights_key
        return weights_key(self.delta_dino, self.device)

    def refined_is_stale(self) -> bool:
        """True when the cached refined volume was computed by cache_refined_embeddings() with Delta-DINO parameters
        that have changed since (load_weights / load_state_dict / an optimiser step)."""
        return self._refined is not None and self._refined_key is not None and self._refined_key != self._delta_key()

    def uncache_refined_embeddings(self, move_dino_to_gpu=False):
        self.refined_features = None
        torch.cuda.empty_cache()
        gc.collect()

    # ---- checkpoints (tracker.py:144-156) ------------------------------------------------------------------
    def save_weights(self, iter):
        torch.save(self.tracker_head.state_dict(), Path(self.ckpt_path) / f"tracker_head_{iter}.pt")
        torch.save(self.delta_dino.state_dict(), Path(self.ckpt_path) / f"delta_dino_{iter}.pt")

    def load_weights(self, iter):
        self.tracker_head = load_pre_trained_model(
            torch.load(os.path.join(self.ckpt_path, f"tracker_head_{iter}.pt"), map_location=self.device), self.tracker_head)
        self.delta_dino = load_pre_trained_model(
            torch.load(os.path.join(self.ckpt_path, f"delta_dino_{iter}.pt"), map_location=self.device), self.delta_dino)
        if self._refined_key is not None:
            self.refined_features = None  # refined with the previous Delta-DINO weights: never reuse it

    # ---- device state used by ModelInference -----------------------------------------------------------------
    def features(self, use_raw_features=False):
        """(feat [T][HW][C], norms [T][HW], fp16 copy or None) of the volume the tracker correlates against."""
        if not use_raw_features and self.refined_is_stale():
            self.cache_refined_embeddings()
        if use_raw_features or self._refined is None:
            if not use_raw_features and self._refined is None:
                raise RuntimeError("refined features are not cached: call cache_refined_embeddings() first")
            feat, norms = self._packed_dino()
            if self.track_method == ops.TRACK_MFMA and self._dino_f16 is None:
                self._dino_f16 = ops.make_feat_f16(self.geom, feat, norms)
            return feat, norms, self._dino_f16
        if self.track_method == ops.TRACK_MFMA and self._refined_f16 is None:
            self._refined_f16 = ops.make_feat_f16(self.geom, self._refined, self._refined_norms)
        return self._refined, self._refined_norms, self._refined_f16

    def workspace(self, M: int) -> torch.Tensor:
        need = ops.track_workspace_bytes(self.geom, M, self.track_method, self.track_round_sources)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._workspace

    def track_sources(self, feats, emb, src_row, tgt, out_idx, out_xy, M, dM=None, normalized=False):
        feat, norms, f16 = feats
        head = self.tracker_head.packed_params(feat.device)
        stats = TrackStats()
        ops.track(self.geom, feat, norms, f16, head, emb, src_row, tgt, out_idx, out_xy, M, self.workspace(M), dM=dM,
                  normalized=normalized, method=self.track_method, round_sources=self.track_round_sources,
                  tier=self.track_tier, stats=stats)
        self.last_track_stats = stats.as_dict()
        return out_xy

    # ---- the reference's per-call building blocks (tracker.py:158-180), on the device --------------------------
    def cmap_relu(self, x):
        return torch.relu(x)

    def get_corr_maps_for_frame_set(self, source_embeddings, frame_embeddings_set, target_frame_indices):
        """tracker.py:158-169: cosine maps of source b against frame_embeddings_set[target_frame_indices[b]] -> B x 1 x h x w.
        (One map per source: the reference computes B x n maps and keeps the diagonal.)"""
        n, c, h, w = frame_embeddings_set.shape
        if self._refined is not None and frame_embeddings_set is self._refined_chw:
            feat, norms = self._refined, self._refined_norms
        else:
            feat, norms = ops.pack_features(frame_embeddings_set.to(self.device, torch.float32).contiguous())
        g = make_geom(n, c, self.geom.video_h, self.geom.video_w, self.dino_patch_size, self.stride, self.geom.radius)
        if (g.ph, g.pw) != (h, w):
            raise RuntimeError(f"frame embeddings {h}x{w} do not match the {g.ph}x{g.pw} token grid of the video")
        emb = source_embeddings.to(self.device, torch.float32).contiguous()
        tgt = target_frame_indices.to(self.device).to(torch.int32).contiguous()
        return ops.corr_maps(g, feat, norms, emb, tgt)[:, None]

    def _differentiable(self):
        """Training step with autograd on: the torch statement of the path (train_ops) instead of the inference kernels."""
        return self.training and torch.is_grad_enabled()

    def get_point_predictions_from_embeddings(self, source_embeddings, frame_embeddings_set, target_frame_indices):
        if self._differentiable():
            from . import train_ops
            return train_ops.track_points(self.tracker_head, source_embeddings, frame_embeddings_set,
                                          target_frame_indices.to(frame_embeddings_set.device))
        corr_maps = self.get_corr_maps_for_frame_set(source_embeddings, frame_embeddings_set, target_frame_indices)
        return self.tracker_head(self.cmap_relu(corr_maps))

    def get_point_predictions(self, inp, frame_embeddings):
        source_points_unnormalized, source_frame_indices, target_frame_indices, _ = inp
        source_points = self.normalize_points_for_sampling(source_points_unnormalized.to(self.device, torch.float32))
        pts = torch.cat([source_points[:, :-1], source_frame_indices.to(self.device)[:, None].to(torch.float32)], dim=1)
        source_embeddings = self.sample_embeddings(frame_embeddings, pts)
        return self.get_point_predictions_from_embeddings(source_embeddings, frame_embeddings, target_frame_indices)

    # ---- test-time training ---------------------------------------------------------------
    def train(self, mode: bool = True):
        """nn.Module.train; leaving training mode also releases the convolution scratch of the training step (10 GB at
        full size) and the per-batch tensors the loss terms read."""
        super().train(mode)
        if not mode:
            from . import train_ops
            train_ops.release_scratch()
            self.frame_embeddings = self.raw_embeddings = self.residual_embeddings = None
        return self


    def _forward_train(self, inp, use_raw_features=False):
        """tracker.py:303-325 in training mode: refine the batch's frames with gradients, keep the tensors the loss
        terms of dino_tracker.py read (`frame_embeddings`, `raw_embeddings`, `residual_embeddings`)."""
        frames_set_t = inp[-1]
        from . import train_ops as _to
        _to.new_iteration()   # (what the tracker passes of one iteration share starts here)
        if use_raw_features:
            frame_embeddings = raw_embeddings = self.get_dino_embed_video(frames_set_t)
        elif self._refined is not None:  # a cached volume takes precedence, as in the reference (tracker.py:315-317)
            idx = frames_set_t.to(self.device).long()
            frame_embeddings, raw_embeddings = self.refined_features[idx], self.dino_embed_video[idx]
        else:
            frame_embeddings, residual_embeddings, raw_embeddings = self.get_refined_embeddings(
                frames_set_t, return_raw_embeddings=True)
            self.residual_embeddings = residual_embeddings
        if torch.is_grad_enabled() and frame_embeddings.requires_grad:
            from . import train_ops
            frame_embeddings = train_ops.attach_grad_sink(frame_embeddings)  # one gradient buffer for the point / window reads
        self.frame_embeddings = frame_embeddings
        self.raw_embeddings = raw_embeddings
        return self.get_point_predictions(inp, frame_embeddings)

    def _cycle_point_sets(self, frames_set_t, fg_masks):
        """Index plumbing of tracker.py:183-222: cyc_n_frames random (source, target) frame pairs of the batch and, per
        pair, cyc_batch_size_per_frame pixel positions of the source frame split foreground / background by the mask.

        `self.cyc_sampling` (default from $DTK_CYC_SAMPLING, "device"):
          "device"     everything on the device, no host read: per pair one uniform key per pixel, the n_fg largest keys among
                       the foreground pixels and the n_bg largest among the background ones = a uniformly random subset
                       without replacement, which is what `coords[randperm(len)[:k]]` is (tracker.py:215-217);
          "reference"  the reference's own sequence of draws (2 x randint, then a HOST randperm over the foreground and one
                       over the background pixels of each pair): bit-identical index sets for identical seeds (the parity
                       tests run this), at the price of 8 host permutations of up to 4e5 elements and ~20 device
                       synchronisations per iteration."""
        n = frames_set_t.shape[0]
        dev = frames_set_t.device
        source_selector = torch.randint(n, (self.cyc_n_frames,), device=dev)
        target_selector = torch.randint(n, (self.cyc_n_frames,), device=dev)
        h, w = fg_masks.shape[-2:]
        n_fg = int(self.cyc_batch_size_per_frame * self.cyc_fg_points_ratio)
        n_bg = self.cyc_batch_size_per_frame - n_fg
        if self.cyc_sampling == "device":
            pts, src_idx, tgt_idx, valid = self._cycle_point_sets_static(frames_set_t, fg_masks, source_selector, target_selector)
            keep = valid.nonzero()[:, 0]                                                  # the one host read of this function
            return pts[keep], src_idx[keep], tgt_idx[keep]
        pts, src_idx, tgt_idx = [], [], []
        for s_i, t_i in zip(source_selector.tolist(), target_selector.tolist()):
            source_t = int(frames_set_t[s_i])
            fg = (fg_masks[source_t] > 0).reshape(-1)
            cells_fg = fg.nonzero()[:, 0]
            cells_bg = (~fg).nonzero()[:, 0]
            cells_fg = cells_fg[torch.randperm(cells_fg.shape[0])[:n_fg].to(cells_fg.device)]
            cells_bg = cells_bg[torch.randperm(cells_bg.shape[0])[:n_bg].to(cells_bg.device)]
            cells = torch.cat([cells_fg, cells_bg])
            xy = torch.stack([(cells % w).float(), torch.div(cells, w, rounding_mode="floor").float(),
                              torch.full_like(cells, source_t, dtype=torch.float32)], dim=1)
            pts.append(xy.to(dev))
            src_idx.append(torch.full((xy.shape[0],), s_i, device=dev, dtype=torch.long))
            tgt_idx.append(torch.full((xy.shape[0],), t_i, device=dev, dtype=torch.long))
        return torch.cat(pts), torch.cat(src_idx), torch.cat(tgt_idx)

    def _cycle_point_sets_static(self, frames_set_t, fg_masks, source_selector=None, target_selector=None):
        """The "device" sampling of _cycle_point_sets with static shapes: cyc_n_frames x cyc_batch_size_per_frame points and
        a flag per point (False where a frame has fewer foreground / background pixels than asked for)."""
        n = frames_set_t.shape[0]
        dev = frames_set_t.device
        k = self.cyc_n_frames
        if source_selector is None:
            source_selector = torch.randint(n, (k,), device=dev)
            target_selector = torch.randint(n, (k,), device=dev)
        h, w = fg_masks.shape[-2:]
        n_fg = int(self.cyc_batch_size_per_frame * self.cyc_fg_points_ratio)
        n_bg = self.cyc_batch_size_per_frame - n_fg
        src_t = frames_set_t[source_selector].long()                                  # [k] frame numbers
        fg = (fg_masks[src_t.to(fg_masks.device)] > 0).reshape(k, h * w).to(dev)
        keys = torch.rand(k, h * w, device=dev)
        neg = torch.full_like(keys, -1.0)
        from .train_ops import topk_rows                                               # (torch.topk, graph-safe for rows this long)
        pick_fg = topk_rows(torch.where(fg, keys, neg), min(n_fg, h * w))
        pick_bg = topk_rows(torch.where(fg, neg, keys), min(n_bg, h * w))
        cells = torch.cat([pick_fg[1], pick_bg[1]], dim=1)                            # [k, n_fg + n_bg]
        valid = torch.cat([pick_fg[0], pick_bg[0]], dim=1) >= 0
        t_col = src_t[:, None].expand_as(cells)
        pts = torch.stack([(cells % w).float(), torch.div(cells, w, rounding_mode="floor").float(), t_col.float()], dim=2)
        src_idx = source_selector[:, None].expand_as(cells)
        tgt_idx = target_selector[:, None].expand_as(cells)
        return pts.reshape(-1, 3), src_idx.reshape(-1).long(), tgt_idx.reshape(-1).long(), valid.reshape(-1)

    def get_cycle_consistency_terms(self, frames_set_t, fg_masks):
        """get_cycle_consistent_preds (tracker.py:182-301) with static shapes and no host read: EVERY sampled point is tracked
        source -> target with gradients (its detached result is the target point) and target -> source with gradients -- the
        detached result of that second pass IS the return point of the reference's no-grad filter pass (the same embeddings,
        the same inputs), so two tracker passes serve where the reference runs four (round 3 ran three); `keep` [M]
        flags the points that are valid and return within cyc_thresh px -- the rows the reference would have kept.  The caller
        weights by `keep` (a batch without any consistent point gives zero loss instead of the reference's re-draw)."""
        pts, src_idx, tgt_idx, valid = self._cycle_point_sets_static(frames_set_t, fg_masks)
        unnorm = lambda c: self.range_normalizer.unnormalize(c, src=(-1, 1), dims=[0, 1])
        t_of = frames_set_t.to(pts.device).float()
        src_tgt = self.get_point_predictions((pts, src_idx, tgt_idx, frames_set_t), self.frame_embeddings)
        with torch.no_grad():
            tgt_pts = torch.cat([unnorm(src_tgt.detach()), t_of[tgt_idx][:, None]], dim=1)
        tgt_src = self.get_point_predictions((tgt_pts, tgt_idx, src_idx, frames_set_t), self.frame_embeddings)
        with torch.no_grad():
            back_xy = unnorm(tgt_src.detach())
            dist = torch.norm(pts[:, :2] - back_xy[:, :2], dim=1)
            keep = valid & (dist <= self.cyc_thresh)
        return {
            "source_coords": self.range_normalizer(pts, dst=[-1, 1]),
            "target_coords": self.range_normalizer(tgt_pts, dst=[-1, 1]),
            "source_target_coords": src_tgt[:, :2],
            "target_source_coords": tgt_src[:, :2],
            "cycle_consistency_dists": dist,
            "keep": keep,
        }

    @torch.no_grad()
    def get_cycle_consistent_coords(self, frames_set_t, fg_masks):
        """tracker.py:182-259: track the sampled points source -> target -> source with the current embeddings
        (`self.frame_embeddings` of the preceding forward) and keep those that come back within cyc_thresh px.  All pairs
        go through the tracker kernels as ONE batch per direction."""
        src_pts, src_idx, tgt_idx = self._cycle_point_sets(frames_set_t, fg_masks)
        emb = self.frame_embeddings.detach()
        unnorm = lambda c: self.range_normalizer.unnormalize(c, src=(-1, 1), dims=[0, 1])
        t_of = frames_set_t.to(src_pts.device).float()
        tgt_xy = unnorm(self.get_point_predictions((src_pts, src_idx, tgt_idx, frames_set_t), emb))
        tgt_pts = torch.cat([tgt_xy, t_of[tgt_idx][:, None]], dim=1)
        back_xy = unnorm(self.get_point_predictions((tgt_pts, tgt_idx, src_idx, frames_set_t), emb))
        keep = (torch.norm(src_pts[:, :2] - back_xy[:, :2], dim=1) <= self.cyc_thresh).nonzero()[:, 0]  # ONE host read
        src_t, tgt_t = t_of[src_idx][keep], t_of[tgt_idx][keep]
        norm_t = lambda t: self.range_normalizer(t[:, None].repeat(1, 3), dst=(-1, 1), dims=[2])[:, 2]
        return {
            "source_points": src_pts[keep],
            "target_points": tgt_pts[keep],
            "cycle_points": back_xy[keep],
            "source_frame_indices": src_idx[keep],
            "target_frame_indices": tgt_idx[keep],
            "source_times_normalized": norm_t(src_t),
            "target_times_normalized": norm_t(tgt_t),
        }

    def get_cycle_consistent_preds(self, frames_set_t, fg_masks):
        """tracker.py:261-301: the cycle-consistent pairs re-tracked in both directions WITH gradients, plus the
        quantities of the cycle loss (dino_tracker.py:332-339)."""
        while True:
            cyc = self.get_cycle_consistent_coords(frames_set_t, fg_masks)
            if cyc["source_points"].shape[0] > 0:
                break
        src_tgt = self.get_point_predictions(
            (cyc["source_points"], cyc["source_frame_indices"], cyc["target_frame_indices"], frames_set_t), self.frame_embeddings)
        tgt_src = self.get_point_predictions(
            (cyc["target_points"], cyc["target_frame_indices"], cyc["source_frame_indices"], frames_set_t), self.frame_embeddings)
        return {
            "source_coords": self.range_normalizer(cyc["source_points"], dst=[-1, 1]),
            "target_coords": self.range_normalizer(cyc["target_points"], dst=[-1, 1]),
            "source_target_coords": src_tgt[:, :2],
            "target_source_coords": tgt_src[:, :2],
            "cycle_consistency_dists": torch.norm(cyc["cycle_points"][:, :2] - cyc["source_points"][:, :2], dim=1),
            "cycle_points": cyc["cycle_points"],
        }

    # ---- forward (tracker.py:303-325) ----------------------------------------------------------------------------
    def forward(self, inp, use_raw_features=False):
        """inp = (source_points B x 3 in pixels (x,y,t), source_frame_indices B, target_frame_indices B,
        frames_set_t n): embeddings are sampled at source_points in frame frames_set_t[source_frame_indices] and
        tracked into frame frames_set_t[target_frame_indices]; returns B x 2 in [-1,1]."""
        source_points, source_frame_indices, target_frame_indices, frames_set_t = inp
        if self.training:
            return self._forward_train(inp, use_raw_features)
        if not use_raw_features and (self._refined is None or self.refined_is_stale()):
            self.cache_refined_embeddings()
        feats = self.features(use_raw_features)
        fs = frames_set_t.to(self.device).long()
        src_t = fs[source_frame_indices.to(self.device).long()].to(torch.int32).contiguous()
        tgt_t = fs[target_frame_indices.to(self.device).long()].to(torch.int32).contiguous()
        xy = source_points[:, :2].to(self.device, torch.float32).contiguous()
        emb = ops.sample_points(self.geom, feats[0], xy, src_t)
        B = xy.shape[0]
        out = torch.empty((B, 2), dtype=torch.float32, device=self.device)
        order = torch.argsort(tgt_t, stable=True).to(torch.int32)  # index plumbing: group sources by target frame
        self.track_sources(feats, emb, order, tgt_t[order.long()].contiguous(), order, out, B, normalized=True)
        return out
