"""DINO best buddies -- preprocessing_dino_bb/extract_dino_best_buddies.py:13-54 on the device.

For every ordered frame pair (s, t), s != t: affinity = cosine(F_s[i], F_t[j]) over all cell pairs, row arg-max
a_st[i] = argmax_j, column arg-max = a_ts[j] = argmax_i, and the mutual pairs i <-> a_st[i] with a_ts[a_st[i]] == i
(`source_bb_indices = feature_range == affinity_target_max[affinity_source_max]`, :40).  The reference builds the
HW x HW affinity matrix per pair with an einsum (8107^2 fp32 = 263 MB, 51 GFLOP at C = 384; T (T-1) pairs); here the
row arg-max of ALL pairs is one `dtk_argmax_cells` call -- the tracker's own fp16-MFMA candidate search with fp32
re-scoring, which is exact -- over sources = every cell of every frame against every other frame, and the mutual test is
index arithmetic on the [T, T, HW] result.  Output format = the reference's dict of
{f"{s}_{t}": {"source_coords", "target_coords", "cos_sims"}}.
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, Optional

import torch

from . import ops
from ._lib import make_geom


def create_meshgrid(h: int, w: int, step: int = 7, patch_size: int = 14, device="cpu") -> torch.Tensor:
    """preprocessing_dino_bb/dino_bb_utils.py:5-15: (x, y) pixel centre of every token, row-major."""
    start = patch_size // 2
    x = torch.arange(start, w, step=step, device=device).float()
    y = torch.arange(start, h, step=step, device=device).float()
    yy, xx = torch.meshgrid(y, x, indexing="ij")
    return torch.stack([xx.reshape(-1), yy.reshape(-1)], dim=-1)


@torch.no_grad()
def row_argmax_all_pairs(feat: torch.Tensor, norms: torch.Tensor, g, method: int = ops.TRACK_MFMA,
                         frames_per_call: Optional[int] = None):
    """feat [T, HW, C] token-major -> (cell [T, T, HW] int32, cos [T, T, HW] f32): [s, t, i] = arg-max over the cells of
    frame t of cos(F_s[i], F_t[.]) and its value (the diagonal s == t is computed too and ignored by the caller)."""
    T, HW, C = feat.shape
    dev = feat.device
    f16 = ops.make_feat_f16(g, feat, norms) if method == ops.TRACK_MFMA else None
    emb = feat.reshape(T * HW, C)
    cell = torch.empty((T, T, HW), dtype=torch.int32, device=dev)
    cos = torch.empty((T, T, HW), dtype=torch.float32, device=dev)
    # sources of one call: every cell of every frame against `tt` target frames, ordered by target frame
    tt = frames_per_call or max(1, min(T, (1 << 23) // (T * HW)))
    rows = torch.arange(T * HW, device=dev, dtype=torch.int32)
    ws = None
    for t0 in range(0, T, tt):
        n = min(tt, T - t0)
        src_row = rows.repeat(n)
        tgt = torch.arange(t0, t0 + n, device=dev, dtype=torch.int32).repeat_interleave(T * HW)
        M = src_row.shape[0]
        need = ops.track_workspace_bytes(g, M, method)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        c, v = ops.argmax_cells(g, feat, norms, f16, emb, src_row, tgt, method, ws)
        cell[:, t0:t0 + n] = c.view(n, T, HW).permute(1, 0, 2)
        cos[:, t0:t0 + n] = v.view(n, T, HW).permute(1, 0, 2)
    return cell, cos


@torch.no_grad()
def extract_best_buddies(features: torch.Tensor, h: int, w: int, stride: int = 7, patch_size: int = 14,
                         device: str = "cuda:0", method: int = ops.TRACK_MFMA) -> Dict[str, Dict[str, torch.Tensor]]:
    """features: T x C x H' x W' (dino_embed_video.pt) -> the reference's best-buddies dict."""
    T, C, hh, ww = features.shape
    g = make_geom(T, C, h, w, patch_size, stride)
    if (g.ph, g.pw) != (hh, ww):
        raise RuntimeError(f"features {hh}x{ww} do not match the {g.ph}x{g.pw} token grid of a {h}x{w} frame")
    if C % 32 != 0:
        method = ops.TRACK_EXACT
    feat, norms = ops.pack_features(features.to(device, torch.float32).contiguous())
    cell, cos = row_argmax_all_pairs(feat, norms, g, method)
    coords = create_meshgrid(h, w, stride, patch_size, device)
    idx = torch.arange(hh * ww, device=device)
    out = {}
    for s in range(T):
        for t in range(T):
            if s == t:
                continue
            a_st = cell[s, t].long()          # affinity_source_max
            a_ts = cell[t, s].long()          # affinity_target_max (arg-max over sources for every target cell)
            mutual = a_ts[a_st] == idx
            tgt_idx = a_st[mutual]
            out[f"{s}_{t}"] = {"source_coords": coords[mutual], "target_coords": coords[tgt_idx], "cos_sims": cos[s, t][mutual]}
    return out


def run(args):
    features = torch.load(args.dino_emb_path)
    bb = extract_best_buddies(features, args.h, args.w, args.stride)
    os.makedirs(os.path.dirname(args.out_path), exist_ok=True)
    torch.save(bb, args.out_path)
    print(f"Saved best buddies to {args.out_path}")


if __name__ == "__main__":  # same flags as the reference script
    parser = argparse.ArgumentParser()
    parser.add_argument("--dino-emb-path", type=str, required=True)
    parser.add_argument("--h", type=int, required=True)
    parser.add_argument("--w", type=int, required=True)
    parser.add_argument("--stride", type=int, default=7)
    parser.add_argument("--out-path", type=str, required=True)
    run(parser.parse_args())
