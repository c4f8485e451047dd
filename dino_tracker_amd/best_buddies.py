"""DINO best buddies -- preprocessing_dino_bb/extract_dino_best_buddies.py:13-54 on the device.

For every ordered frame pair (s, t), s != t: affinity = cosine(F_s[i], F_t[j]) over all cell pairs, row arg-max
a_st[i] = argmax_j, column arg-max = a_ts[j] = argmax_i, and the mutual pairs i <-> a_st[i] with a_ts[a_st[i]] == i
(`source_bb_indices = feature_range == affinity_target_max[affinity_source_max]`, :40).  The reference builds the
HW x HW affinity matrix per pair with an einsum (8107^2 fp32 = 263 MB, 51 GFLOP at C = 384; T (T-1) pairs); here the
row arg-max of ALL pairs is one `dtk_argmax_cells` call -- the tracker's own fp16-MFMA candidate search with fp32
re-scoring, which is exact -- over sources = every cell of every frame against every other frame, and the mutual test is
index arithmetic on the [T, T, HW] result.  Output format = the reference's dict of
{f"{s}_{t}": {"source_coords", "target_coords", "cos_sims"}}.

Second half, preprocessing_dino_bb/compute_dino_bb_nms.py:12-78 (`compute_bb_nms_all`, `python -m
dino_tracker_amd.best_buddies nms ...`): per best-buddy source the ambiguity ratio r = (second NMS peak of its affinity row) /
(first peak) from `dtk_bb_nms` -- all frame pairs in ONE call --, then `compute_max_r` (:68-78: the larger of a pair's r and
its reverse pair's) as index arithmetic; adds the reference's keys `peak_coords` (None), `peak_affs` [N, 2], `r` [N].
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


def _cells_of(coords: torch.Tensor, pw: int, stride: int, patch_size: int) -> torch.Tensor:
    """pixel centres (x, y) -> flat cell index (preprocessing_dino_bb/dino_bb_utils.py:17-19: fxy = (xy - patch/2) / stride)."""
    fx = ((coords[:, 0] - patch_size // 2) / stride).long()
    fy = ((coords[:, 1] - patch_size // 2) / stride).long()
    return fy * pw + fx


@torch.no_grad()
def compute_bb_nms_all(dino_bb: Dict[str, Dict[str, Optional[torch.Tensor]]], features: torch.Tensor, h: int = 476, w: int = 854,
                       stride: int = 7, box_size: float = 50.0, iou_thresh: float = 0.2, topk: int = 400,
                       patch_size: int = 14, device: str = "cuda:0") -> Dict[str, Dict[str, Optional[torch.Tensor]]]:
    """compute_dino_bb_nms.py:81-106 (`run`) on the device: every key of the best-buddies dict gets `peak_coords` (None),
    `peak_affs` and `r`, r already maximised over the pair and its reverse (`compute_max_r`).  Keys without buddies
    (`source_coords` None or empty) get None like in the reference."""
    T, C, hh, ww = features.shape
    g = make_geom(T, C, h, w, patch_size, stride)
    if (g.ph, g.pw) != (hh, ww):
        raise RuntimeError(f"features {hh}x{ww} do not match the {g.ph}x{g.pw} token grid of a {h}x{w} frame")
    HW = hh * ww
    if topk > HW:
        raise RuntimeError(f"topk={topk} > {HW} cells per frame (torch.topk would raise in the reference)")
    feat, norms = ops.pack_features(features.to(device, torch.float32).contiguous())
    emb = feat.reshape(T * HW, C)
    keys, rows, tgts, sizes = [], [], [], []
    for key, bb in dino_bb.items():
        sc = bb.get("source_coords")
        if sc is None or sc.shape[0] == 0:
            bb["peak_coords"], bb["peak_affs"], bb["r"] = None, None, None
            continue
        sf, tf = (int(x) for x in key.split("_"))
        cells = _cells_of(sc.to(device), ww, stride, patch_size)
        keys.append(key)
        rows.append((sf * HW + cells).to(torch.int32))
        tgts.append(torch.full((cells.shape[0],), tf, dtype=torch.int32, device=device))
        sizes.append(cells.shape[0])
    if not keys:
        return dino_bb
    peak, r = ops.bb_nms(g, feat, norms, emb, torch.cat(rows), torch.cat(tgts), box_size, iou_thresh, topk)
    pos = 0
    for key, n in zip(keys, sizes):
        dino_bb[key]["peak_coords"] = None
        dino_bb[key]["peak_affs"] = peak[pos:pos + n]
        dino_bb[key]["r"] = r[pos:pos + n].clone()
        pos += n
    # compute_max_r (:68-78): pair i of (s, t) <-> the pair of (t, s) whose source is i's target; both get the larger r
    done = set()
    for key in keys:
        sf, tf = key.split("_")
        rev = f"{tf}_{sf}"
        if key in done or rev not in dino_bb or dino_bb[rev].get("r") is None:
            continue
        bb, bbr = dino_bb[key], dino_bb[rev]
        tgt_cells = _cells_of(bb["target_coords"].to(device), ww, stride, patch_size)
        rev_src = _cells_of(bbr["source_coords"].to(device), ww, stride, patch_size)
        where = torch.full((HW,), -1, dtype=torch.long, device=device)
        where[rev_src] = torch.arange(rev_src.shape[0], device=device)
        j = where[tgt_cells]
        if bool((j < 0).any()):
            raise RuntimeError(f"best buddies {key} / {rev} are not mutual (compute_dino_bb_nms.py:73 asserts the same)")
        m = torch.maximum(bb["r"], bbr["r"][j])
        bb["r"] = m
        bbr["r"][j] = m
        done.add(key)
        done.add(rev)
    return dino_bb


def run_nms(args):
    """The reference's compute_dino_bb_nms.py command line (same flags)."""
    dino_bb = torch.load(args.dino_bb_path)
    features = torch.load(args.dino_emb_path)
    out = compute_bb_nms_all(dino_bb, features, 476, 854, args.stride, args.box_size, args.iou_thresh)
    os.makedirs(os.path.dirname(args.out_path), exist_ok=True)
    torch.save(out, args.out_path)
    print(f"Saved best buddies with NMS ratios to {args.out_path}")


def run(args):
    features = torch.load(args.dino_emb_path)
    bb = extract_best_buddies(features, args.h, args.w, args.stride)
    os.makedirs(os.path.dirname(args.out_path), exist_ok=True)
    torch.save(bb, args.out_path)
    print(f"Saved best buddies to {args.out_path}")


if __name__ == "__main__":  # same flags as the reference scripts; `nms` as first argument selects compute_dino_bb_nms.py's
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "nms":
        parser = argparse.ArgumentParser()
        parser.add_argument("--dino-bb-path", type=str, required=True)
        parser.add_argument("--dino-emb-path", type=str, required=True)
        parser.add_argument("--out-path", type=str, required=True)
        parser.add_argument("--stride", type=int, default=7)
        parser.add_argument("--box-size", type=int, default=50)
        parser.add_argument("--iou-thresh", type=float, default=0.2)
        run_nms(parser.parse_args(sys.argv[2:]))
        sys.exit(0)
    parser = argparse.ArgumentParser()
    parser.add_argument("--dino-emb-path", type=str, required=True)
    parser.add_argument("--h", type=int, required=True)
    parser.add_argument("--w", type=int, required=True)
    parser.add_argument("--stride", type=int, default=7)
    parser.add_argument("--out-path", type=str, required=True)
    run(parser.parse_args())
