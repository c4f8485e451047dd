"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement (plain torch) of DINO-Tracker's inference hot path.

This module is the ORACLE the HIP path is checked against.  It may be imported only from `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg -- never from `dino_tracker_amd/`.

Pinning status (see tests/test_oracle_vs_reference.py and tests/golden/):
  * tracker path (sampling, cosine maps, TrackerHead, ModelInference.infer, occlusion): PINNED against the
    un-modified reference run on CPU in the build container (oracle/ref_harness.py) and against golden
    fixtures written by tests/golden/make_golden.py from that reference.
  * DeltaDINO / align_cnn_vit_features: PINNED against the reference modules, with `antialiased_cnns.BlurPool`
    being a restatement itself (oracle/shims/antialiased_cnns) -- BlurPool is "parity unpinned".
  * DINOv2 encoder: upstream facebookresearch/dinov2 is un-vendored (torch.hub, network).  Everything the
    REFERENCE does around the network (models/extractor.py:23-150: stride surgery, `_fix_pos_enc`, hooks, layer
    mean, qkv facets; utils.py:33-72: normalisation, CLS removal, layout) is PINNED: tests/golden/p1_small.npz is
    written by the un-modified VitExtractor / get_dino_features_video running on a DINOv2-API module
    (oracle/shims/dinov2_stub) and `vit_tokens` / `vit_all_tokens` / `vit_qkv` reproduce it.  The block arithmetic
    INSIDE that module is restated from upstream's published definition and cross-checked against the independent
    `transformers.Dinov2Model` port: "parity unpinned" for the blocks themselves.

Device form (round 5): every function takes its device from its inputs, so the same restatement runs on `cuda` in fp32 --
the full benchmark configuration (854 x 476 x 90, 1024 queries: 8.4 M correlation maps) costs seconds there and the parity
tests compare EVERY position with it.  On a CUDA tensor the vendor convolution / attention libraries are avoided on purpose
(MIOpen may pick Winograd or JIT-compile per shape; SDPA may pick a fused kernel): convolutions become unfold + matmul,
BatchNorm / BlurPool explicit arithmetic, attention softmax(QK^T)V -- plain fp32 sums either way (`_plain` below).  The CPU
path is unchanged (it is the one pinned against the reference); tests/test_gpu_fullsize.py pins the device form against the
CPU form on a sample.

All citations are file:line in /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-8  # models/tracker.py:14


def _plain(x: torch.Tensor) -> bool:
    """True on a device tensor: use plain-arithmetic forms instead of the vendor convolution / attention libraries."""
    return x.is_cuda


def _conv2d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], padding: int = 0, dilation: int = 1,
            stride: int = 1) -> torch.Tensor:
    """F.conv2d; on a device tensor the same sums as unfold (im2col) + fp32 matrix products, a few images at a time (the
    column matrix of a 5x5 Delta-DINO layer is 0.65 GB per frame at 238 x 427, that of the head's second layer 4.7 MB per map)."""
    if not _plain(x):
        return F.conv2d(x, w, b, padding=padding, dilation=dilation, stride=stride)
    n, ci, h, wd = x.shape
    co, _, kh, kw = w.shape
    ho = (h + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    wo = (wd + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    wm = w.reshape(co, -1)
    out = x.new_empty((n, co, ho, wo))
    per_image = ci * kh * kw * ho * wo * 4
    bs = max(1, (1 << 31) // per_image)
    for i in range(0, n, bs):
        cols = F.unfold(x[i:i + bs], (kh, kw), dilation=dilation, padding=padding, stride=stride)   # [bs, ci kh kw, L]
        y = torch.matmul(wm, cols)
        if b is not None:
            y = y + b[None, :, None]
        out[i:i + bs] = y.reshape(-1, co, ho, wo)
    return out


# --------------------------------------------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------------------------------------------
def feature_grid(video_h: int, video_w: int, patch: int = 14, stride: int = 7) -> Tuple[int, int]:
    """Number of ViT tokens per axis, models/extractor.py:171-177."""
    return 1 + (video_h - patch) // stride, 1 + (video_w - patch) // stride


def points_to_grid_coords(points: torch.Tensor, video_h: int, video_w: int, patch: int = 14, stride: int = 7):
    """models/tracker.py:77-94: pixel (x, y[, t]) -> grid_sample coordinates in [-1, 1] of the token grid
    whose first / last token centres sit at patch/2 and last_coord."""
    half = patch / 2
    last_h = ((video_h - patch) // stride) * stride + half
    last_w = ((video_w - patch) // stride) * stride + half
    ah, aw = 2 / (last_h - half), 2 / (last_w - half)
    bh, bw = 1 - last_h * 2 / (last_h - half), 1 - last_w * 2 / (last_w - half)
    a = torch.tensor([[aw, ah, 1.0]], dtype=points.dtype, device=points.device)
    b = torch.tensor([[bw, bh, 0.0]], dtype=points.dtype, device=points.device)
    return a * points + b


def sample_embeddings_literal(emb: torch.Tensor, pts_norm: torch.Tensor) -> torch.Tensor:
    """Literal form of Tracker.sample_embeddings (models/tracker.py:96-111 -> utils.py:75-101):
    5-D grid_sample of the [1,C,T,h,w] volume, time normalised by (T-1), border clamp, align_corners."""
    t = emb.shape[0]
    vol = emb.permute(1, 0, 2, 3)[None]  # 1 C T h w
    g = pts_norm[None, None, :, None].clone()
    if t > 1:
        g[..., 2] = g[..., 2] / (t - 1)
    g[..., 2] = g[..., 2] * 2 - 1
    out = F.grid_sample(vol, g, align_corners=True, padding_mode="border")  # 1 C 1 B 1
    return out[0, :, 0, :, 0].permute(1, 0)


def sample_bilinear(emb: torch.Tensor, xy_px: torch.Tensor, t_idx: torch.Tensor, video_h: int, video_w: int,
                    patch: int = 14, stride: int = 7) -> torch.Tensor:
    """Algorithmic form: cell coordinate u=(x-patch/2)/stride, v=(y-patch/2)/stride, clamped to the
    grid, bilinear inside frame t (t integral).  emb [T,C,h,w]; xy_px [B,2]; t_idx [B] -> [B,C]."""
    _, c, h, w = emb.shape
    half = patch / 2
    u = ((xy_px[:, 0] - half) / stride).clamp(0, w - 1)
    v = ((xy_px[:, 1] - half) / stride).clamp(0, h - 1)
    u0 = u.floor().clamp(max=w - 1)
    v0 = v.floor().clamp(max=h - 1)
    fu, fv = u - u0, v - v0
    u0, v0 = u0.long(), v0.long()
    u1, v1 = (u0 + 1).clamp(max=w - 1), (v0 + 1).clamp(max=h - 1)
    t = t_idx.long()
    f00, f01 = emb[t, :, v0, u0], emb[t, :, v0, u1]
    f10, f11 = emb[t, :, v1, u0], emb[t, :, v1, u1]
    fu, fv = fu[:, None], fv[:, None]
    return (f00 * (1 - fu) + f01 * fu) * (1 - fv) + (f10 * (1 - fu) + f11 * fu) * fv


# --------------------------------------------------------------------------------------------------------------
# correlation + tracker head
# --------------------------------------------------------------------------------------------------------------
def cosine_maps(src: torch.Tensor, frames: torch.Tensor) -> torch.Tensor:
    """models/tracker.py:158-169 for the diagonal only: src [B,C], frames [B,C,h,w] (already the target frame of
    each source) -> [B,h,w]: <s, F(:,r,c)> / max(|s| |F(:,r,c)|, 1e-8)."""
    dot = torch.einsum("bc,bchw->bhw", src, frames)
    den = src.norm(dim=1)[:, None, None] * frames.norm(dim=1)
    return dot / den.clamp(min=EPS)


def normalized_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """models/networks/conv_norm.py:34-46: each (out,in) 3x3 kernel divided by its own sum; |sum|<1e-8 -> sign*1e-8."""
    s = w.sum(dim=(2, 3), keepdim=True).clone()
    bad = s.abs() < EPS
    s[bad] = torch.sign(s[bad]) * EPS
    return w / s


def head_refiner(x: torch.Tensor, head: Dict[str, torch.Tensor]) -> torch.Tensor:
    """cnn_refiner of TrackerHead (models/networks/tracker_head.py:54-58): x [B,1,h,w] -> z [B,1,h,w]."""
    w1 = normalized_conv_weight(head["cnn_refiner.0.weight"])
    w2 = normalized_conv_weight(head["cnn_refiner.2.weight"])
    if _plain(x):
        # device form: the two 3x3 layers as two plain fp32 matrix products over all cells of all maps, channel-last
        #   hid[cell, 16] = relu(taps[cell, 9] W1^T + b1);  P[cell, 9] = hid W2 (one plane per tap);  z = b2 + sum of the shifted planes
        n, _, h, w = x.shape
        co = w1.shape[0]
        xp = F.pad(x[:, 0], (1, 1, 1, 1))
        taps = torch.stack([xp[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], dim=-1)
        hid = F.relu(taps.reshape(-1, 9) @ w1.reshape(co, 9).t() + head["cnn_refiner.0.bias"])
        pp = F.pad((hid @ w2.reshape(co, 9)).reshape(n, h, w, 9), (0, 0, 1, 1, 1, 1))
        z = None
        for dy in range(3):
            for dx in range(3):
                term = pp[:, dy:dy + h, dx:dx + w, dy * 3 + dx]
                z = term if z is None else z + term
        return (z + head["cnn_refiner.2.bias"])[:, None]
    hid = F.relu(F.conv2d(x, w1, head["cnn_refiner.0.bias"], padding=1))
    return F.conv2d(hid, w2, head["cnn_refiner.2.bias"], padding=1)


def tracker_head(x: torch.Tensor, head: Dict[str, torch.Tensor], video_h: int, video_w: int, patch: int = 14,
                 stride: int = 7, radius: float = 35.0, return_aux: bool = False,
                 force_argmax: Optional[torch.Tensor] = None):
    """TrackerHead.forward (models/networks/tracker_head.py:107-121) on the ReLU'd cost volume x [B,h,w] (>=0).
    Returns normalised (x,y) in [-1,1] ([B,2]); with return_aux also (argmax_flat, z, fallback_mask).
    force_argmax [B] (test helper, not in the reference): the flat cell to take instead of the arg-max -- what the head
    returns when a near-tie of the map is decided the other way (tie_arbiter below)."""
    b, h, w = x.shape
    k = x.reshape(b, -1).argmax(dim=1) if force_argmax is None else force_argmax.long()  # :115  (first maximum)
    row, col = k // w, k % w
    z = head_refiner(x[:, None], head)[:, 0]
    p = torch.softmax(z.reshape(b, -1), dim=1).reshape(b, h, w)  # :100-105
    half = patch // 2
    ys = torch.arange(h, dtype=torch.float32, device=x.device) * stride + half  # :72-78
    xs = torch.arange(w, dtype=torch.float32, device=x.device) * stride + half
    px, py = (col * stride + half).float(), (row * stride + half).float()
    d2 = (xs[None, None, :] - px[:, None, None]) ** 2 + (ys[None, :, None] - py[:, None, None]) ** 2
    mask = d2.sqrt() <= radius  # :84
    q = p * mask
    qs = q.sum(dim=(1, 2))
    fb = qs < 1e-8  # :86-94
    if fb.any():
        uni = 1.0 / mask[fb].sum(dim=(1, 2)).float()
        q[fb] = (q[fb] + uni[:, None, None]) * mask[fb]
        qs[fb] = q[fb].sum(dim=(1, 2))
    x_hat = (q * xs[None, None, :]).sum(dim=(1, 2)) / qs  # :96
    y_hat = (q * ys[None, :, None]).sum(dim=(1, 2)) / qs
    out = torch.stack([2 * x_hat / (video_w - 1) - 1, 2 * y_hat / (video_h - 1) - 1], dim=1)  # :112,121
    if return_aux:
        return out, k, z, fb
    return out


def track(src: torch.Tensor, feats: torch.Tensor, tgt: torch.Tensor, head: Dict[str, torch.Tensor], video_h: int,
          video_w: int, patch: int = 14, stride: int = 7, chunk: Optional[int] = None,
          src_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Track source embeddings src [M,C] into frames tgt [M] of feats [T,C,h,w]; returns pixel (x,y) [M,2]
    (models/tracker.py:171-180 + model_inference.py:52 un-normalisation).  Sources are grouped by target frame so
    that each frame's maps are one matrix product (the same sums as `cosine_maps`, which gathers a frame per source).
    src_row [M] (optional): source m is row src_row[m] of the table `src` (the anchor stage repeats every trajectory
    embedding once per anchor frame; the table form keeps 8.3 M x C floats from being materialised)."""
    m = src.shape[0] if src_row is None else src_row.shape[0]
    out = src.new_zeros((m, 2))
    if m == 0:
        return out
    if chunk is None:
        chunk = 4096 if _plain(src) else 64   # maps per product: a host-cache-sized slice, or a launch-amortising one
    _, c, h, w = feats.shape
    order = torch.argsort(tgt, stable=True)
    frames, counts = torch.unique_consecutive(tgt[order], return_counts=True)
    snorm = src.norm(dim=1)
    pos = 0
    for f, n in zip(frames.tolist(), counts.tolist()):
        idx = order[pos:pos + n]
        pos += n
        fr = feats[f].reshape(c, h * w)
        fnorm = fr.norm(dim=0)
        for i in range(0, n, chunk):
            ii = idx[i:i + chunk]
            rows = ii if src_row is None else src_row[ii]
            x = (src[rows] @ fr) / (snorm[rows, None] * fnorm[None]).clamp(min=EPS)
            o = tracker_head(F.relu(x).reshape(-1, h, w), head, video_h, video_w, patch, stride)
            out[ii] = torch.stack([(o[:, 0] + 1) / 2 * (video_w - 1), (o[:, 1] + 1) / 2 * (video_h - 1)], dim=1)
    return out


def fp32_dot_band(c: int) -> float:
    """What fp32 rounding can move a cosine of two C-vectors by, between two evaluations that sum in different orders:
    sqrt(C) 2^-24 per evaluation (the standard probabilistic bound on a length-C dot product of unit vectors; the worst case
    C 2^-24 is 20 x larger and never approached), one evaluation on each side -> 2 sqrt(C) 2^-24 (2.3e-6 at C = 384; the
    near-ties met so far differ by 1 - 3 fp32 ulps of a cosine, 6e-8 - 1.9e-7)."""
    return 2.0 * math.sqrt(c) * 2.0 ** -24


def tie_arbiter(feats: torch.Tensor, q_xy_t: torch.Tensor, t: int, got_xy: torch.Tensor, head: Dict[str, torch.Tensor],
                video_h: int, video_w: int, delta: float, patch: int = 14, stride: int = 7, tol_px: float = 1e-3,
                max_cells: int = 256, dev_feats: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """Test helper (not in the reference): decides whether a position `got_xy` that differs from this oracle's for the
    query q_xy_t = (x, y, t_q) in frame t is the reference's answer for a map whose near-tie went the other way.
    The cosine map is evaluated in FLOAT64 from the fp32 features.  A cell k is admissible when its float64 cosine is within
    its band of the float64 maximum:
        band(k) = delta                                          (the caller's bound on fp32 rounding: fp32_dot_band(C))
                + dev(k*) + dev(k)   if dev_feats is given       (MEASURED, for this map and these cells only: dev(k) = |cos64 of
                                                                  the same query / frame evaluated from the features a device
                                                                  path computed - cos64 from the oracle's features|)
    and for each admissible cell the head (models/networks/tracker_head.py:68-121, fp32 as the reference runs it) is evaluated
    with its arg-max forced there.  Returns the float64 gap of the best-matching admissible cell, its band, its distance to
    got_xy, and the counts.  (Round 4 used one global band of twice the relative feature deviation, 6.6e-4: 3000 x the gap it
    had to cover.  A genuinely wrong cell now has to sit within the fp32 band plus what the device's features measurably
    moved THESE two cosines by.)"""
    tq = int(q_xy_t[2])
    _, c, h, w = feats.shape

    def map64(vol):
        q = sample_bilinear(vol[tq:tq + 1].double(), q_xy_t[None, :2].double(), torch.zeros(1, dtype=torch.long, device=vol.device),
                            video_h, video_w, patch, stride)[0]
        fr = vol[t].double().reshape(c, h * w)
        return F.relu((q @ fr) / (q.norm() * fr.norm(dim=0)).clamp(min=EPS))

    m64 = map64(feats)
    top = m64.max()
    band = torch.full_like(m64, float(delta))
    if dev_feats is not None:
        dev = (map64(dev_feats) - m64).abs()
        band = band + dev + dev[m64.argmax()]
    cells = torch.nonzero(m64 >= top - band)[:, 0]
    if cells.numel() > max_cells:
        cells = cells[torch.argsort(m64[cells], descending=True)[:max_cells]]
    q32 = sample_bilinear(feats, q_xy_t[None, :2], q_xy_t[2:3].long(), video_h, video_w, patch, stride)
    x32 = F.relu(cosine_maps(q32, feats[t][None]))  # the map the reference's head sees
    o = tracker_head(x32.expand(cells.numel(), -1, -1), head, video_h, video_w, patch, stride, force_argmax=cells)
    xy = torch.stack([(o[:, 0] + 1) / 2 * (video_w - 1), (o[:, 1] + 1) / 2 * (video_h - 1)], dim=1)
    d = (xy - got_xy[None].to(xy.device)).norm(dim=1)
    i = int(d.argmin())
    return {"admissible_cells": int(cells.numel()), "gap64": float(top - m64[cells[i]]), "dist_px": float(d[i]),
            "cell": int(cells[i]), "ok": bool(d[i] <= tol_px), "delta": float(band[cells[i]]), "fp32_band": float(delta)}


# --------------------------------------------------------------------------------------------------------------
# ModelInference.infer -- six-step algorithmic restatement
# --------------------------------------------------------------------------------------------------------------
def lower_median(x: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return torch.median(x, dim=dim).values  # torch.median = lower median


def occlusion_for_query(green: torch.Tensor, traj_xy: torch.Tensor, cs: torch.Tensor, anchor_th: float,
                        cos_th: float) -> torch.Tensor:
    """models/model_inference.py:169-177. green [A,T,2], traj_xy [T,2], cs [T] -> occ [T] bool."""
    vis = cs >= anchor_th
    d = (green - traj_xy[vis][:, None]).norm(dim=-1)  # [A,T]
    tau = lower_median(d[:, vis], 0).max()
    return (lower_median(d, 0) > tau) | (cs < cos_th)


def occlusion_margins(green: torch.Tensor, traj_xy: torch.Tensor, cs: torch.Tensor, anchor_th: float,
                      cos_th: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """How far each occlusion decision of occlusion_for_query is from flipping: (|median - tau| in px for frames that
    are not anchors -- anchors satisfy median <= tau by construction --, |cs - cos_th|).  Test helper."""
    vis = cs >= anchor_th
    d = (green - traj_xy[vis][:, None]).norm(dim=-1)
    med = lower_median(d, 0)
    tau = lower_median(d[:, vis], 0).max()
    return (med - tau).abs(), (cs - cos_th).abs()


def infer(feats: torch.Tensor, queries: torch.Tensor, head: Dict[str, torch.Tensor], video_h: int, video_w: int,
          anchor_th: float = 0.7, cos_th: float = 0.6, patch: int = 14, stride: int = 7,
          return_aux: bool = False):
    """ModelInference.infer (models/model_inference.py:203-216) on cached refined features feats [T,C,h,w];
    queries [N,3] = (x,y,t) at model resolution.  Returns traj [N,T,2] f32 (pixels), occ [N,T] bool."""
    t_len = feats.shape[0]
    n = queries.shape[0]
    tq = queries[:, 2].long()
    # 1-2: query embeddings and first-pass trajectories (:8-74)
    q_emb = sample_bilinear(feats, queries[:, :2], tq, video_h, video_w, patch, stride)
    src = q_emb[:, None].expand(n, t_len, -1).reshape(n * t_len, -1)
    tgt = torch.arange(t_len, device=feats.device).repeat(n)
    traj = track(src, feats, tgt, head, video_h, video_w, patch, stride).reshape(n, t_len, 2)
    # 4: embeddings along the trajectory and their cosine to the one at the query frame (:110-126)
    s_emb = sample_bilinear(feats, traj.reshape(-1, 2), tgt, video_h, video_w, patch, stride).reshape(n, t_len, -1)
    ref = s_emb[torch.arange(n, device=feats.device), tq]
    cs = F.cosine_similarity(ref[:, None], s_emb, dim=-1)
    # 5-6: anchors and occlusion (:130-200).  The anchor trajectories of ALL queries go through `track` as one batch (it groups
    # its sources by target frame, so every frame's maps are a few large products instead of one small product per query:
    # the same dot products, ~5 x less wall time on a many-core host); per query they are then cut back out.
    occ = torch.zeros(n, t_len, dtype=torch.bool, device=feats.device)
    greens: List[torch.Tensor] = []
    anchors_of = [torch.nonzero(cs[i] >= anchor_th)[:, 0] for i in range(n)]
    for a in anchors_of:
        if a.numel() == 0:
            raise RuntimeError("stack expects a non-empty TensorList")  # torch.stack([]) at :152
    # source (i, a, t) = the embedding at traj[i, t], tracked into anchor frame a: row i T + t of the table of trajectory embeddings
    t_ar = torch.arange(t_len, device=feats.device)
    a_row = torch.cat([(i * t_len + t_ar)[None].expand(a.numel(), -1).reshape(-1) for i, a in enumerate(anchors_of)])
    a_tgt = torch.cat([a[:, None].expand(-1, t_len).reshape(-1) for a in anchors_of])
    g_all = track(s_emb.reshape(n * t_len, -1), feats, a_tgt, head, video_h, video_w, patch, stride, src_row=a_row)
    pos = 0
    for i, a in enumerate(anchors_of):
        g = g_all[pos:pos + a.numel() * t_len].reshape(a.numel(), t_len, 2)
        pos += a.numel() * t_len
        greens.append(g)
        occ[i] = occlusion_for_query(g, traj[i], cs[i], anchor_th, cos_th)
    if return_aux:
        return traj, occ, cs, greens
    return traj, occ


# --------------------------------------------------------------------------------------------------------------
# Delta-DINO (models/networks/delta_dino.py) + alignment (models/utils.py:7-45)
# --------------------------------------------------------------------------------------------------------------
def blur_pool(x: torch.Tensor) -> torch.Tensor:
    """antialiased_cnns.BlurPool(stride=2, filt_size=4, reflect): see oracle/shims/antialiased_cnns (unpinned)."""
    a = torch.tensor([1.0, 3.0, 3.0, 1.0], device=x.device)
    k2 = a[:, None] * a[None, :] / 64.0
    xp = F.pad(x, (1, 2, 1, 2), mode="reflect")
    if _plain(x):   # the depthwise 4x4 / stride-2 filter as 16 shifted, weighted slices (no grouped-convolution library call)
        ho, wo = (xp.shape[-2] - 4) // 2 + 1, (xp.shape[-1] - 4) // 2 + 1
        out = None
        for i in range(4):
            for j in range(4):
                term = k2[i, j] * xp[..., i:i + 2 * ho - 1:2, j:j + 2 * wo - 1:2]
                out = term if out is None else out + term
        return out
    k = k2[None, None].repeat(x.shape[1], 1, 1, 1)
    return F.conv2d(xp, k, stride=2, groups=x.shape[1])


def delta_dino_cnn(frames: torch.Tensor, sd: Dict[str, torch.Tensor], bn_eps: float = 1e-5) -> torch.Tensor:
    """DeltaDINO.layers in eval mode (delta_dino.py:23-46): 4x [5x5 reflect conv -> BN(running stats) -> ReLU* ->
    BlurPool*] (*: not on the last layer, whose conv has dilation 2).  frames [B,3,H,W] in [0,1]."""
    x = frames
    conv_ids, bn_ids = (0, 4, 8, 12), (1, 5, 9, 13)
    for li, (ci, bi) in enumerate(zip(conv_ids, bn_ids)):
        dil = 2 if li == 3 else 1
        pad = 2 * dil
        x = _conv2d(F.pad(x, (pad,) * 4, mode="reflect"), sd[f"layers.{ci}.weight"], sd[f"layers.{ci}.bias"],
                    dilation=dil)
        if _plain(x):   # eval-mode BatchNorm spelled out: (x - mean) / sqrt(var + eps) * weight + bias
            inv = torch.rsqrt(sd[f"layers.{bi}.running_var"] + bn_eps) * sd[f"layers.{bi}.weight"]
            x = (x - sd[f"layers.{bi}.running_mean"][None, :, None, None]) * inv[None, :, None, None] \
                + sd[f"layers.{bi}.bias"][None, :, None, None]
        else:
            x = F.batch_norm(x, sd[f"layers.{bi}.running_mean"], sd[f"layers.{bi}.running_var"],
                             sd[f"layers.{bi}.weight"], sd[f"layers.{bi}.bias"], training=False, eps=bn_eps)
        if li < 3:
            x = blur_pool(F.relu(x))
    return x


def align_to_vit_grid(cnn: torch.Tensor, vit_h: int, vit_w: int, patch: int = 14, vit_stride: int = 7,
                      cnn_stride: int = 8) -> torch.Tensor:
    """models/utils.py:7-45: bilinear (border, align_corners) resample of the stride-8 CNN map (cell j at pixel 8j)
    at the ViT token centres 7i+7."""
    ch, cw = cnn.shape[-2:]
    br_y, br_x = (ch - 1) * cnn_stride, (cw - 1) * cnn_stride
    vx = torch.arange(vit_w, dtype=torch.float32, device=cnn.device) * vit_stride + patch / 2.0
    vy = torch.arange(vit_h, dtype=torch.float32, device=cnn.device) * vit_stride + patch / 2.0
    gx = -1.0 - 1.0 / br_x + 2.0 * vx / br_x
    gy = -1.0 - 1.0 / br_y + 2.0 * vy / br_y
    grid = torch.stack(torch.meshgrid(gx, gy, indexing="xy"), dim=-1)[None].expand(cnn.shape[0], -1, -1, -1)
    return F.grid_sample(cnn, grid, mode="bilinear", padding_mode="border", align_corners=True)


def refine_features(frames: torch.Tensor, dino: torch.Tensor, sd: Dict[str, torch.Tensor], batch: int = 8) -> torch.Tensor:
    """Tracker.get_refined_embeddings (models/tracker.py:113-129): dino + align(DeltaDINO(frames))."""
    out = torch.empty_like(dino)
    for i in range(0, frames.shape[0], batch):
        cnn = delta_dino_cnn(frames[i:i + batch], sd)
        out[i:i + batch] = dino[i:i + batch] + align_to_vit_grid(cnn, dino.shape[-2], dino.shape[-1])
    return out


# --------------------------------------------------------------------------------------------------------------
# DINOv2 ViT encoder as driven by VitExtractor (models/extractor.py) -- parity unpinned (un-vendored upstream)
# --------------------------------------------------------------------------------------------------------------
VIT_CONFIGS = {  # models/extractor.py:183-222
    "dinov2_vits14": dict(dim=384, depth=12, heads=6),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16),
}


def vit_pos_embed(sd: Dict[str, torch.Tensor], h0: int, w0: int) -> torch.Tensor:
    """models/extractor.py:57-85 (`_fix_pos_enc`).  NB the upstream caller passes (x, w:=H, h:=W) so the reference's
    local `w0` is the token-ROW count and `h0` the token-COLUMN count; here h0/w0 are rows/cols directly.
    bicubic, align_corners=False, scale_factor=((rows+.1)/n, (cols+.1)/n), recompute_scale_factor=False."""
    pe = sd["pos_embed"]
    n = int(math.sqrt(pe.shape[1] - 1))
    d = pe.shape[-1]
    grid = pe[:, 1:].reshape(1, n, n, d).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=((h0 + 0.1) / n, (w0 + 0.1) / n), mode="bicubic", align_corners=False,
                         recompute_scale_factor=False)
    assert grid.shape[-2] == h0 and grid.shape[-1] == w0
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, d)
    return torch.cat([pe[:, :1], grid], dim=1)


def vit_block(x: torch.Tensor, sd: Dict[str, torch.Tensor], i: int, heads: int, return_qkv: bool = False):
    """DINOv2 NestedTensorBlock at eval: x += g1*proj(MHSA(LN1 x)); x += g2*fc2(GELU(fc1(LN2 x))); LN eps 1e-6.
    return_qkv: also the output of attn.qkv [B,S,3D] (what the reference's qkv hook records, extractor.py:107-118)."""
    p = f"blocks.{i}."
    b, s, d = x.shape
    y = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6)
    qkv_flat = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv_flat.reshape(b, s, 3, heads, d // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    if _plain(x):   # softmax(Q K^T / sqrt(d)) V spelled out in fp32 (no fused-attention backend)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5), dim=-1) @ v
    else:
        a = F.scaled_dot_product_attention(q, k, v)
    a = a.transpose(1, 2).reshape(b, s, d)
    x = x + sd[p + "ls1.gamma"] * F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    y = F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                 sd[p + "mlp.fc2.bias"])
    out = x + sd[p + "ls2.gamma"] * y
    return (out, qkv_flat) if return_qkv else out


IMAGENET_MEAN = (0.485, 0.456, 0.406)  # utils.py:46
IMAGENET_STD = (0.229, 0.224, 0.225)


def vit_all_tokens(frame: torch.Tensor, sd: Dict[str, torch.Tensor], model_name: str, layer: Optional[int] = None,
                   stride: int = 7, patch: int = 14, normalize: bool = True, return_qkv: bool = False):
    """VitExtractor.get_feature_from_input(img, [layer]) (models/extractor.py:137-150) for frames [B,3,H,W] in [0,1]
    (normalize=True applies utils.py:46 first): block-`layer` output [B, 1+ph*pw, D], CLS first; with return_qkv also
    the qkv record of that block [B, 1+ph*pw, 3D]."""
    cfg = VIT_CONFIGS[model_name]
    layer = cfg["depth"] - 1 if layer is None else layer
    x = frame
    if normalize:
        m = torch.tensor(IMAGENET_MEAN, dtype=frame.dtype, device=frame.device).view(1, 3, 1, 1)
        s = torch.tensor(IMAGENET_STD, dtype=frame.dtype, device=frame.device).view(1, 3, 1, 1)
        x = (x - m) / s
    ph, pw = feature_grid(frame.shape[-2], frame.shape[-1], patch, stride)
    tok = _conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=stride)
    tok = tok.flatten(2).transpose(1, 2)
    tok = torch.cat([sd["cls_token"].expand(tok.shape[0], -1, -1), tok], dim=1) + vit_pos_embed(sd, ph, pw)
    qkv = None
    for i in range(layer + 1):
        if return_qkv and i == layer:
            tok, qkv = vit_block(tok, sd, i, cfg["heads"], return_qkv=True)
        else:
            tok = vit_block(tok, sd, i, cfg["heads"])
    return (tok, qkv) if return_qkv else tok


def vit_qkv(frame: torch.Tensor, sd: Dict[str, torch.Tensor], model_name: str, layer: int, normalize: bool = True):
    """get_qkv_feature_from_input(img)[layer] (models/extractor.py:152-158): [B, 1+ph*pw, 3D]."""
    return vit_all_tokens(frame, sd, model_name, layer, normalize=normalize, return_qkv=True)[1]


def vit_tokens(frame: torch.Tensor, sd: Dict[str, torch.Tensor], model_name: str, layer: Optional[int] = None,
               stride: int = 7, patch: int = 14, normalize: bool = True) -> torch.Tensor:
    """get_dino_features_video for one frame, facet 'tokens' (utils.py:54-67, models/extractor.py:137-150):
    frame [1,3,H,W] in [0,1] -> block-`layer` output without CLS, laid out [C, ph, pw]."""
    ph, pw = feature_grid(frame.shape[-2], frame.shape[-1], patch, stride)
    tok = vit_all_tokens(frame, sd, model_name, layer, stride, patch, normalize)
    return tok[0, 1:].reshape(ph, pw, -1).permute(2, 0, 1)


# --------------------------------------------------------------------------------------------------------------
# TAP-Vid metrics (eval/metrics.py) -- numpy restatement, pinned against the reference in tests/test_oracle_vs_reference.py
# --------------------------------------------------------------------------------------------------------------
def tapvid_metrics(query_frames, gt_occluded, gt_tracks, pred_occluded, pred_tracks, pred_size, gt_size,
                   query_mode: str = "strided") -> Dict[str, float]:
    """compute_tapvid_metrics_for_video + compute_tapvid_metrics for one video (eval/metrics.py:7-147,204-223):
    tracks [N,T,2] (x,y), flags [N,T], query_frames [N]; both track sets are scaled to the 256 x 256 raster in float32."""
    import numpy as np
    gt = np.array(gt_tracks, dtype=np.float32)
    pr = np.array(pred_tracks, dtype=np.float32)
    gt[..., 0] *= 256 / gt_size[0]
    gt[..., 1] *= 256 / gt_size[1]
    pr[..., 0] *= 256 / pred_size[0]
    pr[..., 1] *= 256 / pred_size[1]
    gocc = np.asarray(gt_occluded).astype(bool)
    pocc = np.asarray(pred_occluded).astype(bool)
    n, t = gocc.shape
    ts = np.arange(t)[None, :]
    qf = np.round(np.asarray(query_frames)).astype(np.int64)[:, None]
    ev = (ts != qf) if query_mode == "strided" else (ts > qf)
    vis, pvis = ~gocc, ~pocc
    out = {"occlusion_accuracy": float(((pocc == gocc) & ev).sum() / ev.sum())}
    d2 = np.square(pr - gt).sum(axis=-1)
    jac, frac = [], []
    for th in (1, 2, 4, 8, 16):
        within = d2 < np.square(th)
        correct = within & vis
        frac.append(float((correct & ev).sum() / (vis & ev).sum()))
        fp = ((~vis) & pvis) | ((~within) & pvis)
        jac.append(float((correct & pvis & ev).sum() / ((vis & ev).sum() + (fp & ev).sum())))
        out[f"pts_within_{th}"] = frac[-1]
        out[f"jaccard_{th}"] = jac[-1]
    out["average_jaccard"] = float(np.mean(jac))
    out["average_pts_within_thresh"] = float(np.mean(frac))
    return out


# --------------------------------------------------------------------------------------------------------------
# DINO best buddies (preprocessing_dino_bb/extract_dino_best_buddies.py:26-48) for one ordered frame pair
# --------------------------------------------------------------------------------------------------------------
def best_buddies_pair(fs: torch.Tensor, ft: torch.Tensor):
    """fs, ft [HW, C] features of the source / target frame -> (source cell indices, target cell indices, cosines) of the
    mutual nearest neighbours, in source order."""
    aff = fs @ ft.t()
    aff = aff / torch.clamp(fs.norm(dim=1)[:, None] * ft.norm(dim=1)[None], min=1e-8)
    smax = torch.argmax(aff, dim=1)
    tmax = torch.argmax(aff, dim=0)
    rng = torch.arange(fs.shape[0], device=fs.device)
    keep = rng == tmax[smax]
    return rng[keep], smax[keep], aff[rng[keep], smax[keep]]


def bb_nms_ratio(aff: torch.Tensor, pw: int, box_size: float = 50.0, iou_thresh: float = 0.2, topk: int = 400,
                 patch: int = 14, stride: int = 7):
    """preprocessing_dino_bb/compute_dino_bb_nms.py:12-44 (`get_bb_sim_indices`) for affinity rows aff [B, HW] over a
    token grid of width pw: (top2 [B, 2], r [B]).  Restated WITHOUT the NMS sweep: of the top-k entries the arg-max is
    always kept by greedy NMS, and the second kept box is the best-scoring one whose IoU with the arg-max box is <= thresh
    (everything scored in between overlaps the arg-max, the only box kept until then); suppressed entries count as 0
    (`affs_filt * filt_idx_mask`, :36) and the answer is the top-2 of that masked list (:38).  Pinned against the
    reference function run with a restated torchvision.ops.batched_nms (tests/test_oracle_vs_reference.py)."""
    b, hw = aff.shape
    vals, idx = torch.topk(aff, k=topk, dim=1)                      # sorted descending; [:, 0] is the arg-max
    half = patch // 2
    cx = (half + stride * (idx % pw)).float()
    cy = (half + stride * (idx // pw)).float()
    x1, x2, y1, y2 = cx - box_size, cx + box_size, cy - box_size, cy + box_size
    iw = (torch.minimum(x2, x2[:, :1]) - torch.maximum(x1, x1[:, :1])).clamp(min=0)
    ih = (torch.minimum(y2, y2[:, :1]) - torch.maximum(y1, y1[:, :1])).clamp(min=0)
    inter = iw * ih
    area = (x2 - x1) * (y2 - y1)
    supp = inter / (area[:, :1] + area - inter) > iou_thresh
    supp[:, 0] = False
    neg = torch.full_like(vals, float("-inf"))
    second = torch.where(~supp, vals, neg)[:, 1:].max(dim=1).values if topk > 1 else neg[:, 0]
    nsupp = supp.sum(dim=1)
    zero1 = torch.where(nsupp >= 1, torch.zeros(b, device=aff.device), neg[:, 0])
    zero2 = torch.where(nsupp >= 2, torch.zeros(b, device=aff.device), neg[:, 0])
    top2 = torch.stack([vals[:, 0], second, zero1, zero2], dim=1).topk(2, dim=1).values
    return top2, top2[:, 1] / top2[:, 0]
