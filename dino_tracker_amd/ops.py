"""Thin tensor-level wrappers over the C-ABI (include/dtk.h): pointer extraction, shape checks, stream hand-off.
PyTorch is used for device memory and streams only; every function here requires GPU tensors."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import Geom, TrackOpts, TrackStats, check, lib

TRACK_EXACT, TRACK_MFMA = 0, 1
TIER_AUTO, TIER_WHOLE_MAP = 0, 1


def _p(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("dino_tracker_amd: tensor is not on a GPU -- the hot path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("dino_tracker_amd: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"dino_tracker_amd: expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pack_features(chw: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[T,C,h,w] fp32 -> token-major [T,h*w,C] + per-cell norms [T,h*w]."""
    T, C, h, w = chw.shape
    thwc = torch.empty((T, h * w, C), dtype=torch.float32, device=chw.device)
    norms = torch.empty((T, h * w), dtype=torch.float32, device=chw.device)
    check(lib().dtk_pack_features(_p(chw, torch.float32), _p(thwc), _p(norms), T, C, h * w, _stream()))
    return thwc, norms


def unpack_features(thwc: torch.Tensor, h: int, w: int) -> torch.Tensor:
    T, HW, C = thwc.shape
    chw = torch.empty((T, C, h, w), dtype=torch.float32, device=thwc.device)
    check(lib().dtk_unpack_features(_p(thwc, torch.float32), _p(chw), T, C, HW, _stream()))
    return chw


def feature_norms(thwc: torch.Tensor) -> torch.Tensor:
    T, HW, C = thwc.shape
    norms = torch.empty((T, HW), dtype=torch.float32, device=thwc.device)
    check(lib().dtk_feature_norms(_p(thwc, torch.float32), _p(norms), T, C, HW, _stream()))
    return norms


def sample_points(g: Geom, feat: torch.Tensor, xy: torch.Tensor, t_idx: torch.Tensor,
                  out: Optional[torch.Tensor] = None, out_row: Optional[torch.Tensor] = None) -> torch.Tensor:
    B = xy.shape[0]
    if out is None:
        out = torch.empty((B, g.C), dtype=torch.float32, device=feat.device)
    check(lib().dtk_sample_points(g, _p(feat, torch.float32), _p(xy, torch.float32), _p(t_idx, torch.int32),
                                  _p(out_row, torch.int32), _p(out, torch.float32), B, _stream()))
    return out


def sample_grid(feat: torch.Tensor, ph: int, pw: int, pts: torch.Tensor) -> torch.Tensor:
    """utils.bilinear_interpolate_video semantics: feat token-major [T, ph*pw, C], pts [B,3] = (x, y, t) in [-1,1]."""
    T, HW, C = feat.shape
    if HW != ph * pw:
        raise RuntimeError(f"sample_grid: {HW} cells != {ph} x {pw}")
    B = pts.shape[0]
    out = torch.empty((B, C), dtype=torch.float32, device=feat.device)
    check(lib().dtk_sample_grid(_p(feat, torch.float32), T, C, ph, pw, _p(pts, torch.float32), _p(out), B, _stream()))
    return out


def normalized_conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """NormalizedConv2d.forward (stride 1, padding k // 2) on the device."""
    B, Cin, H, W = x.shape
    Cout, Cin2, k, k2 = weight.shape
    if Cin2 != Cin or k != k2:
        raise RuntimeError(f"normalized_conv2d: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}")
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    check(lib().dtk_normalized_conv2d(_p(x, torch.float32), _p(weight, torch.float32), _p(bias, torch.float32), _p(y), B,
                                      Cin, Cout, H, W, k, _stream()))
    return y


def corr_maps(g: Geom, feat: torch.Tensor, norms: torch.Tensor, emb: torch.Tensor, tgt: torch.Tensor,
              relu: bool = False) -> torch.Tensor:
    """Cosine maps [M, ph, pw] of emb[m] against frame tgt[m] (models/tracker.py:158-169)."""
    M = emb.shape[0]
    maps = torch.empty((M, g.ph, g.pw), dtype=torch.float32, device=feat.device)
    scratch = torch.empty(max(M, 1), dtype=torch.float32, device=feat.device)
    check(lib().dtk_corr_maps(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(emb, torch.float32), None,
                              _p(tgt, torch.int32), _p(maps), _p(scratch), M, int(relu), _stream()))
    return maps


def head_prepare(sd: Dict[str, torch.Tensor], device) -> torch.Tensor:
    """TrackerHead.cnn_refiner state dict -> packed normalised parameters (dtk.h DTK_HEAD_PARAMS)."""
    w1 = sd["cnn_refiner.0.weight"].detach().to(device=device, dtype=torch.float32).contiguous()
    b1 = sd["cnn_refiner.0.bias"].detach().to(device=device, dtype=torch.float32).contiguous()
    w2 = sd["cnn_refiner.2.weight"].detach().to(device=device, dtype=torch.float32).contiguous()
    b2 = sd["cnn_refiner.2.bias"].detach().to(device=device, dtype=torch.float32).contiguous()
    if tuple(w1.shape) != (16, 1, 3, 3) or tuple(w2.shape) != (1, 16, 3, 3):
        raise RuntimeError("dino_tracker_amd: TrackerHead must be 1->16->1 with 3x3 kernels")
    head = torch.empty(305, dtype=torch.float32, device=device)
    check(lib().dtk_head_prepare(_p(w1), _p(b1), _p(w2), _p(b2), _p(head), _stream()))
    return head


def head_forward(g: Geom, head: torch.Tensor, maps: torch.Tensor, normalized: bool = True) -> torch.Tensor:
    B = maps.shape[0]
    out = torch.empty((B, 2), dtype=torch.float32, device=maps.device)
    check(lib().dtk_head_forward(g, _p(head, torch.float32), _p(maps, torch.float32), _p(out), B, int(normalized),
                                 _stream()))
    return out


def track_workspace_bytes(g: Geom, M: int, method: int, round_sources: int = 0) -> int:
    return int(lib().dtk_track_workspace_bytes(g, M, TrackOpts(method, 0, round_sources, TIER_AUTO, 0)))


def feat_f16_bytes(g: Geom) -> int:
    return int(lib().dtk_feat_f16_bytes(g))


def make_feat_f16(g: Geom, feat: torch.Tensor, norms: torch.Tensor) -> torch.Tensor:
    buf = torch.empty(feat_f16_bytes(g), dtype=torch.uint8, device=feat.device)
    check(lib().dtk_make_feat_f16(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(buf), _stream()))
    return buf


def track(g: Geom, feat: torch.Tensor, norms: torch.Tensor, feat_f16: Optional[torch.Tensor], head: torch.Tensor,
          emb: torch.Tensor, src_row: Optional[torch.Tensor], tgt: torch.Tensor, out_idx: Optional[torch.Tensor],
          out_xy: torch.Tensor, M: int, workspace: torch.Tensor, dM: Optional[torch.Tensor] = None,
          normalized: bool = False, method: int = TRACK_EXACT, round_sources: int = 0, tier: int = TIER_AUTO,
          stats: Optional[TrackStats] = None) -> torch.Tensor:
    """dtk_track.  `stats` (a TrackStats) receives the tier sizes / sync count of this call."""
    opts = TrackOpts(method, int(normalized), round_sources, tier, int(emb.shape[0]) if src_row is not None else 0)
    check(lib().dtk_track(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(feat_f16), _p(head, torch.float32),
                          _p(emb, torch.float32), _p(src_row, torch.int32), _p(tgt, torch.int32),
                          _p(out_idx, torch.int32), _p(out_xy, torch.float32), M, _p(dM, torch.int32), opts, stats,
                          _p(workspace), workspace.numel() * workspace.element_size(), _stream()))
    return out_xy


def argmax_cells(g: Geom, feat: torch.Tensor, norms: torch.Tensor, feat_f16: Optional[torch.Tensor], emb: torch.Tensor,
                 src_row: Optional[torch.Tensor], tgt: torch.Tensor, method: int = TRACK_MFMA,
                 workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """dtk_argmax_cells: (cell [M] int32, cosine [M] f32) of the first maximum of each source's raw cosine map."""
    M = tgt.shape[0]
    cell = torch.empty(M, dtype=torch.int32, device=feat.device)
    cos = torch.empty(M, dtype=torch.float32, device=feat.device)
    if workspace is None:
        workspace = torch.empty(track_workspace_bytes(g, M, method), dtype=torch.uint8, device=feat.device)
    check(lib().dtk_argmax_cells(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(feat_f16), _p(emb, torch.float32),
                                 _p(src_row, torch.int32), _p(tgt, torch.int32), _p(cell), _p(cos), M, method,
                                 _p(workspace), workspace.numel(), _stream()))
    return cell, cos


def bb_nms(g: Geom, feat: torch.Tensor, norms: torch.Tensor, emb: torch.Tensor, src_row: Optional[torch.Tensor],
           tgt: torch.Tensor, box_size: float = 50.0, iou_thresh: float = 0.2, topk: int = 400,
           max_workspace_bytes: int = 1 << 30) -> Tuple[torch.Tensor, torch.Tensor]:
    """dtk_bb_nms: (peak_affs [M, 2], r [M]) of preprocessing_dino_bb/compute_dino_bb_nms.py:12-44 for M sources."""
    M = tgt.shape[0]
    peak = torch.empty((M, 2), dtype=torch.float32, device=feat.device)
    r = torch.empty(M, dtype=torch.float32, device=feat.device)
    if M == 0:
        return peak, r
    nb = min(int(lib().dtk_bb_nms_workspace_bytes(g, M)), max_workspace_bytes)
    ws = torch.empty(nb, dtype=torch.uint8, device=feat.device)
    check(lib().dtk_bb_nms(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(emb, torch.float32),
                           _p(src_row, torch.int32), _p(tgt, torch.int32), float(box_size), float(iou_thresh), int(topk),
                           _p(peak), _p(r), M, _p(ws), nb, _stream()))
    return peak, r


def traj_cos_sims(S: torch.Tensor, tq: torch.Tensor, N: int, T: int) -> torch.Tensor:
    C = S.shape[-1]
    cs = torch.empty((N, T), dtype=torch.float32, device=S.device)
    check(lib().dtk_traj_cos_sims(_p(S, torch.float32), _p(tq, torch.int32), _p(cs), N, T, C, _stream()))
    return cs


class AnchorSources:
    """Device-side result of dtk_build_anchor_sources (worst-case sized buffers, valid prefix given by counts)."""

    def __init__(self, N: int, T: int, device):
        i32 = dict(dtype=torch.int32, device=device)
        self.N, self.T = N, T
        self.n_anchors = torch.empty(N, **i32)
        self.pair_off = torch.empty(N + 1, **i32)
        self.pair_frame = torch.empty(N * T, **i32)
        self.src_row = torch.empty(N * T * T, **i32)
        self.tgt = torch.empty(N * T * T, **i32)
        self.out_idx = torch.empty(N * T * T, **i32)
        self.counts = torch.zeros(4, **i32)
        self.scratch = torch.empty(2 * T + 2, **i32)


def build_anchor_sources(cs: torch.Tensor, anchor_th: float, buf: Optional[AnchorSources] = None) -> AnchorSources:
    N, T = cs.shape
    if buf is None or buf.N != N or buf.T != T:
        buf = AnchorSources(N, T, cs.device)
    check(lib().dtk_build_anchor_sources(_p(cs, torch.float32), float(anchor_th), N, T, _p(buf.n_anchors),
                                         _p(buf.pair_off), _p(buf.pair_frame), _p(buf.src_row), _p(buf.tgt),
                                         _p(buf.out_idx), _p(buf.counts), _p(buf.scratch), _stream()))
    return buf


def occlusion(green: torch.Tensor, pair_off: torch.Tensor, pair_frame: torch.Tensor, traj: torch.Tensor,
              cs: torch.Tensor, anchor_th: float, cos_th: float) -> torch.Tensor:
    N, T = cs.shape
    occ = torch.empty((N, T), dtype=torch.uint8, device=cs.device)
    check(lib().dtk_occlusion(_p(green, torch.float32), _p(pair_off, torch.int32), _p(pair_frame, torch.int32),
                              _p(traj, torch.float32), _p(cs, torch.float32), float(anchor_th), float(cos_th),
                              _p(occ), N, T, _stream()))
    return occ.bool()


def profile_enable(on: bool) -> None:
    check(lib().dtk_profile_enable(int(on)))


def profile_collect() -> Dict[str, Tuple[float, int]]:
    """{kernel name: (total ms, launches)} since profile_enable(True); synchronises."""
    n = lib().dtk_profile_collect()
    return {lib().dtk_profile_name(i).decode(): (lib().dtk_profile_ms(i), lib().dtk_profile_launches(i)) for i in range(n)}


def batchnorm_train_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
                            running_var: Optional[torch.Tensor], momentum: float, eps: float, relu: bool,
                            pre_bias: Optional[torch.Tensor] = None):
    """Train-mode BatchNorm2d (+ fused ReLU) over x [N,C,H,W]: returns (y, save_mean, save_rstd); the running statistics
    are updated in place (dtk_batchnorm_train_forward)."""
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = int(lib().dtk_batchnorm_workspace_bytes(C))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib().dtk_batchnorm_train_forward(_p(x, torch.float32), _p(gamma, torch.float32), _p(beta, torch.float32),
                                            _p(pre_bias, torch.float32), _p(running_mean, torch.float32),
                                            _p(running_var, torch.float32), float(momentum),
                                            float(eps), int(relu), _p(y), _p(mean), _p(rstd), N, C, H * W, _p(ws), nb, _stream()))
    # the kernel wrote the running statistics through raw pointers: tell torch (consumers that cache by `_version`, e.g.
    # delta_dino.weights_key -> the packed eval-mode BN constants, must see the change)
    for buf in (running_mean, running_var):
        if buf is not None:
            torch.autograd.graph.increment_version(buf)
    return y, mean, rstd


def batchnorm_train_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, mean: torch.Tensor,
                             rstd: torch.Tensor, relu: bool, want_pre_bias: bool = False):
    """(dx, dgamma, dbeta, dpre_bias or None) of batchnorm_train_forward (dtk_batchnorm_train_backward)."""
    N, C, H, W = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    dpre = torch.empty(C, dtype=torch.float32, device=x.device) if want_pre_bias else None
    nb = int(lib().dtk_batchnorm_workspace_bytes(C))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib().dtk_batchnorm_train_backward(_p(x, torch.float32), _p(dy, torch.float32), _p(gamma, torch.float32),
                                             _p(beta, torch.float32), _p(mean, torch.float32), _p(rstd, torch.float32),
                                             int(relu), _p(dx), _p(dgamma), _p(dbeta), _p(dpre), N, C, H * W, _p(ws), nb,
                                             _stream()))
    return dx, dgamma, dbeta, dpre


def blurpool_forward(x: torch.Tensor) -> torch.Tensor:
    """BlurPool (filt 4, stride 2, reflect) of x [N,C,H,W] (dtk_blurpool_forward)."""
    N, C, H, W = x.shape
    y = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    check(lib().dtk_blurpool_forward(_p(x, torch.float32), _p(y), N * C, H, W, _stream()))
    return y


def blurpool_backward(dy: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """Adjoint of blurpool_forward: dy [N,C,Ho,Wo] -> dx [N,C,H,W] (dtk_blurpool_backward)."""
    N, C, Ho, Wo = dy.shape
    if (Ho, Wo) != ((H - 1) // 2 + 1, (W - 1) // 2 + 1):
        raise RuntimeError(f"blurpool_backward: {Ho}x{Wo} is not the pooled size of {H}x{W}")
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    check(lib().dtk_blurpool_backward(_p(dy, torch.float32), _p(dx), N * C, H, W, _stream()))
    return dx


# ---- N1: convolutions of the training step on the split-fp16 MFMA GEMM (csrc/train.hip) -----------------------------------
def gemm_nt(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, lda: int, ldb: int, ldc: int,
            batch: int = 1, stride_a: int = 0, stride_b: int = 0, stride_c: int = 0, split_k: int = 1, accumulate: int = 0,
            scale_a: Optional[torch.Tensor] = None, scale_b: Optional[torch.Tensor] = None) -> None:
    """dtk_gemm_nt_f32: C[b][m][n] (+)= sum_k A[b][m][k] B[b][n][k], fp32-grade on the fp16 matrix cores.  The tensors are
    passed as flat storage + explicit leading dimensions / batch strides (in floats); scales are 1-element device tensors."""
    check(lib().dtk_gemm_nt_f32(_p(A, torch.float32), _p(B, torch.float32), _p(C, torch.float32), M, N, K, lda, ldb, ldc, batch,
                                stride_a, stride_b, stride_c, split_k, accumulate, _p(scale_a, torch.float32),
                                _p(scale_b, torch.float32), _stream()))


def im2col(x: torch.Tensor, cols: torch.Tensor, ksize: int, pad: int, dil: int, reflect: bool, layout: int, Kp: int,
           Lp: int = 0) -> None:
    n, C, H, W = x.shape
    check(lib().dtk_im2col(_p(x, torch.float32), _p(cols, torch.float32), n, C, H, W, ksize, pad, dil, int(reflect), layout, Kp, Lp,
                           _stream()))


def col2im(dcols: torch.Tensor, dx: torch.Tensor, ksize: int, pad: int, dil: int, reflect: bool, Kp: int) -> None:
    n, C, H, W = dx.shape
    check(lib().dtk_col2im(_p(dcols, torch.float32), _p(dx, torch.float32), n, C, H, W, ksize, pad, dil, int(reflect), Kp, _stream()))


def transpose_f32(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, batch: int = 1) -> None:
    check(lib().dtk_transpose_f32(_p(src, torch.float32), _p(dst, torch.float32), rows, cols, batch, _stream()))


def resample2d_forward(src: torch.Tensor, dst: torch.Tensor, ylo, ywhi, xlo, xwhi) -> None:
    """dtk_resample2d_forward: src [n, C, hs, ws] -> dst [n, C, hd, wd] through the per-axis two-tap tables."""
    n, c, hs, ws = src.shape
    hd, wd = dst.shape[-2:]
    check(lib().dtk_resample2d_forward(_p(src, torch.float32), _p(dst, torch.float32), n * c, hs, ws, hd, wd, _p(ylo, torch.int32),
                                       _p(ywhi, torch.float32), _p(xlo, torch.int32), _p(xwhi, torch.float32), _stream()))


def resample2d_backward(ddst: torch.Tensor, dsrc: torch.Tensor, yranges, ywhi, xranges, xwhi) -> None:
    n, c, hd, wd = ddst.shape
    hs, ws = dsrc.shape[-2:]
    check(lib().dtk_resample2d_backward(_p(ddst, torch.float32), _p(dsrc, torch.float32), n * c, hs, ws, hd, wd,
                                        _p(yranges, torch.int32), _p(ywhi, torch.float32), _p(xranges, torch.int32),
                                        _p(xwhi, torch.float32), _stream()))


def head_forward_train(g: Geom, packed_head: torch.Tensor, maps: torch.Tensor, normalized: bool = True):
    """dtk_head_forward_train: maps [B, ph*pw] (>= 0) -> (xy [B, 2], stats [B, 4])."""
    B = maps.shape[0]
    out = torch.empty((B, 2), dtype=torch.float32, device=maps.device)
    stats = torch.empty((B, 4), dtype=torch.float32, device=maps.device)
    check(lib().dtk_head_forward_train(g, _p(packed_head, torch.float32), _p(maps, torch.float32), _p(out), _p(stats), B,
                                       int(normalized), _stream()))
    return out, stats


def head_backward(g: Geom, packed_head: torch.Tensor, maps: torch.Tensor, stats: torch.Tensor, grad_out: torch.Tensor,
                  normalized: bool = True):
    """dtk_head_backward: (dmaps [B, ph*pw], dhead [305]) -- see include/dtk.h for the no-fallback condition."""
    B = maps.shape[0]
    dmaps = torch.zeros_like(maps)
    part = torch.empty((B, 305), dtype=torch.float32, device=maps.device)
    check(lib().dtk_head_backward(g, _p(packed_head, torch.float32), _p(maps, torch.float32), _p(stats, torch.float32),
                                  _p(grad_out, torch.float32), _p(dmaps), _p(part), B, int(normalized), _stream()))
    return dmaps, part.sum(dim=0)


def corr_window_backward(g: Geom, feat: torch.Tensor, norms: torch.Tensor, emb: torch.Tensor, tgt: torch.Tensor,
                         maps: torch.Tensor, dmaps: torch.Tensor, stats: torch.Tensor, dfeat: torch.Tensor) -> torch.Tensor:
    """dtk_corr_window_backward: returns demb [B, C]; accumulates into dfeat [T, ph*pw, C] (token-major)."""
    B = emb.shape[0]
    demb = torch.empty_like(emb)
    check(lib().dtk_corr_window_backward(g, _p(feat, torch.float32), _p(norms, torch.float32), _p(emb, torch.float32),
                                         _p(tgt, torch.int32), _p(maps, torch.float32), _p(dmaps, torch.float32),
                                         _p(stats, torch.float32), _p(demb), _p(dfeat, torch.float32), B, _stream()))
    return demb


def conv_split_pack(weight: torch.Tensor, flip_transpose: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """dtk_conv_split_pack: weight [Cout, Cin, 5, 5] -> (Wh, Wl) fp16 planes.  flip_transpose packs the operator of the data
    gradient (its Cin is the forward's Cout)."""
    w = weight.detach().to(torch.float32).contiguous()
    cout, cin = w.shape[:2]
    k_in, k_out = (cout, cin) if flip_transpose else (cin, cout)
    nh = int(lib().dtk_conv_split_weight_halves(k_in, k_out))
    wh = torch.empty(nh, dtype=torch.float16, device=w.device)
    wl = torch.empty(nh, dtype=torch.float16, device=w.device)
    check(lib().dtk_conv_split_pack(_p(w), k_in, k_out, int(flip_transpose), _p(wh), _p(wl), _stream()))
    return wh, wl


def conv_split_input(x: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, border: int = 0,
                     scale: Optional[torch.Tensor] = None) -> None:
    n, c, h, w = x.shape
    check(lib().dtk_conv_split_input(_p(x, torch.float32), n, c, h, w, border, _p(scale, torch.float32), _p(hi), _p(lo), _stream()))


def conv_split_run(hi: torch.Tensor, lo: torch.Tensor, wh: torch.Tensor, wl: torch.Tensor, out_nhwc: torch.Tensor, n: int, h: int,
                   w: int, cin: int, cout: int, dilation: int, zero_pad: bool, fp16_only: bool = False) -> None:
    check(lib().dtk_conv_split_run(_p(hi), _p(lo), _p(wh), _p(wl), _p(out_nhwc, torch.float32), n, h, w, cin, cout, dilation,
                                   int(zero_pad), int(fp16_only), _stream()))


def conv_split_output(y_nhwc: torch.Tensor, out: torch.Tensor, border: int = 0, reflect_fold: bool = False,
                      scale: Optional[torch.Tensor] = None) -> None:
    n, c, h, w = out.shape
    check(lib().dtk_conv_split_output(_p(y_nhwc, torch.float32), n, c, h, w, border, int(reflect_fold), _p(scale, torch.float32),
                                      _p(out, torch.float32), _stream()))


def conv_wgrad_split(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, dilation: int, reflect: bool,
                     scale_dy: Optional[torch.Tensor] = None, fp16_only: bool = False) -> None:
    """dtk_conv_wgrad_split: dw [Cout, Cin, 5, 5] = the weight gradient of the 5 x 5 'same' convolution (overwritten)."""
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    nb = int(lib().dtk_conv_wgrad_split_workspace_bytes(n, cin, cout, h, w, dilation))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib().dtk_conv_wgrad_split(_p(x, torch.float32), _p(dy, torch.float32), _p(dw, torch.float32), n, cin, cout, h, w, dilation,
                                     int(reflect), _p(scale_dy, torch.float32), int(fp16_only), _p(ws), nb, _stream()))


def emb_reg_forward(x: torch.Tensor, raw: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """dtk_emb_reg_forward: x, raw [F, C, h, w] -> (out [2], per-cell sums [3, F * h * w])."""
    f, c, h, w = x.shape
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    sums = torch.empty(3, f * h * w, dtype=torch.float32, device=x.device)
    check(lib().dtk_emb_reg_forward(_p(x, torch.float32), _p(raw, torch.float32), f, c, h * w, _p(sums), _p(out), _stream()))
    return out, sums


def emb_reg_backward(x: torch.Tensor, raw: torch.Tensor, sums: torch.Tensor, gout: torch.Tensor) -> torch.Tensor:
    f, c, h, w = x.shape
    dx = torch.empty_like(x)
    check(lib().dtk_emb_reg_backward(_p(x, torch.float32), _p(raw, torch.float32), _p(sums, torch.float32), _p(gout, torch.float32),
                                     f, c, h * w, _p(dx), _stream()))
    return dx


def contrastive_forward(fe: torch.Tensor, a: torch.Tensor, fidx: torch.Tensor, temp: float):
    """dtk_contrastive_forward: fe [F, C, h, w], a [Q, B, C], fidx [Q] int32 -> (lse [Q, B], saved tensors for backward)."""
    F, C = fe.shape[:2]
    n = fe.shape[2] * fe.shape[3]
    Q, B = a.shape[:2]
    np_ = (n + 3) // 4 * 4
    dev = fe.device
    fet = torch.empty(F * n * C, dtype=torch.float32, device=dev)
    nf = torch.empty(F, n, dtype=torch.float32, device=dev)
    na = torch.empty(Q, B, dtype=torch.float32, device=dev)
    S = torch.empty(Q, B, np_, dtype=torch.float32, device=dev)
    lse = torch.empty(Q, B, dtype=torch.float32, device=dev)
    check(lib().dtk_contrastive_forward(_p(fe, torch.float32), _p(a, torch.float32), _p(fidx, torch.int32), float(temp), Q, B, C, n, F,
                                        _p(fet), _p(nf), _p(na), _p(S), _p(lse), _stream()))
    return lse, (nf, na, S)


def contrastive_backward(fe: torch.Tensor, a: torch.Tensor, fidx: torch.Tensor, temp: float, saved, lse: torch.Tensor,
                         g: torch.Tensor):
    """dtk_contrastive_backward -> (da [Q, B, C], dfe like fe)."""
    F, C = fe.shape[:2]
    n = fe.shape[2] * fe.shape[3]
    Q, B = a.shape[:2]
    nf, na, S = saved
    da = torch.empty_like(a)
    dfe = torch.empty_like(fe)
    nbytes = int(lib().dtk_contrastive_workspace_bytes(Q, B, C, n))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=fe.device)
    check(lib().dtk_contrastive_backward(_p(fe, torch.float32), _p(a, torch.float32), _p(fidx, torch.int32), float(temp), Q, B, C, n, F,
                                         _p(nf, torch.float32), _p(na, torch.float32), _p(S, torch.float32), _p(lse, torch.float32),
                                         _p(g, torch.float32), _p(da), _p(dfe), _p(ws), nbytes, _stream()))
    return da, dfe
