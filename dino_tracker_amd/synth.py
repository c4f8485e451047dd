"""Seeded synthetic workloads: there is no dataset, checkpoint or network in this
environment, so benchmarks and parity tests run on generated videos / feature volumes / weights.

Everything is produced on the CPU with an explicit torch.Generator so that the build container, the GPU box and
the golden-fixture script (tests/golden/make_golden.py) see bit-identical inputs.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _box(x: torch.Tensor, k: int) -> torch.Tensor:
    return F.avg_pool2d(x[None], k, 1, k // 2)[0]


def _shifted_crop(base: torch.Tensor, out_h: int, out_w: int, dx: float, dy: float) -> torch.Tensor:
    """Bilinear crop of base [C,Hb,Wb] whose top-left corner sits at (dx, dy) (sub-pixel)."""
    _, hb, wb = base.shape
    ys = torch.arange(out_h, dtype=torch.float32) + dy
    xs = torch.arange(out_w, dtype=torch.float32) + dx
    gy = 2 * ys / (hb - 1) - 1
    gx = 2 * xs / (wb - 1) - 1
    grid = torch.stack(torch.meshgrid(gx, gy, indexing="xy"), dim=-1)[None]
    return F.grid_sample(base[None], grid, mode="bilinear", padding_mode="border", align_corners=True)[0]


def synth_video(t: int, h: int = 476, w: int = 854, seed: int = 1, vx: float = 4.2, vy: float = 2.1) -> torch.Tensor:
    """Globally translating texture: frame k is the h x w window of a box-filtered random texture whose corner is at
    (vx*k, vy*k) px, plus N(0, 0.01^2) noise, clipped to [0,1].  Returns [T,3,h,w] float32."""
    g = torch.Generator().manual_seed(seed)
    margin_x, margin_y = int(vx * t) + 8, int(vy * t) + 8
    base = _box(torch.rand(3, h + margin_y, w + margin_x, generator=g), 5)
    base = (base - base.min()) / (base.max() - base.min())
    frames = []
    for k in range(t):
        f = _shifted_crop(base, h, w, vx * k, vy * k) + 0.01 * torch.randn(3, h, w, generator=g)
        frames.append(f.clamp_(0, 1))
    return torch.stack(frames)


def synth_features(t: int, c: int, ph: int = 67, pw: int = 121, seed: int = 0, vx: float = 0.6, vy: float = 0.3,
                   noise: float = 0.02, smooth: int = 3) -> torch.Tensor:
    """Feature-level analogue: box-filtered random field translating by (vx, vy) cells per frame + noise.
    Gives dense anchors (cos-sim along the true track stays high).  Returns [T,C,ph,pw] float32."""
    g = torch.Generator().manual_seed(seed)
    mx, my = int(vx * t) + 4, int(vy * t) + 4
    base = _box(torch.randn(c, ph + my, pw + mx, generator=g), smooth)
    out = []
    for k in range(t):
        out.append(_shifted_crop(base, ph, pw, vx * k, vy * k) + noise * torch.randn(c, ph, pw, generator=g))
    return torch.stack(out)


def synth_head_weights(seed: int = 3, benign: bool = True) -> Dict[str, torch.Tensor]:
    """TrackerHead.cnn_refiner state dict (models/networks/tracker_head.py:54-58 key names).
    benign: kernels with a positive mean (sum well away from 0, like a trained smoothing/sharpening refiner);
    otherwise the reference's default kaiming-uniform range, where W/sum(W) can be large."""
    g = torch.Generator().manual_seed(seed)

    def u(*shape, bound):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    shift1, shift2 = (0.25, 0.05) if benign else (0.0, 0.0)
    return {
        "cnn_refiner.0.weight": u(16, 1, 3, 3, bound=1 / 3) + shift1,
        "cnn_refiner.0.bias": u(16, bound=1 / 3),
        "cnn_refiner.2.weight": u(1, 16, 3, 3, bound=1 / 12) + shift2,
        "cnn_refiner.2.bias": u(1, bound=1 / 12),
    }


def synth_delta_dino_weights(c: int, seed: int = 4) -> Dict[str, torch.Tensor]:
    """DeltaDINO state dict (models/networks/delta_dino.py key names layers.{0,4,8,12}/{1,5,9,13}/{3,7,11}.filt).
    The reference zero-initialises the last conv (residual == 0); here it is re-randomised and BN running stats are
    made non-trivial so the whole CNN is exercised."""
    g = torch.Generator().manual_seed(seed)
    chans = [3, 64, 128, 256, c]
    sd: Dict[str, torch.Tensor] = {}
    for li in range(4):
        cin, cout = chans[li], chans[li + 1]
        ci, bi = 4 * li, 4 * li + 1
        bound = (1.0 / (cin * 25)) ** 0.5
        sd[f"layers.{ci}.weight"] = (torch.rand(cout, cin, 5, 5, generator=g) * 2 - 1) * bound * 1.7
        sd[f"layers.{ci}.bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
        sd[f"layers.{bi}.weight"] = (0.05 if li == 3 else 1.0) * (1.0 + 0.2 * torch.randn(cout, generator=g))
        sd[f"layers.{bi}.bias"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"layers.{bi}.running_mean"] = 0.1 * torch.randn(cout, generator=g)
        sd[f"layers.{bi}.running_var"] = 0.5 + torch.rand(cout, generator=g)
        sd[f"layers.{bi}.num_batches_tracked"] = torch.tensor(100)
        if li < 3:
            a = torch.tensor([1.0, 3.0, 3.0, 1.0])
            sd[f"layers.{4 * li + 3}.filt"] = (a[:, None] * a[None, :] / 64.0)[None, None].repeat(cout, 1, 1, 1)
    return sd


def grid_queries(nx: int, ny: int, h: int = 476, w: int = 854, t: int = 0, margin: float = 60.0) -> torch.Tensor:
    """nx*ny query points (x, y, t) on a regular grid inside the frame (x in [60, W-61], y in [60, H-61])."""
    xs = torch.linspace(margin, w - 1 - margin, nx)
    ys = torch.linspace(margin, h - 1 - margin, ny)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([xx.reshape(-1), yy.reshape(-1), torch.full((nx * ny,), float(t))], dim=1)


VIT_CONFIGS = {  # models/extractor.py:183-222
    "dinov2_vits14": dict(dim=384, depth=12, heads=6),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16),
}


def make_vit_weights(model_name: str, seed: int = 2, pos_grid: int = 37, patch: int = 14,
                     layerscale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with upstream's parameter names (no checkpoint exists in this environment).
    `layerscale` is the mean of the LayerScale gammas: 1.0 is the hub models' init_values; with untrained attention
    (near-uniform averaging) that makes every token collapse onto a common vector, unlike trained DINOv2 features,
    so the benchmark uses a small value that keeps the residual stream dominated by the patch content."""
    cfg = VIT_CONFIGS[model_name]
    d, depth = cfg["dim"], cfg["depth"]
    g = torch.Generator().manual_seed(seed)

    def tn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd = {
        "cls_token": tn(1, 1, d, std=1e-6 * 1e4),
        "pos_embed": tn(1, 1 + pos_grid * pos_grid, d),
        "patch_embed.proj.weight": tn(d, 3, patch, patch, std=0.05),
        "patch_embed.proj.bias": tn(d),
    }
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + tn(d, std=0.1)
        sd[p + "norm1.bias"] = tn(d, std=0.05)
        sd[p + "attn.qkv.weight"] = tn(3 * d, d, std=0.04)
        sd[p + "attn.qkv.bias"] = tn(3 * d)
        sd[p + "attn.proj.weight"] = tn(d, d, std=0.04)
        sd[p + "attn.proj.bias"] = tn(d)
        sd[p + "ls1.gamma"] = layerscale * (1.0 + tn(d, std=0.1))  # upstream hub models: init_values=1.0
        sd[p + "norm2.weight"] = 1.0 + tn(d, std=0.1)
        sd[p + "norm2.bias"] = tn(d, std=0.05)
        sd[p + "mlp.fc1.weight"] = tn(4 * d, d, std=0.04)
        sd[p + "mlp.fc1.bias"] = tn(4 * d)
        sd[p + "mlp.fc2.weight"] = tn(d, 4 * d, std=0.03)
        sd[p + "mlp.fc2.bias"] = tn(d)
        sd[p + "ls2.gamma"] = layerscale * (1.0 + tn(d, std=0.1))
    return sd


def make_outlier_vit_weights(scale_fc2: float = 300.0, model_name: str = "dinov2_vits14") -> Dict[str, torch.Tensor]:
    """ViT-S weights with the features of a TRAINED DINOv2 that the seeded initialisation lacks (used by
    tests/test_gpu_p1.py and scripts/e2e_error.py): 'massive activations' (a few tokens whose residual stream carries 1e3-scale
    values in a few channels, planted through the position encoding of 5 grid cells and the CLS token), LayerNorm gains up to
    8, LayerScale up to 3, a block with sharper attention, and one block whose MLP output is large (fc2 x scale_fc2)."""
    sd = {k: v.clone() for k, v in make_vit_weights(model_name, seed=12, layerscale=0.5).items()}
    g = torch.Generator().manual_seed(13)
    d = VIT_CONFIGS[model_name]["dim"]
    pe = sd["pos_embed"]  # [1, 1 + 37*37, d]
    for cell in (0, 1, 400, 401 + 37, 1369):
        ch = torch.randint(0, d, (3,), generator=g)
        pe[0, cell, ch] += torch.tensor([1500.0, -900.0, 600.0])
    for i in range(VIT_CONFIGS[model_name]["depth"]):
        sd[f"blocks.{i}.norm1.weight"] *= 1.0 + 7.0 * torch.rand(d, generator=g) ** 4
        sd[f"blocks.{i}.ls2.gamma"] *= 1.0 + 5.0 * torch.rand(d, generator=g) ** 4
    sd["blocks.2.attn.qkv.weight"][:2 * d] *= 1.5   # sharper attention in one block (scores to +-100 binades)
    sd["blocks.3.mlp.fc2.weight"] *= scale_fc2
    return sd

