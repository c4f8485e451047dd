"""Where the fast ViT path's feature error comes from (round 6; CPU only, float64): the oracle's block with fp16 / bf16 / split
(hi + lo) ROUNDING inserted at exactly the tensors the device stores in 16 bits -- LN output, the four weight matrices, Q
(pre-scaled), K, V, P, the attention output, the MLP hidden, the pending residual update.  It reproduces the device's measured
feature error (benchmark weights 1.3e-4, LayerScale 1.0 5.8e-4, outlier weights 2.1e-3: tests/test_gpu_p1.py) without a GPU, and
shows (a) that no single tensor dominates -- so the escalation has to be the whole block (csrc/vit_split.h) --, (b) what hi + lo
operands give: 2e-7 .. 4e-6, the level of the fp32 oracle itself.
    python scripts/p1_error_budget.py [outlier | <layerscale>] [layer]          e.g.  outlier 5   |   0.1 11   |   1.0 11"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dino_tracker_amd import synth  # noqa: E402
from oracle import ref_algo as A  # noqa: E402

ALL = ["ln1", "wqkv", "q", "k", "v", "p", "ao", "wproj", "delta", "ln2", "wfc1", "hid", "wfc2"]


def round16(x, dt):
    return x.to(dt).to(x.dtype)


def split16(x, dt):
    hi = x.to(dt).to(x.dtype)
    return hi + (x - hi).to(dt).to(x.dtype)


def block(x, sd, i, heads, R):
    """oracle.ref_algo.vit_block in the device's operation order (exp2-domain softmax, P rounded before the P V product, the
    row sum taken from the unrounded P), with R = {tensor name: True} choosing what is rounded, R["fn"] how."""
    p = f"blocks.{i}."
    b, s, d = x.shape
    fn = R.get("fn")
    rr = (lambda t, key: fn(t) if R.get(key) and (R.get("blocks") is None or i in R["blocks"]) else t)
    y = rr(F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps=1e-6), "ln1")
    qkv = F.linear(y, rr(sd[p + "attn.qkv.weight"], "wqkv"), sd[p + "attn.qkv.bias"]).reshape(b, s, 3, heads, d // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    q = rr(q * (0.125 * 1.4426950408889634), "q")
    k, v = rr(k, "k"), rr(v, "v")
    sc = q @ k.transpose(-1, -2)
    m = sc.amax(dim=-1, keepdim=True)
    e = torch.exp2(sc - m)
    a = (rr(e, "p") @ v) / e.sum(-1, keepdim=True)
    a = rr(a.transpose(1, 2).reshape(b, s, d), "ao")
    x = x + rr(sd[p + "ls1.gamma"] * F.linear(a, rr(sd[p + "attn.proj.weight"], "wproj"), sd[p + "attn.proj.bias"]), "delta")
    y = rr(F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps=1e-6), "ln2")
    hdn = rr(F.gelu(F.linear(y, rr(sd[p + "mlp.fc1.weight"], "wfc1"), sd[p + "mlp.fc1.bias"])), "hid")
    y = F.linear(hdn, rr(sd[p + "mlp.fc2.weight"], "wfc2"), sd[p + "mlp.fc2.bias"])
    return x + rr(sd[p + "ls2.gamma"] * y, "delta")


def run(frame, sd, layer, R):
    sd = {k: v.double() for k, v in sd.items()}
    frame = frame.double()
    m = torch.tensor(A.IMAGENET_MEAN, dtype=frame.dtype).view(1, 3, 1, 1)
    s = torch.tensor(A.IMAGENET_STD, dtype=frame.dtype).view(1, 3, 1, 1)
    ph, pw = A.feature_grid(frame.shape[-2], frame.shape[-1], 14, 7)
    tok = F.conv2d((frame - m) / s, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=7).flatten(2).transpose(1, 2)
    tok = torch.cat([sd["cls_token"].expand(tok.shape[0], -1, -1), tok], dim=1) + A.vit_pos_embed(sd, ph, pw)
    for i in range(layer + 1):
        tok = block(tok, sd, i, 6, R)
    return tok[0, 1:]


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    which = sys.argv[1] if len(sys.argv) > 1 else "outlier"
    layer = int(sys.argv[2]) if len(sys.argv) > 2 else (5 if which == "outlier" else 11)
    sd = synth.make_outlier_vit_weights(300.0) if which == "outlier" else synth.make_vit_weights("dinov2_vits14", seed=2, layerscale=float(which))
    video = synth.synth_video(1, 238, 322, seed=80)
    ref = run(video, sd, layer, {})
    every = {k: True for k in ALL}
    f16 = lambda t: round16(t, torch.float16)    # noqa: E731
    print(f"weights {which}, block {layer}, 238 x 322 frame; relative Frobenius error of the block output vs float64")
    print(f"  every tensor fp16      : {rel(run(video, sd, layer, dict(every, fn=f16)), ref):.3e}   <- the fast path (measured on MI355X: see tests/test_gpu_p1.py)")
    print(f"  every tensor bf16      : {rel(run(video, sd, layer, dict(every, fn=lambda t: round16(t, torch.bfloat16))), ref):.3e}")
    print(f"  every tensor fp16 hi+lo: {rel(run(video, sd, layer, dict(every, fn=lambda t: split16(t, torch.float16))), ref):.3e}   <- precision='split'")
    print(f"  every tensor bf16 hi+lo: {rel(run(video, sd, layer, dict(every, fn=lambda t: split16(t, torch.bfloat16))), ref):.3e}   <- the range escalation")
    ref32 = A.vit_tokens(video, sd, "dinov2_vits14", layer=layer).permute(1, 2, 0).reshape(-1, 384).double()
    print(f"  the fp32 oracle itself : {rel(ref32, ref):.3e}")
    for key in ALL:
        print(f"  only {key:6s} fp16       : {rel(run(video, sd, layer, {key: True, 'fn': f16}), ref):.3e}")
    for bi in range(layer + 1):
        r = rel(run(video, sd, layer, dict(every, fn=f16, blocks=[b for b in range(layer + 1) if b != bi])), ref)
        print(f"  every tensor fp16, block {bi:2d} exact: {r:.3e}")
