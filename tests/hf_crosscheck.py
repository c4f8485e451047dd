"""Standalone cross-check of oracle.ref_algo.vit_block against transformers.Dinov2Model layers (random weights)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers as tr
from oracle import ref_algo as A
cfg = tr.Dinov2Config(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, mlp_ratio=4, image_size=28,
                      patch_size=14, layerscale_value=1.0)
hf = tr.Dinov2Model(cfg).eval()
g = torch.Generator().manual_seed(0)
sd = {}
for i, layer in enumerate(hf.encoder.layer):
    p = f"blocks.{i}."
    for prm in layer.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * 0.1 + (1.0 if prm.ndim == 1 and prm.shape[0] == 64 else 0.0) * 0
    att = layer.attention.attention
    sd[p + "norm1.weight"], sd[p + "norm1.bias"] = layer.norm1.weight.data, layer.norm1.bias.data
    sd[p + "attn.qkv.weight"] = torch.cat([att.query.weight.data, att.key.weight.data, att.value.weight.data])
    sd[p + "attn.qkv.bias"] = torch.cat([att.query.bias.data, att.key.bias.data, att.value.bias.data])
    sd[p + "attn.proj.weight"] = layer.attention.output.dense.weight.data
    sd[p + "attn.proj.bias"] = layer.attention.output.dense.bias.data
    sd[p + "ls1.gamma"] = layer.layer_scale1.lambda1.data
    sd[p + "norm2.weight"], sd[p + "norm2.bias"] = layer.norm2.weight.data, layer.norm2.bias.data
    sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = layer.mlp.fc1.weight.data, layer.mlp.fc1.bias.data
    sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = layer.mlp.fc2.weight.data, layer.mlp.fc2.bias.data
    sd[p + "ls2.gamma"] = layer.layer_scale2.lambda1.data
x = torch.randn(1, 37, 64, generator=g)
with torch.no_grad():
    y = x
    for layer in hf.encoder.layer:
        y = layer(y)
        y = y[0] if isinstance(y, tuple) else y
    z = x
    for i in range(2):
        z = A.vit_block(z, sd, i, heads=4)
err = (y - z).abs().max().item()
print("max err", err)
assert err < 1e-4
