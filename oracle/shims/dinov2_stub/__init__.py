"""TEST INFRASTRUCTURE ONLY -- a minimal module with the API surface of facebookresearch/dinov2's
`DinoVisionTransformer` that the reference's VitExtractor touches (models/extractor.py:23-150):

    model.patch_embed.proj (Conv2d whose .stride VitExtractor overwrites, :51), model.pos_embed, model.cls_token,
    model.interpolate_pos_encoding (replaced by VitExtractor._fix_pos_enc, :53), model.blocks[i] (forward hooks, :97-105)
    with .attn (hooked), .attn.qkv (hooked), .attn.attn_drop (hooked), and model(x) -> prepare_tokens_with_masks ->
    self.interpolate_pos_encoding(x, w, h) with upstream's argument convention (w := H, h := W).

Upstream (un-vendored, network-only: torch.hub.load('facebookresearch/dinov2', ...)) is RESTATED here from its
published definition (vision_transformer.py / layers/{attention,block,layer_scale,mlp,patch_embed}.py): pre-norm
blocks with LayerScale, qkv bias, MHSA scale d_head^-0.5 applied to q, erf-GELU MLP ratio 4, LayerNorm eps 1e-6,
no register tokens, img_size 518 (37 x 37 position grid), `block_chunks=0`.  What this stub buys: the reference's OWN
VitExtractor / get_dino_features_video code (hooks, stride surgery, position-encoding interpolation with the +0.1
fudge and swapped w/h, layer selection, CLS removal, rearrange) runs un-modified around it, so rows a1-a3 of
SURVEY.md section 8 are pinned by the reference itself; the block arithmetic inside stays "restated upstream".

Parameter names equal upstream's, so dino_tracker_amd.synth.make_vit_weights() loads with load_state_dict().
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

CONFIGS = {
    "dinov2_vits14": dict(dim=384, depth=12, heads=6),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16),
}


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.patch_size = (patch, patch)
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)
        self.norm = nn.Identity()

    def forward(self, x):
        _, _, H, W = x.shape
        assert H % self.patch_size[0] == 0, f"Input image height {H} is not a multiple of patch height {self.patch_size[0]}"
        assert W % self.patch_size[1] == 0, f"Input image width {W} is not a multiple of patch width: {self.patch_size[1]}"
        x = self.proj(x)
        return self.norm(x.flatten(2).transpose(1, 2))


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1.0):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim, bias=True)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        attn = self.attn_drop(attn)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * dim, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.ls1 = LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim)
        self.ls2 = LayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class DinoVisionTransformer(nn.Module):
    def __init__(self, model_name, patch=14, img_size=518):
        super().__init__()
        cfg = CONFIGS[model_name]
        dim = cfg["dim"]
        self.patch_size = patch
        self.patch_embed = PatchEmbed(patch, dim)
        n = (img_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.blocks = nn.ModuleList([Block(dim, cfg["heads"]) for _ in range(cfg["depth"])])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Identity()

    def interpolate_pos_encoding(self, x, w, h):  # upstream's own version; VitExtractor replaces it when stride != patch
        npatch = x.shape[1] - 1
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        class_pos_embed = self.pos_embed[:, 0]
        patch_pos_embed = self.pos_embed[:, 1:]
        dim = x.shape[-1]
        w0, h0 = w // self.patch_size + 0.1, h // self.patch_size + 0.1
        s = int(math.sqrt(N))
        patch_pos_embed = F.interpolate(patch_pos_embed.reshape(1, s, s, dim).permute(0, 3, 1, 2),
                                        scale_factor=(w0 / s, h0 / s), mode="bicubic")
        return torch.cat((class_pos_embed.unsqueeze(0), patch_pos_embed.permute(0, 2, 3, 1).view(1, -1, dim)), dim=1)

    def prepare_tokens_with_masks(self, x, masks=None):
        B, nc, w, h = x.shape
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x, w, h)

    def forward_features(self, x, masks=None):
        x = self.prepare_tokens_with_masks(x, masks)
        for blk in self.blocks:
            x = blk(x)
        x_norm = self.norm(x)
        return {"x_norm_clstoken": x_norm[:, 0], "x_norm_patchtokens": x_norm[:, 1:], "x_prenorm": x}

    def forward(self, x):
        return self.head(self.forward_features(x)["x_norm_clstoken"])


def build(model_name, state_dict):
    """The module torch.hub.load('facebookresearch/dinov2', model_name) would return, with the given weights (the
    final `norm` is not part of the hooked path and keeps its default initialisation)."""
    m = DinoVisionTransformer(model_name)
    missing, unexpected = m.load_state_dict(state_dict, strict=False)
    assert not unexpected and all(k.startswith("norm.") for k in missing), (missing, unexpected)
    return m.eval()
