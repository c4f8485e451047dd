"""VitExtractor -- the reference's models/extractor.py:16-274 API on the HIP ViT encoder (dtk_vit_forward).

The reference obtains the network through torch.hub (models/extractor.py:26), which needs the network.  Here the
weights are a plain state dict in upstream facebookresearch/dinov2 naming (`patch_embed.proj.weight`,
`blocks.{i}.attn.qkv.weight`, `blocks.{i}.ls1.gamma`, ...), taken from, in order:
  1. the `state_dict=` argument,
  2. the file named by $DTK_DINOV2_WEIGHTS (e.g. the official dinov2_vits14_pretrain.pth),
  3. a seeded random initialisation if `random_seed=` is given (synthetic benchmarks / parity tests).
Anything else raises: there is no silent fallback.
Facets: `tokens` (block outputs, models/extractor.py:137-150) is the hot path.  The qkv hook output of a block
(models/extractor.py:107-118) comes from the same device program (`qkv_out` of dtk_vit_forward: fp32 output of the
16-bit-operand GEMM), so the key / query / value getters (:224-267) are reshapes of it like in the reference; the
attention-map facet and the key self-similarity are small torch expressions over it (not on the hot path).

Operand type of the matrix units: IEEE fp16 by default (`operand_dtype="fp16"`; the residual stream, the statistics and
every accumulation are fp32) -- the same MFMA rate as bf16 with 8x less operand rounding.  fp16 ends at 65504: activations
beyond that SATURATE on the device and set an overflow word -- for every value of every frame (residual updates in the
LayerNorm that applies them; Q / K / V and the MLP hidden inside the epilogues of the GEMMs that store them).  What happens
then is `on_overflow`: "bf16" (default) re-encodes the call with bf16 operands (fp32's range, 8 mantissa bits), counts it in
`range_fallbacks`, warns once, and keeps the extractor on bf16 from then on (the out-of-range activations of a trained ViT
are systematic: the same few channels in every image); "raise" turns the word into a RuntimeError as rounds 3-4 did.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import VIT_ATTENTION_V2, VIT_GEMM_WS_V1, VIT_BF16, VIT_CHECK_RANGE, VIT_TILED_GEMMS, VitLayer, VitModel, check, lib
from .synth import VIT_CONFIGS, make_vit_weights

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class VitExtractor(nn.Module):
    BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY = "block", "attn", "patch_imd", "qkv"
    KEY_LIST = [BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY]

    def __init__(self, model_name, stride, device, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 random_seed: Optional[int] = None, operand_dtype: str = "fp16", check_range: bool = False,
                 on_overflow: str = "bf16"):
        super().__init__()
        if operand_dtype not in ("fp16", "bf16"):
            raise ValueError(f"operand_dtype {operand_dtype!r}: 'fp16' or 'bf16'")
        if on_overflow not in ("bf16", "raise"):
            raise ValueError(f"on_overflow {on_overflow!r}: 'bf16' (re-encode with bf16 operands) or 'raise'")
        self.operand_dtype = operand_dtype
        self.on_overflow = on_overflow
        self.range_fallbacks = 0        # encode() calls that left the fp16 range and were re-run with bf16 operands
        self.last_overflow = 0          # overflow word of the most recent check (bit 1 residual update, 2 Q/K/V, 4 MLP hidden)
        # What is checked for fp16 saturation (include/dtk.h: dtk_vit_model.overflow): every residual update of every token,
        # and every Q / K / V / MLP-hidden value of every frame (inside the GEMM epilogues) -- always.  check_range=True adds a
        # scan of the stored tensors (one extra pass per block): the cross-check of the tests.
        self.check_range = check_range
        self.attention_v2 = False       # the round-2/3 attention kernel instead of the one-wave-per-SIMD one (cross-check)
        self.gemm_ws_v1 = False         # the round 1-3 form of the K = 384 weight-stationary GEMMs (A / B measurement)
        self.frame_batch = 0            # frames per pass of the encoder; 0 = the library's default
        if model_name not in VIT_CONFIGS:
            raise NotImplementedError(f"{model_name}: the HIP encoder covers dinov2_vit{{s,b,l}}14 (d_head 64)")
        self.model_name, self.stride, self.device = model_name, stride, device
        self.cfg = VIT_CONFIGS[model_name]
        if state_dict is None and os.environ.get("DTK_DINOV2_WEIGHTS"):
            state_dict = torch.load(os.environ["DTK_DINOV2_WEIGHTS"], map_location="cpu")
        if state_dict is None and random_seed is not None:
            state_dict = make_vit_weights(model_name, seed=random_seed)
        if state_dict is None:
            raise RuntimeError("no DINOv2 weights: pass state_dict=, set $DTK_DINOV2_WEIGHTS to a checkpoint in "
                               "upstream naming, or ask for random_seed= explicitly (torch.hub needs the network)")
        self.n_layers = self.get_n_layers()
        self.tiled_gemms = False  # run every GEMM on the tiled kernel (dtk_vit_model.flags; cross-check in the tests)
        self._sd = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in state_dict.items()
                    if k.startswith(("cls_token", "pos_embed", "patch_embed.", "blocks."))}
        self._keep = []      # device tensors referenced by the C structs
        self._layers_of = {}  # operand type -> ctypes array of dtk_vit_layer (bf16 built on first need)
        self._pos_cache = {}
        self._layers = self._build_layers(self.operand_dtype)

    # ---- weights -> C structs -----------------------------------------------------------------------------------
    def _build_layers(self, operand_dtype):
        if operand_dtype in self._layers_of:
            return self._layers_of[operand_dtype]
        depth = self.cfg["depth"]
        arr = (VitLayer * depth)()
        sd = self._sd

        def f32(name):
            t = sd[name].contiguous()
            self._keep.append(t)
            return t.data_ptr()

        wdt = torch.float16 if operand_dtype == "fp16" else torch.bfloat16

        def b16(name):  # matrix weights in the operand type of the MFMA kernels
            w = sd[name]
            if wdt == torch.float16 and float(w.abs().max()) >= 65504.0:
                raise RuntimeError(f"{name}: |w| reaches the fp16 limit; use operand_dtype='bf16'")
            t = w.to(wdt).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        for i in range(depth):
            p = f"blocks.{i}."
            L = arr[i]
            L.ln1_w, L.ln1_b = f32(p + "norm1.weight"), f32(p + "norm1.bias")
            L.qkv_w, L.qkv_b = b16(p + "attn.qkv.weight"), f32(p + "attn.qkv.bias")
            L.proj_w, L.proj_b = b16(p + "attn.proj.weight"), f32(p + "attn.proj.bias")
            L.ls1 = f32(p + "ls1.gamma")
            L.ln2_w, L.ln2_b = f32(p + "norm2.weight"), f32(p + "norm2.bias")
            L.fc1_w, L.fc1_b = b16(p + "mlp.fc1.weight"), f32(p + "mlp.fc1.bias")
            L.fc2_w, L.fc2_b = b16(p + "mlp.fc2.weight"), f32(p + "mlp.fc2.bias")
            L.ls2 = f32(p + "ls2.gamma")
        self._layers_of[operand_dtype] = arr
        return arr

    def _pos_embed(self, ph: int, pw: int):
        """models/extractor.py:57-85 (`_fix_pos_enc`): bicubic, align_corners=False, scale_factor with the +0.1 fudge,
        recompute_scale_factor=False; frame independent, so computed once per grid size (torch, one-off)."""
        key = (ph, pw)
        if key not in self._pos_cache:
            pe = self._sd["pos_embed"]
            n = int(math.sqrt(pe.shape[1] - 1))
            d = pe.shape[-1]
            grid = pe[:, 1:].reshape(1, n, n, d).permute(0, 3, 1, 2)
            grid = F.interpolate(grid, scale_factor=((ph + 0.1) / n, (pw + 0.1) / n), mode="bicubic",
                                 align_corners=False, recompute_scale_factor=False)
            assert grid.shape[-2] == ph and grid.shape[-1] == pw
            pos = grid.permute(0, 2, 3, 1).reshape(ph * pw, d).contiguous()
            cls_pos = (self._sd["cls_token"].reshape(-1) + pe[0, 0]).contiguous()
            self._pos_cache[key] = (pos, cls_pos)
        return self._pos_cache[key]

    # ---- forward ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, frames: torch.Tensor, layer: Optional[int] = None, normalize: bool = True, want: str = "feat",
               defer_check: bool = False):
        """frames [n,3,H,W] fp32 -> `feat`: token-major [n, ph*pw, D] (CLS dropped) or `tokens`: [n, 1+ph*pw, D].
        normalize=True applies the ImageNet mean/std inside the patch-embedding kernel (utils.py:46,55).
        A saturated fp16 activation (see __init__) raises RuntimeError -- here, behind one stream synchronisation, or, with
        defer_check=True, in the caller's next `check_overflow()`: the call then returns with the kernels still in flight (what
        it hands to them stays alive in this object) and the overflow word keeps accumulating until it is read."""
        frames = frames.to(self.device, torch.float32).contiguous()
        n, _, H, W = frames.shape
        layer = self.n_layers - 1 if layer is None else layer
        if not -1 <= layer < self.n_layers:  # -1: patch embedding + position encoding only (no block)
            raise ValueError(f"layer {layer} out of range")
        patch = self.get_patch_size()
        ph, pw = 1 + (H - patch) // self.stride, 1 + (W - patch) // self.stride
        pos, cls_pos = self._pos_embed(ph, pw)
        if not hasattr(self, "_ms"):
            self._ms = {flag: torch.tensor((IMAGENET_MEAN + IMAGENET_STD) if flag else (0.0, 0.0, 0.0, 1.0, 1.0, 1.0),
                                           dtype=torch.float32, device=self.device) for flag in (True, False)}
            self._overflow = torch.zeros(1, dtype=torch.int32, device=self.device)
        ms, overflow = self._ms[bool(normalize)], self._overflow
        D = self.cfg["dim"]
        S = ph * pw + 1
        if want not in ("tokens", "feat", "qkv"):
            raise ValueError(want)

        def run(operand_dtype):
            flags = (VIT_TILED_GEMMS if self.tiled_gemms else 0) | (VIT_BF16 if operand_dtype == "bf16" else 0) | \
                (VIT_CHECK_RANGE if self.check_range else 0) | (VIT_ATTENTION_V2 if self.attention_v2 else 0) | \
                (VIT_GEMM_WS_V1 if self.gemm_ws_v1 else 0)
            m = VitModel(D, self.cfg["heads"], layer + 1, patch, self.stride, 1e-6, flags,
                         self._sd["patch_embed.proj.weight"].data_ptr(),
                         self._sd["patch_embed.proj.bias"].data_ptr(), cls_pos.data_ptr(), pos.data_ptr(), ms.data_ptr(),
                         ctypes.cast(self._build_layers(operand_dtype), ctypes.POINTER(VitLayer)), int(self.frame_batch),
                         overflow.data_ptr())
            ws_bytes = int(lib().dtk_vit_workspace_bytes(m, H, W, n))
            # the workspace (2.6 GB for 30 frames of 854 x 476) is kept between calls: handing it back to the caching allocator
            # and asking again costs a device allocation (~25 ms) whenever the block has been split or released in between
            ws = getattr(self, "_ws", None)
            if ws is None or ws.numel() < ws_bytes or ws.device != frames.device:   # (frames.device carries the resolved index)
                self._ws = ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            tokens = torch.empty((n, S, D), dtype=torch.float32, device=self.device) if want == "tokens" else None
            feat = torch.empty((n, ph * pw, D), dtype=torch.float32, device=self.device) if want == "feat" else None
            qkv = torch.empty((n, S, 3 * D), dtype=torch.float32, device=self.device) if want == "qkv" else None
            check(lib().dtk_vit_forward(m, ops._p(frames), n, H, W, ops._p(tokens), ops._p(feat), ops._p(qkv), ops._p(ws),
                                        ws_bytes, ops._stream()))
            # (`ms`, `ws`, the position encoding and the overflow word are members: they outlive the launches)
            return {"tokens": tokens, "feat": feat, "qkv": qkv}[want]

        out = run(self.operand_dtype)
        if not defer_check and self.check_overflow(heal=True):
            out = run(self.operand_dtype)          # the fp16 pass saturated: the same call on bf16 operands (sticky, see check_overflow)
            self.check_overflow()                  # (bf16 sets no bits; a non-finite residual update would still raise)
        return out

    def check_overflow(self, heal: bool = False) -> bool:
        """Reads (one stream synchronisation) and clears the overflow word of the encode() calls since the last check.
        A non-zero word means those calls' features are not trustworthy.  With on_overflow="bf16" and fp16 operands the
        extractor switches to bf16 operands for the rest of its life and counts the event; `heal=True` (encode's own call)
        then returns True so that the caller re-runs the pass.  Otherwise -- "raise", a deferred check whose features were
        already handed on, or bf16 operands (only non-finite values can set the word there) -- RuntimeError."""
        if not hasattr(self, "_overflow"):
            return False
        self.last_overflow = int(self._overflow.item())
        if not self.last_overflow:
            return False
        self._overflow.zero_()
        what = " and ".join(n for b, n in ((1, "a residual update"), (2, "Q / K / V"), (4, "the MLP hidden")) if self.last_overflow & b)
        if self.operand_dtype == "fp16" and self.on_overflow == "bf16":
            self.operand_dtype = "bf16"
            self.range_fallbacks += 1
            import warnings
            warnings.warn(f"VitExtractor: {what} left the fp16 range (saturated at 65504); this extractor now runs on bf16 "
                          "operands (range_fallbacks counts the re-encoded calls)", RuntimeWarning, stacklevel=3)
            if heal:
                return True
            raise RuntimeError(f"dtk_vit_forward: {what} left the fp16 range in a call whose check was deferred: its features "
                               "are saturated and must be discarded -- encode the video again (this extractor is on bf16 now)")
        raise RuntimeError(f"dtk_vit_forward: {what} left the fp16 range (saturated at 65504) or is not finite; "
                           "construct the extractor with operand_dtype='bf16'")

    def release_workspace(self):
        """Frees the encoder's activation workspace (2.6 GB for 30 frames of 854 x 476): preprocessing is done."""
        torch.cuda.current_stream().synchronize()
        self._ws = None

    def get_feature_from_input(self, input_img, layers: List[int]):  # models/extractor.py:137-150
        """input_img [B,3,H,W] ALREADY ImageNet-normalised (as the reference's caller does) -> mean over `layers` of
        the block outputs, [B, 1+ph*pw, D]."""
        outs = [self.encode(input_img, layer=l, normalize=False, want="tokens") for l in layers]
        return torch.stack(outs).mean(dim=0)

    # ---- the other facets (models/extractor.py:152-274) -----------------------------------------------------------
    class _PerLayer:
        """What the reference's hook lists are to their callers: indexable by layer, computed on demand (the reference
        runs the whole network once and records every layer; here layer l costs one run of blocks 0..l)."""

        def __init__(self, n, fn):
            self._n, self._fn, self._cache = n, fn, {}

        def __len__(self):
            return self._n

        def __getitem__(self, layer):
            layer = range(self._n)[layer]
            if layer not in self._cache:
                self._cache[layer] = self._fn(layer)
            return self._cache[layer]

        def __iter__(self):
            return (self[i] for i in range(self._n))

    def get_qkv_feature_from_input(self, input_img):
        """models/extractor.py:152-158: per layer, the output of blocks[l].attn.qkv, [B, 1+ph*pw, 3D]."""
        return VitExtractor._PerLayer(self.n_layers, lambda l: self.encode(input_img, layer=l, normalize=False, want="qkv"))

    def get_attn_feature_from_input(self, input_img):
        """models/extractor.py:160-166: per layer, softmax(q k^T / sqrt(d_head)) [B, heads, S, S] (the input of attn_drop).
        Not a hot path: S x S per head is materialised with torch on the device."""
        heads = self.cfg["heads"]

        def attn(l):
            qkv = self.encode(input_img, layer=l, normalize=False, want="qkv")
            b, s, _ = qkv.shape
            q, k, _ = qkv.reshape(b, s, 3, heads, -1).permute(2, 0, 3, 1, 4)
            return torch.softmax((q * q.shape[-1] ** -0.5) @ k.transpose(-2, -1), dim=-1)

        return VitExtractor._PerLayer(self.n_layers, attn)

    def _from_qkv(self, qkv, input_img_shape, which):
        b = input_img_shape[0]
        return qkv.reshape(b, self.get_patch_num(input_img_shape), 3, self.cfg["dim"])[:, :, which, :]

    def get_queries_from_qkv(self, qkv, input_img_shape):
        return self._from_qkv(qkv, input_img_shape, 0)

    def get_keys_from_qkv(self, qkv, input_img_shape):
        return self._from_qkv(qkv, input_img_shape, 1)

    def get_values_from_qkv(self, qkv, input_img_shape):
        return self._from_qkv(qkv, input_img_shape, 2)

    def _facet_from_input(self, input_img, layers, which):
        qkv = self.get_qkv_feature_from_input(input_img)
        return torch.cat([self._from_qkv(qkv[l], input_img.shape, which) for l in layers], dim=2)

    def get_queries_from_input(self, input_img, layers):
        return self._facet_from_input(input_img, layers, 0)

    def get_keys_from_input(self, input_img, layers):
        return self._facet_from_input(input_img, layers, 1)

    def get_values_from_input(self, input_img, layers):
        return self._facet_from_input(input_img, layers, 2)

    def get_keys_self_sim_from_input(self, input_img, layer_num):
        """models/extractor.py:269-274 with attn_cosine_sim (:8-13)."""
        keys = self.get_keys_from_input(input_img, layers=[layer_num])
        h, t, d = keys.shape
        x = keys.transpose(0, 1).reshape(t, h * d)[None]
        norm = x.norm(dim=2, keepdim=True)
        return (x @ x.permute(0, 2, 1)) / torch.clamp(norm @ norm.permute(0, 2, 1), min=1e-8)

    # ---- static model facts (models/extractor.py:168-222) ----------------------------------------------------------
    def get_patch_size(self):
        return 8 if "8" in self.model_name else 14

    def get_width_patch_num(self, input_img_shape):
        b, c, h, w = input_img_shape
        return 1 + (w - self.get_patch_size()) // self.stride

    def get_height_patch_num(self, input_img_shape):
        b, c, h, w = input_img_shape
        return 1 + (h - self.get_patch_size()) // self.stride

    def get_patch_num(self, input_img_shape):
        return 1 + self.get_height_patch_num(input_img_shape) * self.get_width_patch_num(input_img_shape)

    def get_n_layers(self):
        return self.cfg["depth"]

    def get_head_num(self):
        return self.cfg["heads"]

    @staticmethod
    def get_embedding_dim(model_name=None):
        # the reference defines this name twice (instance method shadowed by the static one, :207-222)
        return VIT_CONFIGS[model_name]["dim"]
