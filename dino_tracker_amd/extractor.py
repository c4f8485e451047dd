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
every accumulation are fp32) -- the same MFMA rate as bf16 with 8x less operand rounding.

Precision (round 6).  `precision="fast"` (default) rounds every matrix operand to ONE 16-bit number: the features are
1.3e-4 (benchmark weights) .. 2.1e-3 (DINOv2-like outlier statistics) from the fp32 reference, relative.  `precision="split"`
runs every product of every block on split operands (x = hi + lo, three MFMAs per product, csrc/vit_split.h): fp32-grade
features (2e-7 .. 4e-6 relative, what the fp32 reference itself is from float64) at ~3x the matrix work.  A list of block
indices escalates those blocks only.  `precision="auto"` MEASURES which one is needed: the first frames of the first call
are encoded both ways and the extractor stays on split operands if the fast features differ by more than `auto_tol`
(relative, default 2.5e-4: the level below which the end-to-end positions were measured within 1e-3 px); the measurement is
kept in `calibration`.  `precision="auto-blocks"` goes one step further when the fast pass is NOT good enough: it measures what
each block contributes when it alone runs on single 16-bit operands (one pass of the calibration frames per block, all other
blocks split), keeps on the fast kernels the largest set of blocks whose contributions -- added in quadrature, then MEASURED
together -- stay below `block_margin * auto_tol`, and escalates the rest.  `precision_report()` returns what the last call ran on.

fp16 ends at 65504: activations beyond that SATURATE on the device and set an overflow word -- for every value of every frame
(residual updates in the LayerNorm that applies them; Q / K / V and the MLP hidden inside the epilogues of the GEMMs that store
them).  What happens then is `on_overflow`: "split-bf16" (default since round 6) re-encodes the call on SPLIT bf16 operands --
fp32's range AND 16 significant bits, i.e. closer to the reference than the plain fp16 pass it replaces --, counts it in
`range_fallbacks`, warns once and stays there (the out-of-range activations of a trained ViT are systematic: the same few
channels in every image); "bf16" is the round-5 behaviour (plain bf16 operands: 8 significant bits, the fast kernels; opt-in
because it is 7-10x further from the reference than fp16); "raise" turns the word into a RuntimeError.
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
from ._lib import VIT_ATTENTION_V2, VIT_ATTENTION_V4, VIT_GEMM_WIDE_V1, VIT_GEMM_WS_V1, VIT_NO_LN_FUSION, VIT_BF16, VIT_CHECK_RANGE, VIT_TILED_GEMMS, VitLayer, VitModel, check, lib
from .synth import VIT_CONFIGS, make_vit_weights

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class VitExtractor(nn.Module):
    BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY = "block", "attn", "patch_imd", "qkv"
    KEY_LIST = [BLOCK_KEY, ATTN_KEY, PATCH_IMD_KEY, QKV_KEY]

    def __init__(self, model_name, stride, device, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 random_seed: Optional[int] = None, operand_dtype: str = "fp16", check_range: bool = False,
                 on_overflow: str = "split-bf16", precision=None, auto_tol: float = 2.5e-4, calibration_frames: int = 2,
                 block_margin: float = 0.8):
        super().__init__()
        if operand_dtype not in ("fp16", "bf16"):
            raise ValueError(f"operand_dtype {operand_dtype!r}: 'fp16' or 'bf16'")
        if on_overflow not in ("split-bf16", "bf16", "raise"):
            raise ValueError(f"on_overflow {on_overflow!r}: 'split-bf16' (re-encode on split bf16 operands), 'bf16' (plain bf16 "
                             "operands) or 'raise'")
        self.operand_dtype = operand_dtype
        self.on_overflow = on_overflow
        self.range_fallbacks = 0        # encode() calls that left the fp16 range and were re-run on (split) bf16 operands
        self.last_overflow = 0          # overflow word of the most recent check (bit 1 residual update, 2 Q/K/V, 4 MLP hidden)
        self._deferred_pending = 0      # encode(defer_check=True) calls whose overflow bits have not been read yet
        # What is checked for fp16 saturation (include/dtk.h: dtk_vit_model.overflow): every residual update of every token,
        # and every Q / K / V / MLP-hidden value of every frame (inside the GEMM epilogues) -- always.  check_range=True adds a
        # scan of the stored tensors (one extra pass per block): the cross-check of the tests.
        self.check_range = check_range
        self.auto_tol, self.calibration_frames = float(auto_tol), int(calibration_frames)
        self.block_margin = float(block_margin)   # precision="auto-blocks": the mixed pass must measure <= block_margin * auto_tol
        self.calibration = None         # precision="auto": {"frames", "layer", "rel_fast_vs_split", "tol", "chosen"} once measured
        self.attention_v2 = False       # the round-2/3 attention kernel instead of the one-wave-per-SIMD one (cross-check)
        self.attention_v4 = os.environ.get("DTK_VIT_ATTENTION_V4", "0") == "1"   # rounds 4-5: 64 queries per wave (round 6: 128; A / B)
        self.gemm_ws_v1 = False         # the round 1-3 form of the K = 384 weight-stationary GEMMs (A / B measurement)
        self.no_ln_fusion = os.environ.get("DTK_VIT_NO_LN_FUSION", "0") == "1"   # D = 384: LayerNorm-1 of the next block as its own launch (A / B)
        self.gemm_wide_v1 = os.environ.get("DTK_VIT_GEMM_WIDE_V1", "0") == "1"   # the LDS-DMA GEMMs with round 5's direct-store epilogues (A / B; bit-identical)
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
        self._wcache = {}    # 16-bit weight tensors / split planes by (name, operand type[, "split"])
        self._layers_of = {}  # (operand type, split blocks) -> ctypes array of dtk_vit_layer (built on first need)
        self._pos_cache = {}
        # (precision=None: $DTK_VIT_PRECISION -- how the reference's UN-MODIFIED preprocessing/save_dino_embed_video.py, which
        #  passes no such argument, is run on split operands -- or "fast")
        self.set_precision(precision if precision is not None else os.environ.get("DTK_VIT_PRECISION", "fast"))
        self._layers = self._build_layers(self.operand_dtype, self.split_blocks)

    def set_precision(self, precision):
        """"fast" | "split" | "auto" | "auto-blocks" | an iterable of block indices that run on split operands (see the module docstring)."""
        depth = VIT_CONFIGS[self.model_name]["depth"]
        if isinstance(precision, str):
            if precision not in ("fast", "split", "auto", "auto-blocks"):
                raise ValueError(f"precision {precision!r}: 'fast', 'split', 'auto', 'auto-blocks' or a list of block indices")
            self.precision = precision
            self.split_blocks = frozenset(range(depth)) if precision == "split" else frozenset()
        else:
            blocks = frozenset(int(b) for b in precision)
            if any(not 0 <= b < depth for b in blocks):
                raise ValueError(f"precision: block indices must be in [0, {depth})")
            self.precision, self.split_blocks = "blocks", blocks
        self.calibration = None   # a new request is measured afresh ("auto" / "auto-blocks": on the next encode; split_blocks restarts empty)

    def precision_report(self):
        """What the most recent encode() ran on, for callers that must know the feature-error class of their results:
        relative feature error vs the fp32 reference, measured on this repo's weights (docs/PARITY.md): split fp16 2e-7 .. 4e-6,
        split bf16 ~2e-5, fp16 1.3e-4 .. 2.1e-3, bf16 1e-3 .. 1.9e-2."""
        depth = self.cfg["depth"]
        full = len(self.split_blocks) == depth
        cls = {("fp16", True): "fp32-grade (<= 4e-6)", ("bf16", True): "2^-16 (~2e-5)", ("fp16", False): "2^-12 (1.3e-4 .. 2.1e-3)",
               ("bf16", False): "2^-9 (1e-3 .. 1.9e-2)"}[(self.operand_dtype, full)]
        return {"operand_dtype": self.operand_dtype, "precision": self.precision, "split_blocks": sorted(self.split_blocks),
                "range_fallbacks": self.range_fallbacks, "last_overflow": self.last_overflow, "calibration": self.calibration,
                "feature_error_class": cls if (full or not self.split_blocks) else f"mixed ({len(self.split_blocks)}/{depth} blocks split)"}

    # ---- weights -> C structs -----------------------------------------------------------------------------------
    def _build_layers(self, operand_dtype, split_blocks=frozenset()):
        key = (operand_dtype, frozenset(split_blocks))
        if key in self._layers_of:
            return self._layers_of[key]
        depth = self.cfg["depth"]
        arr = (VitLayer * depth)()
        sd = self._sd

        def f32(name):
            t = sd[name].contiguous()
            self._keep.append(t)
            return t.data_ptr()

        wdt = torch.float16 if operand_dtype == "fp16" else torch.bfloat16

        def b16(name):  # matrix weights in the operand type of the MFMA kernels
            ck = (name, operand_dtype)
            if ck not in self._wcache:
                w = sd[name]
                if wdt == torch.float16 and float(w.abs().max()) >= 65504.0:
                    raise RuntimeError(f"{name}: |w| reaches the fp16 limit; use operand_dtype='bf16'")
                self._wcache[ck] = w.to(wdt).contiguous()
            return self._wcache[ck].data_ptr()

        def split16(names):
            """hi / lo planes of scale * W for the four matrices of a split block (include/dtk.h: dtk_vit_layer.qkv_w_lo).  One
            power-of-two scale per block: 2^8 for fp16 (the lo halves of |w| ~ 1e-2 stay normal numbers), less if a weight
            would leave the range; 1 for bf16."""
            ck = (names[0], operand_dtype, "split")
            if ck not in self._wcache:
                scale = 1.0
                if wdt == torch.float16:
                    amax = max(float(sd[n].abs().max()) for n in names)
                    scale = 256.0
                    while scale > 1.0 and amax * scale >= 32768.0:
                        scale /= 2.0
                    if amax * scale >= 65504.0:
                        raise RuntimeError(f"{names[0]}: |w| reaches the fp16 limit; use operand_dtype='bf16'")
                planes = []
                for n in names:
                    w = sd[n] * scale
                    hi = w.to(wdt)
                    lo = (w - hi.float()).to(wdt)
                    planes.append((hi.contiguous(), lo.contiguous()))
                self._wcache[ck] = (scale, planes)
            return self._wcache[ck]

        for i in range(depth):
            p = f"blocks.{i}."
            L = arr[i]
            L.ln1_w, L.ln1_b = f32(p + "norm1.weight"), f32(p + "norm1.bias")
            L.qkv_b, L.proj_b = f32(p + "attn.qkv.bias"), f32(p + "attn.proj.bias")
            L.ls1 = f32(p + "ls1.gamma")
            L.ln2_w, L.ln2_b = f32(p + "norm2.weight"), f32(p + "norm2.bias")
            L.fc1_b, L.fc2_b = f32(p + "mlp.fc1.bias"), f32(p + "mlp.fc2.bias")
            L.ls2 = f32(p + "ls2.gamma")
            names = [p + "attn.qkv.weight", p + "attn.proj.weight", p + "mlp.fc1.weight", p + "mlp.fc2.weight"]
            if i in split_blocks:
                scale, planes = split16(names)
                (L.qkv_w, L.qkv_w_lo), (L.proj_w, L.proj_w_lo), (L.fc1_w, L.fc1_w_lo), (L.fc2_w, L.fc2_w_lo) = \
                    [(h.data_ptr(), l.data_ptr()) for h, l in planes]
                L.w_scale = scale
            else:
                L.qkv_w, L.proj_w, L.fc1_w, L.fc2_w = [b16(n) for n in names]
                L.qkv_w_lo = L.proj_w_lo = L.fc1_w_lo = L.fc2_w_lo = None
                L.w_scale = 1.0
        self._layers_of[key] = arr
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
               defer_check: bool = False, taps: Optional[List[int]] = None):
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
        if want not in ("tokens", "feat", "qkv", "taps"):
            raise ValueError(want)
        if want == "taps":   # the mean of the block outputs of `taps` in ONE pass (dtk_vit_model.tap_out); layer = the deepest of them
            taps = sorted({range(self.n_layers)[int(t)] for t in taps})
            layer = taps[-1]

        def run(operand_dtype, split_blocks, frames=frames, n=n):
            flags = (VIT_TILED_GEMMS if self.tiled_gemms else 0) | (VIT_BF16 if operand_dtype == "bf16" else 0) | \
                (VIT_CHECK_RANGE if self.check_range else 0) | (VIT_ATTENTION_V2 if self.attention_v2 else 0) | (VIT_ATTENTION_V4 if self.attention_v4 else 0) | \
                (VIT_GEMM_WS_V1 if self.gemm_ws_v1 else 0) | (VIT_GEMM_WIDE_V1 if self.gemm_wide_v1 else 0) | \
                (VIT_NO_LN_FUSION if self.no_ln_fusion else 0)
            m = VitModel(D, self.cfg["heads"], layer + 1, patch, self.stride, 1e-6, flags,
                         self._sd["patch_embed.proj.weight"].data_ptr(),
                         self._sd["patch_embed.proj.bias"].data_ptr(), cls_pos.data_ptr(), pos.data_ptr(), ms.data_ptr(),
                         ctypes.cast(self._build_layers(operand_dtype, split_blocks), ctypes.POINTER(VitLayer)),
                         int(self.frame_batch), overflow.data_ptr(), None, 0, 0.0)
            tap_out = None
            if want == "taps":
                tap_out = torch.zeros((n, S, D), dtype=torch.float32, device=self.device)
                m.tap_out, m.tap_mask, m.tap_scale = tap_out.data_ptr(), sum(1 << t for t in taps), 1.0 / len(taps)
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
            return {"tokens": tokens, "feat": feat, "qkv": qkv, "taps": tap_out}[want]

        if self.precision in ("auto", "auto-blocks") and self.calibration is None and layer >= 0:
            # measure what the fast operands cost on THIS network and THESE frames: the first frames both ways (a split pass of two
            # 854 x 476 frames is ~15 ms), fast-vs-split relative difference = the fast path's distance from fp32-grade features
            if self._deferred_pending:
                raise RuntimeError(f"VitExtractor(precision={self.precision!r}): the calibrating call cannot follow un-checked deferred calls")
            k = min(n, max(1, self.calibration_frames))
            depth_all = frozenset(range(self.cfg["depth"]))
            cal = frames[:k].contiguous()
            ref = run(self.operand_dtype, depth_all, cal, k)
            if self.check_overflow(heal=True):     # even the calibration pass saturated: now on split bf16 / bf16 (sticky)
                ref = run(self.operand_dtype, depth_all if self.on_overflow == "split-bf16" else self.split_blocks, cal, k)
                self.check_overflow()
            ref64, ref_norm = ref.double(), ref.double().norm()

            def measure(split_blocks):
                """relative distance of a pass with `split_blocks` escalated from the all-split pass (inf if it saturates fp16)"""
                got = run(self.operand_dtype, frozenset(split_blocks), cal, k)
                ovf = int(self._overflow.item()) if self.operand_dtype == "fp16" else 0
                if ovf:
                    self._overflow.zero_()
                    return float("inf"), True
                return float(((got.double() - ref64).norm() / ref_norm).item()), False

            rel, fast_overflow = measure(frozenset())
            chosen = "split" if not (rel <= self.auto_tol) else "fast"
            self.calibration = {"frames": k, "layer": layer, "rel_fast_vs_split": rel, "tol": self.auto_tol, "chosen": chosen,
                                "fast_pass_saturated": bool(fast_overflow)}
            if len(self.split_blocks):             # (a range fallback above already put every block on split operands)
                self.split_blocks = depth_all
            elif chosen == "split" and self.precision == "auto-blocks":
                self.split_blocks = self._pick_blocks(measure, layer)
            elif chosen == "split":
                self.split_blocks = depth_all

        out = run(self.operand_dtype, self.split_blocks)
        if defer_check:
            self._deferred_pending += 1
        elif self.check_overflow(heal=True):
            out = run(self.operand_dtype, self.split_blocks)   # the fp16 pass saturated: the same call on (split) bf16 operands (sticky)
            self.check_overflow()                  # (bf16 sets no bits; a non-finite residual update would still raise)
        return out

    def _pick_blocks(self, measure, layer):
        """precision="auto-blocks" after the all-fast pass measured beyond auto_tol: which blocks stay on single 16-bit operands.
        e_b = the distance from the all-split features with block b ALONE fast (what b's operand roundings cost at the output,
        amplification by the later blocks included); roundings of different blocks are independent, so a fast set F is predicted
        at sqrt(sum e_b^2) -- F grows from the cheapest block while that stays below block_margin * auto_tol, and is then
        MEASURED as one pass; while the measurement is above the bound the costliest member leaves.  Blocks beyond `layer` do not
        run in the calibration and are escalated.  Fills calibration["blocks"]."""
        used = list(range(layer + 1))
        bound = self.block_margin * self.auto_tol
        alone = {}
        for b in used:
            alone[b], sat = measure(frozenset(used) - {b})
        order = sorted(used, key=lambda b: alone[b])
        fast, acc = [], 0.0
        for b in order:
            if not (acc + alone[b] ** 2) ** 0.5 <= bound:
                break
            fast.append(b)
            acc += alone[b] ** 2
        tried = []
        while fast:
            rel, sat = measure(frozenset(used) - frozenset(fast))
            tried.append({"fast_blocks": sorted(fast), "predicted": sum(alone[b] ** 2 for b in fast) ** 0.5, "measured": rel})
            if rel <= bound:
                break
            fast.pop()                              # the costliest member (fast is in ascending order of e_b)
        split = frozenset(used) - frozenset(fast)
        self.calibration.update({"chosen": "blocks" if fast else "split", "blocks": {
            "bound": bound, "alone": [alone[b] for b in used], "fast_blocks": sorted(fast), "split_blocks": sorted(split),
            "passes": tried, "measured": tried[-1]["measured"] if fast else 0.0}})
        # (blocks beyond `layer` were not measured: escalated, so that a later call to a deeper layer does not run them on trust)
        return (split | frozenset(range(layer + 1, self.cfg["depth"]))) if fast else frozenset(range(self.cfg["depth"]))

    def check_overflow(self, heal: bool = False) -> bool:
        """Reads (one stream synchronisation) and clears the overflow word of the encode() calls since the last check.
        A non-zero word means those calls' features are not trustworthy.  With fp16 operands and on_overflow="split-bf16" /
        "bf16" the extractor switches to split bf16 / plain bf16 operands for the rest of its life and counts the event;
        `heal=True` (encode's own call) then returns True so that the caller re-runs the pass.  Otherwise -- "raise", a deferred
        check whose features were already handed on (also when encode's own check finds bits while deferred calls are still
        un-checked: the bits cannot be attributed), or bf16 operands (only non-finite values can set the word
        there) -- RuntimeError."""
        if not hasattr(self, "_overflow"):
            return False
        self.last_overflow = int(self._overflow.item())
        pending, self._deferred_pending = self._deferred_pending, 0
        if not self.last_overflow:
            return False
        self._overflow.zero_()
        what = " and ".join(n for b, n in ((1, "a residual update"), (2, "Q / K / V"), (4, "the MLP hidden")) if self.last_overflow & b)
        if self.operand_dtype == "fp16" and self.on_overflow in ("split-bf16", "bf16"):
            self.operand_dtype = "bf16"
            to = "bf16"
            if self.on_overflow == "split-bf16":
                self.split_blocks = frozenset(range(self.cfg["depth"]))
                to = "SPLIT bf16 (hi + lo: 16 significant bits, fp32's range)"
            self.range_fallbacks += 1
            import warnings
            warnings.warn(f"VitExtractor: {what} left the fp16 range (saturated at 65504); this extractor now runs on {to} "
                          "operands (range_fallbacks counts the re-encoded calls; precision_report() describes the result)",
                          RuntimeWarning, stacklevel=3)
            if heal and not pending:
                return True
            raise RuntimeError(f"dtk_vit_forward: {what} left the fp16 range in a call whose check was deferred: its features "
                               "are saturated and must be discarded -- encode the video again (this extractor is on bf16 now)")
        raise RuntimeError(f"dtk_vit_forward: {what} left the fp16 range (saturated at 65504) or is not finite; "
                           "construct the extractor with operand_dtype='bf16' (precision='split' keeps 16 significant bits there)")

    def release_workspace(self):
        """Frees the encoder's activation workspace (2.6 GB for 30 frames of 854 x 476): preprocessing is done."""
        torch.cuda.current_stream().synchronize()
        self._ws = None

    def get_feature_from_input(self, input_img, layers: List[int]):  # models/extractor.py:137-150
        """input_img [B,3,H,W] ALREADY ImageNet-normalised (as the reference's caller does) -> mean over `layers` of
        the block outputs, [B, 1+ph*pw, D]."""
        if len(set(layers)) == len(layers) and len(layers) > 1:
            # ONE pass up to the deepest requested block, the outputs of the requested blocks averaged on the way (round 6; rounds 1-5
            # re-ran blocks 0..l once per requested layer: O(L^2) for the reference's multi-layer mean)
            return self.encode(input_img, normalize=False, want="taps", taps=list(layers))
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
