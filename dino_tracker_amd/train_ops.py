"""Differentiable (train-mode) arithmetic of the tracker -- SURVEY.md section 8(f) N1, first cut.

The inference kernels of libdtk have no backward; per-video test-time training (dino_tracker.py:392-448) needs gradients
with respect to the Delta-DINO CNN and the tracker head through

    frames -> Delta-DINO (train-mode BatchNorm) -> align to the ViT grid -> + DINO        models/tracker.py:113-129
    -> bilinear sampling of the source embeddings                                          models/tracker.py:96-111
    -> cosine correlation map of every source with ITS target frame -> ReLU                models/tracker.py:158-173
    -> normalised 3x3 convs -> softmax -> disk-masked soft arg-max                         tracker_head.py:68-121

This module states that arithmetic on torch tensors so that autograd supplies the backward (rocBLAS / MIOpen kernels on
the device); everything that runs without gradients inside a training step (the cycle-consistency filter, the mutual
nearest-neighbour search) stays on the hand-written kernels.  Differences in mechanism from the reference:
  * one correlation map per source (grouped by target frame: one [B_f, C] x [C, HW] product per frame) instead of
    B x n maps of which B are kept (tracker.py:159-160) -- n times less work forward AND backward;
  * the CNN -> ViT grid alignment is two constant interpolation matrices (row, column) applied as matrix products instead
    of a grid_sample over a B x h x w x 2 grid (models/utils.py:30-44): deterministic backward, no atomics;
  * source sampling gathers the four corners of the source's own frame (the reference interpolates trilinearly over a
    frame stack at an integer frame coordinate, utils.py:97-100).
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn.functional as F

EPS = 1e-8  # models/tracker.py:14, conv_norm.py:35, tracker_head.py:86


# ---- Delta-DINO ----------------------------------------------------------------------------------------------------------
class _BlurPool(torch.autograd.Function):
    """csrc/train.hip: one gather kernel forward, one (the adjoint, also a gather) backward."""

    @staticmethod
    def forward(ctx, x):
        from . import ops
        ctx.hw = x.shape[-2:]
        return ops.blurpool_forward(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        return ops.blurpool_backward(dy.contiguous(), *ctx.hw)


def blurpool(x: torch.Tensor, filt: torch.Tensor, stride: int = 2) -> torch.Tensor:
    """antialiased_cnns.BlurPool(filt_size 4, reflect): pad (left 1, right 2, top 1, bottom 2), depthwise 4 x 4 binomial
    filter at `stride` (the same outer([1,3,3,1]) / 64 for every channel: the module's `filt` buffer).  On the device: the
    hand-written kernels; on host tensors (CPU parity tests) two separable passes over strided views."""
    if x.is_cuda and stride == 2 and filt.shape[-1] == 4:
        return _BlurPool.apply(x)
    a = filt[0, 0].sum(dim=1)
    a = (a / a.sum()).tolist()  # the 1-D factor, [1, 3, 3, 1] / 8
    xp = F.pad(x, (1, 2, 1, 2), mode="reflect")
    hp, wp = xp.shape[-2:]
    ho, wo = (hp - 4) // stride + 1, (wp - 4) // stride + 1
    rows = sum(a[k] * xp[:, :, k:k + stride * (ho - 1) + 1:stride, :] for k in range(4))
    return sum(a[k] * rows[:, :, :, k:k + stride * (wo - 1) + 1:stride] for k in range(4))


_WORKSPACE = {}
# A captured iteration (trainer.GraphedIteration) has the addresses of these buffers baked into its launches: once a graph exists a
# buffer that is outgrown (a batch with more frames) is kept alive next to its replacement instead of being freed.
RETAIN_REPLACED_WORKSPACES = False
_RETIRED = []


def _workspace(name: str, numel: int, like: torch.Tensor) -> torch.Tensor:
    """Persistent scratch (per device and name), grown on demand: the unfolded operands of the convolutions are the largest
    tensors of a training step (5 GB each at full size) -- kept out of the caching allocator, whose blocks the reference's
    loop releases every iteration (torch.cuda.empty_cache(), dino_tracker.py:406)."""
    key = (name, like.device, like.dtype)
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < numel:
        if buf is not None and RETAIN_REPLACED_WORKSPACES:
            _RETIRED.append(buf)
        _WORKSPACE[key] = buf = torch.empty(numel, dtype=like.dtype, device=like.device)
    return buf[:numel]


def release_scratch() -> None:
    """Drop the persistent scratch (Tracker.eval() calls this: inference needs none of it)."""
    _WORKSPACE.clear()
    _RETIRED.clear()
    _PACKED.clear()


class _ConvGemm(torch.autograd.Function):
    """Stride-1 convolution = unfold + matrix product, with the unfolded input RECOMPUTED in the backward (into the same
    persistent scratch) instead of being saved: forward  Y_n = W [Cout, K] . cols_n [K, L];  backward  dW = sum_n dY_n .
    cols_n^T,  dcols_n = W^T . dY_n -> col2im -> adjoint of the padding."""

    @staticmethod
    def forward(ctx, x, weight, padding, dilation, padding_mode):
        n, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        xp = F.pad(x, (padding,) * 4, mode=padding_mode) if padding and padding_mode != "zeros" else x
        zpad = padding if padding_mode == "zeros" else 0
        ho, wo = xp.shape[-2] + 2 * zpad - dilation * (kh - 1), xp.shape[-1] + 2 * zpad - dilation * (kw - 1)
        k = cin * kh * kw
        cols = _workspace("cols", n * k * ho * wo, x).view(n, k, ho * wo)
        torch.ops.aten.im2col.out(xp, [kh, kw], [dilation, dilation], [zpad, zpad], [1, 1], out=cols)
        y = torch.matmul(weight.reshape(cout, k), cols).view(n, cout, ho, wo)
        ctx.save_for_backward(x, weight)
        ctx.conf = (padding, dilation, padding_mode, zpad, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        padding, dilation, padding_mode, zpad, ho, wo = ctx.conf
        n, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        k = cin * kh * kw
        dy = dy.reshape(n, cout, ho * wo)
        dw = dx = None
        xp = F.pad(x, (padding,) * 4, mode=padding_mode) if padding and padding_mode != "zeros" else x
        if ctx.needs_input_grad[1]:
            cols = _workspace("cols", n * k * ho * wo, x).view(n, k, ho * wo)
            torch.ops.aten.im2col.out(xp, [kh, kw], [dilation, dilation], [zpad, zpad], [1, 1], out=cols)
            dw = torch.matmul(dy, cols.transpose(1, 2)).sum(dim=0).view_as(weight)
        if ctx.needs_input_grad[0]:
            dcols = _workspace("dcols", n * k * ho * wo, x).view(n, k, ho * wo)
            torch.matmul(weight.reshape(cout, k).t(), dy, out=dcols)
            dxp = F.fold(dcols, xp.shape[-2:], (kh, kw), dilation=dilation, padding=zpad)
            if xp is x:
                dx = dxp
            elif padding_mode == "reflect":
                dx = torch.ops.aten.reflection_pad2d_backward(dxp, x, [padding] * 4)
            else:
                raise NotImplementedError(padding_mode)
        return dx, dw, None, None, None


_SCALARS = {}


def _scalar(value: float, device) -> torch.Tensor:
    key = (float(value), str(device))
    if key not in _SCALARS:
        _SCALARS[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return _SCALARS[key]


W_SCALE = 256.0  # weights carry 2^8 into the fp16 split (their lo halves stay normal numbers), like the inference kernels


def _pow2_scale(t: torch.Tensor) -> torch.Tensor:
    """Device scalar 2^e with max |t| * 2^e in [2^9, 2^10]: the operand scale of a gradient tensor for the fp16 split (its
    entries down to 2^-24 of the largest keep normal halves).  No host synchronisation."""
    if t.is_cuda and t.dtype == torch.float32 and t.numel() > 0:
        # csrc/train.hip dtk_pow2_scale: the library's multi-block reduction is not safe inside a captured graph on this stack
        from . import ops
        from ._lib import check, lib
        t = t.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=t.device)      # [scale, scratch word]
        check(lib().dtk_pow2_scale(t.data_ptr(), t.numel(), out.data_ptr(), out.data_ptr() + 4, ops._stream()))
        return out[:1]
    amax = torch.linalg.vector_norm(t, ord=float("inf")).clamp_min(1e-30)  # one reduction pass, no |t| temporary
    return torch.exp2(torch.floor(10.0 - torch.log2(amax))).reshape(1)


# "fp16": the implicit-GEMM convolutions of the training step use plain fp16 operands (the hi halves only: one matrix-core
# product per term instead of three; 2^-11 relative operand rounding, accumulation in fp32) -- BASELINE.json's config 5 names
# fp16.  Default "split": fp32-grade (hi + lo), what the parity tests against the reference's float32 run use.
CONV_OPERANDS = os.environ.get("DTK_TRAIN_CONV_OPERANDS", "split")
USE_IMPLICIT_CONVS = True  # 5 x 5 layers with Cin % 16 == 0: forward / data gradient on the implicit-GEMM kernel (no im2col)


def _implicit_conv(x, weight, dilation, zero_pad, flip_transpose, border, fold, scale):
    """conv5x5_split_kernel between the two layout kernels (csrc/delta_dino.hip, dtk_conv_split_*): x [n, c, h, w] fp32 ->
    [n, c_out, h, w] fp32.  `border` / `fold` / `scale`: the data-gradient form (see _ConvMfma.backward)."""
    from . import ops
    n, cin, h, w = x.shape
    cout = weight.shape[1] if flip_transpose else weight.shape[0]
    he, we = h + 2 * border, w + 2 * border
    wh, wl = ops.conv_split_pack(weight, flip_transpose)
    planes = _workspace("conv_in_planes", n * he * we * cin, x)      # fp32-sized slot: hi plane | lo plane (2 bytes each)
    hi = planes.view(torch.float16)[:n * he * we * cin]
    lo = planes.view(torch.float16)[n * he * we * cin:2 * n * he * we * cin]
    ops.conv_split_input(x, hi, lo, border, scale)
    y_nhwc = _workspace("conv_out_nhwc", n * he * we * cout, x)
    ops.conv_split_run(hi, lo, wh, wl, y_nhwc, n, he, we, cin, cout, dilation, zero_pad, CONV_OPERANDS == "fp16")
    out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    ops.conv_split_output(y_nhwc, out, border, fold, scale)
    return out


class _ConvMfma(torch.autograd.Function):
    """Stride-1 'same' convolution (k x k, reflect or zero padding, dilation) on the hand-written kernels of csrc/train.hip
    (round 3): im2col -> split-fp16 MFMA GEMM, forward, weight gradient and data gradient; fp32-grade (2^-22 relative per
    operand).  The unfolded operands live in persistent scratch and are recomputed in the backward.
        forward      Y[cout][l]  = sum_k W[cout][k]  cols[l][k]
        weight grad  dW[cout][k] = sum_{frames, l} dY[cout][l] colsT[k][l]      (reduction split over workgroups, atomic adds)
        data grad    dcols[k][l] = sum_c Wt[k][c] dYt[l][c]  -> col2im (with the adjoint of the reflect padding)"""

    @staticmethod
    def forward(ctx, x, weight, padding, dilation, padding_mode):
        from . import ops
        n, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        assert kh == kw and 2 * padding == dilation * (kh - 1) and padding_mode in ("reflect", "zeros")
        k = cin * kh * kw
        kp = (k + 31) // 32 * 32
        L = h * w
        x = x.contiguous()
        ctx.implicit = (USE_IMPLICIT_CONVS and kh == 5 and cin % 16 == 0 and cout % 16 == 0 and dilation in (1, 2)
                        and min(h, w) > 4 * dilation + 1)
        if ctx.implicit:
            y = _implicit_conv(x, weight, dilation, padding_mode == "zeros", False, 0, False, None)
            ctx.save_for_backward(x, weight)
            ctx.conf = (padding, dilation, padding_mode, kp)
            return y
        cols = _workspace("cols", n * L * kp, x)
        ops.im2col(x, cols, kh, padding, dilation, padding_mode == "reflect", 0, kp)
        wp = torch.zeros(cout, kp, dtype=torch.float32, device=x.device)
        wp[:, :k] = weight.detach().reshape(cout, k)
        y = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
        ops.gemm_nt(wp, cols, y, cout, L, kp, kp, kp, L, batch=n, stride_a=0, stride_b=L * kp, stride_c=cout * L,
                    scale_a=_scalar(W_SCALE, x.device))
        ctx.save_for_backward(x, weight)
        ctx.conf = (padding, dilation, padding_mode, kp)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, weight = ctx.saved_tensors
        padding, dilation, padding_mode, kp = ctx.conf
        n, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        k = cin * kh * kw
        L = h * w
        dev = x.device
        dy = dy.contiguous()
        s_dy = _pow2_scale(dy)
        dw = dx = None
        if ctx.needs_input_grad[1] and ctx.implicit:
            dw = torch.empty(weight.shape, dtype=torch.float32, device=dev)
            ops.conv_wgrad_split(x, dy, dw, dilation, padding_mode == "reflect", s_dy, CONV_OPERANDS == "fp16")
        elif ctx.needs_input_grad[1]:
            lp = (L + 31) // 32 * 32
            colst = _workspace("cols", n * kp * lp, x)
            ops.im2col(x, colst, kh, padding, dilation, padding_mode == "reflect", 1, kp, lp)
            if lp != L:  # rows of dY padded to the same (aligned) length, zeros behind
                dyp = torch.zeros(n, cout, lp, dtype=torch.float32, device=dev)
                dyp[:, :, :L] = dy.reshape(n, cout, L)
            else:
                dyp = dy
            dwp = torch.zeros(cout, kp, dtype=torch.float32, device=dev)
            tiles = ((cout + 127) // 128) * ((kp + 127) // 128)
            split = max(1, min(lp // 2048, 2048 // max(1, tiles * n), 65535 // n))
            ops.gemm_nt(dyp, colst, dwp, cout, kp, lp, lp, lp, kp, batch=n, stride_a=cout * lp, stride_b=kp * lp, stride_c=0,
                        split_k=split, accumulate=2, scale_a=s_dy)
            dw = dwp[:, :k].reshape(weight.shape)
        if ctx.needs_input_grad[0] and ctx.implicit:
            # dX = zero-padded convolution of dY with the flipped, transposed kernel; under reflect padding over the PADDED
            # domain (dY inside a ring of 2 d zeros), folded back onto the image by the adjoint of the padding
            if padding_mode == "reflect":
                dx = _implicit_conv(dy, weight, dilation, True, True, padding, True, s_dy)
            else:
                dx = _implicit_conv(dy, weight, dilation, True, True, 0, False, s_dy)
        elif ctx.needs_input_grad[0]:
            wt = torch.zeros(kp, cout, dtype=torch.float32, device=dev)
            wt[:k] = weight.detach().reshape(cout, k).t()
            dyt = _workspace("dyt", n * L * cout, x)
            ops.transpose_f32(dy, dyt, cout, L, batch=n)
            dcols = _workspace("dcols", n * kp * L, x)
            ops.gemm_nt(wt, dyt, dcols, kp, L, cout, cout, cout, L, batch=n, stride_a=0, stride_b=L * cout, stride_c=kp * L,
                        scale_a=_scalar(W_SCALE, dev), scale_b=s_dy)
            dx = torch.empty_like(x)
            ops.col2im(dcols, dx, kh, padding, dilation, padding_mode == "reflect", kp)
        return dx, dw, None, None, None


PAD_THIN_INPUTS = os.environ.get("DTK_TRAIN_LAYER1", "implicit") == "implicit"   # "im2col": round 3's form of the first layer
USE_MFMA_CONVS = True  # device tensors: csrc/train.hip (_ConvMfma); False: round 2's unfold + library GEMM (_ConvGemm)


def conv2d_gemm(x: torch.Tensor, weight: torch.Tensor, bias, padding: int, dilation: int = 1, padding_mode: str = "zeros"):
    """Stride-1 convolution as ONE matrix product over the unfolded input ([Cout, Cin k k] x [Cin k k, H W] per frame):
    forward and both backward products run on the BLAS GEMM kernels, im2col / col2im on plain copy kernels.  (The
    convolution library's per-call solver selection costs the host ~0.3 s per forward call and seconds per backward call on
    this stack -- 40 s per training iteration, against 0.05 s of kernels.)  On the device the unfolded operands live in
    persistent scratch and are recomputed in the backward (_ConvGemm); host tensors take the autograd-traced form."""
    if x.is_cuda:
        kh = weight.shape[-1]
        same = weight.shape[-2] == kh and 2 * padding == dilation * (kh - 1) and padding_mode in ("reflect", "zeros")
        fn = _ConvMfma if (USE_MFMA_CONVS and same and weight.shape[0] % 4 == 0) else _ConvGemm
        if (fn is _ConvMfma and PAD_THIN_INPUTS and USE_IMPLICIT_CONVS and kh == 5 and weight.shape[1] < 16 and weight.shape[0] % 16 == 0
                and dilation in (1, 2) and min(x.shape[-2:]) > 4 * dilation + 1 and not x.requires_grad):
            # Layer 1 (3 -> 64 channels, the full-resolution frames): the implicit-GEMM kernels take input channels in sixteens, so the
            # frames and the weights are padded with zero channels -- forward and weight gradient then run on conv5x5_split /
            # conv_wgrad_split like layers 2-4 instead of im2col (1.25 GB of unfolded pixels written and read twice per iteration) + a
            # GEMM with K = 75.  The padded channels multiply zeros; the weight gradient's slice is F.pad's backward.
            extra = 16 - weight.shape[1]
            x = F.pad(x, (0, 0, 0, 0, 0, extra))
            weight = F.pad(weight, (0, 0, 0, 0, 0, extra))
        y = fn.apply(x, weight, padding, dilation, padding_mode)
        return y if bias is None else y + bias[None, :, None, None]
    n, cin, h, w = x.shape
    cout, _, kh, kw = weight.shape
    if padding and padding_mode != "zeros":
        x = F.pad(x, (padding,) * 4, mode=padding_mode)
        padding = 0
    cols = F.unfold(x, (kh, kw), dilation=dilation, padding=padding)      # [n, Cin kh kw, L]
    ho = x.shape[-2] + 2 * padding - dilation * (kh - 1)
    wo = x.shape[-1] + 2 * padding - dilation * (kw - 1)
    y = torch.matmul(weight.reshape(cout, cin * kh * kw), cols)            # [n, Cout, L]
    if bias is not None:
        y = y + bias[None, :, None]
    return y.reshape(n, cout, ho, wo)


def align_matrix(n_vit: int, n_cnn: int, vit_stride: int, vit_patch: int, cnn_stride: int, device, dtype=torch.float32):
    """[n_vit, n_cnn] interpolation matrix of models/utils.py:30-44 along one axis: ViT centre i sits at pixel
    vit_stride * i + vit_patch / 2, CNN cell j at pixel cnn_stride * j; grid_sample(align_corners=True, border) with
    g = -1 - 1/c_br + 2 px / c_br reads the CNN axis at (px - 0.5) / cnn_stride, clamped to [0, n_cnn - 1]."""
    # float32 in the reference's own order of operations (models/utils.py:30-36, then grid_sample's unnormalisation), so the
    # interpolation weights carry the same rounding as its grid
    px = torch.arange(n_vit, dtype=torch.float32) * vit_stride + vit_patch / 2.0
    c_br = (n_cnn - 1) * cnn_stride
    g = -1.0 - (1.0 / c_br) + (2.0 * px / c_br)
    pos = (((g + 1.0) / 2.0) * (n_cnn - 1)).clamp(0, n_cnn - 1)
    lo = pos.floor().clamp(max=n_cnn - 1)
    hi = (lo + 1).clamp(max=n_cnn - 1)
    w_hi = pos - lo
    m = torch.zeros(n_vit, n_cnn, dtype=torch.float32)
    rows = torch.arange(n_vit)
    m[rows, lo.long()] += 1.0 - w_hi
    m[rows, hi.long()] += w_hi
    return m.to(device=device, dtype=dtype)


def align_tables(n_vit: int, n_cnn: int, vit_stride: int, vit_patch: int, cnn_stride: int):
    """The two-tap form of `align_matrix` along one axis, for the device kernels (dtk_resample2d_*): destination index i
    reads source cells lo[i] and min(lo[i] + 1, n_cnn - 1) with weights 1 - whi[i], whi[i] -- same float32 arithmetic as
    align_matrix -- and, for the backward gather, ranges[4][n_cnn]: start / end of the destination indices whose LO is a,
    then start / end of those whose HI is a (lo and hi are non-decreasing)."""
    px = torch.arange(n_vit, dtype=torch.float32) * vit_stride + vit_patch / 2.0
    c_br = (n_cnn - 1) * cnn_stride
    g = -1.0 - (1.0 / c_br) + (2.0 * px / c_br)
    pos = (((g + 1.0) / 2.0) * (n_cnn - 1)).clamp(0, n_cnn - 1)
    lo = pos.floor().clamp(max=n_cnn - 1)
    whi = pos - lo
    lo = lo.long()
    hi = (lo + 1).clamp(max=n_cnn - 1)
    assert bool((lo[1:] >= lo[:-1]).all()) and bool((hi[1:] >= hi[:-1]).all())
    a = torch.arange(n_cnn)
    ranges = torch.stack([torch.searchsorted(lo, a, right=False), torch.searchsorted(lo, a, right=True),
                          torch.searchsorted(hi, a, right=False), torch.searchsorted(hi, a, right=True)])
    return lo.to(torch.int32), whi.to(torch.float32), ranges.to(torch.int32).contiguous()


_ALIGN_TABLES = {}


def _align_tables_on(device, key):
    k = (str(device),) + key
    if k not in _ALIGN_TABLES:
        _ALIGN_TABLES[k] = tuple(t.to(device) for t in align_tables(*key))
    return _ALIGN_TABLES[k]


class _AlignResample(torch.autograd.Function):
    """CNN -> ViT grid alignment on csrc/train.hip (dtk_resample2d_forward / _backward): 4 reads per output, the backward a
    gather.  (Round 2: two constant interpolation matrices as library GEMMs over 60 x 107 planes, 14 ms per iteration.)"""

    @staticmethod
    def forward(ctx, cnn, h, w, vit_stride, vit_patch, cnn_stride):
        from . import ops
        cnn = cnn.contiguous()
        n, c, hc, wc = cnn.shape
        ylo, ywhi, yr = _align_tables_on(cnn.device, (h, hc, vit_stride, vit_patch, cnn_stride))
        xlo, xwhi, xr = _align_tables_on(cnn.device, (w, wc, vit_stride, vit_patch, cnn_stride))
        out = torch.empty(n, c, h, w, dtype=torch.float32, device=cnn.device)
        ops.resample2d_forward(cnn, out, ylo, ywhi, xlo, xwhi)
        ctx.tables = (yr, ywhi, xr, xwhi, hc, wc)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import ops
        yr, ywhi, xr, xwhi, hc, wc = ctx.tables
        dout = dout.contiguous()
        n, c = dout.shape[:2]
        dcnn = torch.empty(n, c, hc, wc, dtype=torch.float32, device=dout.device)
        ops.resample2d_backward(dout, dcnn, yr, ywhi, xr, xwhi)
        return dcnn, None, None, None, None, None


def align_cnn_to_vit(cnn: torch.Tensor, h: int, w: int, vit_stride: int, vit_patch: int, cnn_stride: int) -> torch.Tensor:
    """models/utils.py:7-45: [n, C, hc, wc] -> [n, C, h, w].  Device float32 tensors: the two-tap resampling kernels; host
    tensors (and other dtypes): the same weights as two matrix products."""
    if cnn.is_cuda and cnn.dtype == torch.float32 and cnn.shape[0] * cnn.shape[1] <= 65535:
        return _AlignResample.apply(cnn, h, w, vit_stride, vit_patch, cnn_stride)
    my = align_matrix(h, cnn.shape[-2], vit_stride, vit_patch, cnn_stride, cnn.device, cnn.dtype)
    mx = align_matrix(w, cnn.shape[-1], vit_stride, vit_patch, cnn_stride, cnn.device, cnn.dtype)
    return torch.matmul(my, torch.matmul(cnn, mx.t()))


class _BatchNormTrain(torch.autograd.Function):
    """Train-mode BatchNorm2d (+ the ReLU behind it) on the hand-written kernels (csrc/train.hip): statistics by pairwise
    merging of exact small-group moments.  The library BatchNorm behind torch.nn.BatchNorm2d loses the variance of channels
    whose mean is large against their spread (5e-3 relative output error measured on Delta-DINO's second layer)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, pre_bias, running_mean, running_var, momentum, eps, relu):
        from . import ops
        x = x.contiguous()
        pb = None if pre_bias is None else pre_bias.detach().contiguous()
        y, mean, rstd = ops.batchnorm_train_forward(x, gamma.detach().contiguous(), beta.detach().contiguous(), running_mean,
                                                    running_var, momentum, eps, relu, pb)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.relu, ctx.has_pre = relu, pre_bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta, dpre = ops.batchnorm_train_backward(x, dy.contiguous(), gamma.detach().contiguous(),
                                                               beta.detach().contiguous(), mean, rstd, ctx.relu, ctx.has_pre)
        return dx, dgamma, dbeta, dpre, None, None, None, None, None


def batchnorm_train(bn: torch.nn.BatchNorm2d, x: torch.Tensor, relu: bool, pre_bias=None) -> torch.Tensor:
    """nn.BatchNorm2d.forward in training mode (+ ReLU) on the device kernels; bookkeeping as torch's module does it
    (num_batches_tracked, momentum None = cumulative average).  `pre_bias`: the bias of the convolution in front, which the
    caller has NOT added to x (it only shifts the batch mean: the kernels account for it in running_mean and return its
    gradient, the float32 residue of an exact zero) -- saves the (n, C, H, W) broadcast add and its reduction."""
    momentum = 0.0 if bn.momentum is None else bn.momentum
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            momentum = 1.0 / float(bn.num_batches_tracked)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, pre_bias, rm, rv, momentum, bn.eps, relu)


def delta_dino_residual(delta_dino, frames: torch.Tensor, h: int, w: int, vit_patch: int = 14) -> torch.Tensor:
    """DeltaDINO.forward (delta_dino.py:53-61) in training mode -> the aligned residual [n, C, h, w].  Convolutions as
    unfold + GEMM (conv2d_gemm); on the device every BatchNorm2d -- with the ReLU that follows it and the bias of the conv in
    front of it -- and every blur-pool run on csrc/train.hip, forward and backward.  On host tensors (the CPU parity tests
    of this arithmetic) the BatchNorm2d modules themselves run."""
    x = frames
    layers = list(delta_dino.layers)
    fused_bn = x.is_cuda
    i = 0
    pending_bias = None  # a conv bias handed to the BatchNorm behind it instead of being added to the conv output
    while i < len(layers):
        layer = layers[i]
        nxt = layers[i + 1] if i + 1 < len(layers) else None
        if isinstance(layer, torch.nn.Conv2d):
            to_bn = fused_bn and isinstance(nxt, torch.nn.BatchNorm2d) and nxt.training and nxt.affine
            x = conv2d_gemm(x, layer.weight, None if to_bn else layer.bias, layer.padding[0], layer.dilation[0],
                            layer.padding_mode)
            pending_bias = layer.bias if to_bn else None
        elif isinstance(layer, torch.nn.BatchNorm2d) and layer.training and fused_bn and layer.affine:
            relu = isinstance(nxt, torch.nn.ReLU)
            x = batchnorm_train(layer, x, relu, pending_bias)
            pending_bias = None
            i += 1 if relu else 0
        else:
            x = layer(x)
        i += 1
    return align_cnn_to_vit(x, h, w, delta_dino.vit_stride, vit_patch, delta_dino.get_total_stride())


# ---- sampling -------------------------------------------------------------------------------------------------------------
# ---- one gradient buffer for the point / window consumers of the frame embeddings ---------------------------------------
class _SinkBuffer:
    """Token-major [F h w, C] gradient buffer shared by the consumers whose gradient touches a few cells per source."""

    def __init__(self):
        self.buf = None

    def get(self, n, h, w, c, like):
        if self.buf is None:
            self.buf = torch.zeros(n * h * w, c, dtype=like.dtype, device=like.device)
        return self.buf


class _GradSink(torch.autograd.Function):
    """Identity on the batch's frame embeddings.  The bilinear point reads and the fused tracker passes of an iteration each
    have a gradient that touches a few hundred cells per source, but autograd would have every one of them return a
    zero-filled full-size tensor (266 MB at C = 1024) and add them pairwise: seven fills and additions per iteration.  Those
    consumers instead ACCUMULATE into the buffer attached to this node's output (`_dtk_sink`) and return no gradient for it;
    this node runs after all of them (they are its dependents in the graph) and adds the buffer to the dense gradients that
    arrived the ordinary way."""

    @staticmethod
    def forward(ctx, emb, sink):
        ctx.sink = sink
        ctx.shape = emb.shape
        ctx.set_materialize_grads(False)   # no dense gradient arrived: `g` is None, not a zero-filled tensor
        return emb.view_as(emb)

    @staticmethod
    def backward(ctx, g):
        buf, ctx.sink.buf = ctx.sink.buf, None
        if buf is None:
            return g, None
        n, c, h, w = ctx.shape
        if buf.is_cuda:
            from . import ops
            d = ops.unpack_features(buf.view(n, h * w, c), h, w)
        else:
            d = buf.view(n, h, w, c).permute(0, 3, 1, 2)
        return (d if g is None else g + d), None


def attach_grad_sink(emb: torch.Tensor) -> torch.Tensor:
    """`emb` [n, C, h, w] (requires grad) -> the same values as a tensor whose point / window consumers share one gradient buffer."""
    sink = _SinkBuffer()
    out = _GradSink.apply(emb, sink)
    out._dtk_sink = sink
    return out


def _bilinear_corners(emb_shape, pts: torch.Tensor):
    """Frame index, the four corner cells (y, x) and their weights of Tracker.sample_embeddings' bilinear read."""
    n, c, h, w = emb_shape
    p = pts.detach()
    fx = ((p[:, 0] + 1) * 0.5 * (w - 1)).clamp(0, w - 1)
    fy = ((p[:, 1] + 1) * 0.5 * (h - 1)).clamp(0, h - 1)
    t = p[:, 2].round().long().clamp(0, n - 1)
    x0 = fx.floor().clamp(max=w - 1)
    y0 = fy.floor().clamp(max=h - 1)
    wx = (fx - x0)[:, None]
    wy = (fy - y0)[:, None]
    x0 = x0.long()
    y0 = y0.long()
    x1 = (x0 + 1).clamp(max=w - 1)
    y1 = (y0 + 1).clamp(max=h - 1)
    corners = ((y0, x0), (y0, x1), (y1, x0), (y1, x1))
    weights = ((1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy)
    return t, corners, weights


def _bilinear_read(emb, t, corners, weights):
    e = emb.permute(0, 2, 3, 1)  # [n, h, w, C]: one gather fetches a whole embedding (no copy of the volume)
    out = e[t, corners[0][0], corners[0][1]] * weights[0]
    for (y, x), wt in zip(corners[1:], weights[1:]):
        out = out + e[t, y, x] * wt
    return out


class _SampleBilinear(torch.autograd.Function):
    """The four-corner read with ONE gradient buffer: the traced form's backward builds a zero-filled full-size tensor per
    corner and sums them (at C = 1024 that is 4 x 266 MB of fills, four scatter kernels and three 0.8 GB additions per call,
    four calls per iteration); here the four corners are row-wise index_add_ into one token-major buffer."""

    @staticmethod
    def forward(ctx, emb, pts):
        t, corners, weights = _bilinear_corners(emb.shape, pts)
        ctx.save_for_backward(t, *[i for yx in corners for i in yx], *weights)
        ctx.shape = emb.shape
        ctx.sink = getattr(emb, "_dtk_sink", None)
        return _bilinear_read(emb, t, corners, weights)

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        t, yx, weights = sv[0], sv[1:9], sv[9:]
        n, c, h, w = ctx.shape
        d = ctx.sink.get(n, h, w, c, g) if ctx.sink is not None else torch.zeros(n * h * w, c, dtype=g.dtype, device=g.device)
        for i in range(4):
            d.index_add_(0, (t * h + yx[2 * i]) * w + yx[2 * i + 1], g * weights[i])
        if ctx.sink is not None:
            return None, None    # left in the shared buffer (_GradSink adds it)
        return d.view(n, h, w, c).permute(0, 3, 1, 2), None


class _SampleBilinearFused(torch.autograd.Function):
    """_SampleBilinear on two kernels (csrc/train.hip: dtk_sample_bilinear_forward / _backward; round 6): the read goes to the
    token-major copy of the batch's embeddings that the tracker passes of the iteration share (_packed_frames), the gradient is
    accumulated straight into the iteration's shared token-major buffer (_GradSink) -- ~65 small library launches per call become 2."""

    @staticmethod
    def forward(ctx, emb, pts):
        from . import ops
        from ._lib import check, lib
        n, c, h, w = emb.shape
        feat, _ = _packed_frames(emb)
        p = pts.detach().to(torch.float32).contiguous()
        out = torch.empty(p.shape[0], c, dtype=torch.float32, device=emb.device)
        check(lib().dtk_sample_bilinear_forward(feat.data_ptr(), p.data_ptr(), out.data_ptr(), p.shape[0], n, h, w, c, ops._stream()))
        ctx.save_for_backward(p)
        ctx.shape = emb.shape
        ctx.sink = getattr(emb, "_dtk_sink", None)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        from ._lib import check, lib
        (p,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        g = g.contiguous()
        d = ctx.sink.get(n, h, w, c, g) if ctx.sink is not None else torch.zeros(n * h * w, c, dtype=g.dtype, device=g.device)
        check(lib().dtk_sample_bilinear_backward(g.data_ptr(), p.data_ptr(), d.data_ptr(), p.shape[0], n, h, w, c, ops._stream()))
        if ctx.sink is not None:
            return None, None    # left in the shared buffer (_GradSink adds it)
        return ops.unpack_features(d.view(n, h * w, c), h, w), None


USE_FUSED_SAMPLE = os.environ.get("DTK_TRAIN_SAMPLE", "fused") == "fused"


def sample_bilinear(emb: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """Tracker.sample_embeddings (tracker.py:96-111): emb [n, C, h, w], pts [B, 3] = (x, y in [-1, 1] token-grid
    coordinates, frame index into emb) -> [B, C]; align_corners, border clamp.  Gradient flows to `emb` only (the
    reference detaches the points, utils.py:91)."""
    if emb.requires_grad and torch.is_grad_enabled():
        if USE_FUSED_SAMPLE and emb.is_cuda and emb.dtype == torch.float32 and pts.shape[0] > 0:
            return _SampleBilinearFused.apply(emb, pts)
        return _SampleBilinear.apply(emb, pts)
    return _bilinear_read(emb, *_bilinear_corners(emb.shape, pts))


# ---- correlation ------------------------------------------------------------------------------------------------------------
def cosine_maps(src: torch.Tensor, frames: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """tracker.py:158-169: rho[b] = <src[b], frames[tgt[b]]> / max(|src[b]| |frames[tgt[b]]|, 1e-8) -> [B, h, w].
    One batched product of every source with every frame of the set, of which each source keeps its target's map (what the
    reference's einsum does, tracker.py:159-160): with n <= 8 frames in a training batch that is a few GFLOP, and -- unlike
    grouping the sources by target frame, round 2's form -- it needs no host read of the target indices (every
    `unique().tolist()` / `nonzero()` is a device synchronisation; they were 9 of the ~55 per iteration)."""
    n, c, h, w = frames.shape
    b = src.shape[0]
    tgt = tgt.long()
    fl = frames.reshape(n, c, h * w)
    rows = torch.arange(b, device=src.device)
    dots = torch.matmul(src, fl)[tgt, rows]                       # [n, B, HW] -> [B, HW]
    den = (src.norm(dim=1)[:, None] * fl.norm(dim=1)[tgt]).clamp(min=EPS)
    return (dots / den).reshape(b, h, w)


# ---- tracker head -----------------------------------------------------------------------------------------------------------
def normalized_weight(weight: torch.Tensor) -> torch.Tensor:
    """conv_norm.py:34-44: W / sum_{kh,kw} W, a sum with |.| < 1e-8 replaced by sign(.) 1e-8."""
    s = weight.sum(dim=[2, 3], keepdim=True)
    s = torch.where(s.abs() < EPS, torch.sign(s) * EPS, s)
    return weight / s


def _shifted_planes(x: torch.Tensor) -> torch.Tensor:
    """[B, h, w] -> [B, 9, h, w]: plane 3 dy + dx holds x shifted so that cell (r, c) reads x[r + dy - 1, c + dx - 1] (zero
    outside): the unfolded 3 x 3 neighbourhood of every cell for the whole batch in one copy kernel."""
    h, w = x.shape[-2:]
    xp = F.pad(x, (1, 1, 1, 1))
    return torch.stack([xp[:, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], dim=1)


class _TapContract(torch.autograd.Function):
    """y[b, o, k] = sum_t w[o, t] x[b, t, k] for a TINY weight matrix (16 x 9 / 9 x 16) and B maps of k = h w cells.
    The point is the backward: autograd's einsum gradient computes dw as ONE product with a 16 x 9 output and a reduction
    over all B h w = 4e6 positions, which the GEMM library runs on a handful of workgroups (2.3 ms per call, six calls = 14 ms
    of a training iteration: `Cijk_..._MT16x16x512` in profiles/r03_train_kernel_trace.md).  Here dw is a batched product per
    map -- B independent 16 x 9 results -- and a sum over the batch."""

    @staticmethod
    def forward(ctx, w, x):
        ctx.save_for_backward(w, x)
        return torch.matmul(w, x)

    @staticmethod
    def backward(ctx, dy):
        w, x = ctx.saved_tensors
        dw = dx = None
        if ctx.needs_input_grad[0]:
            dw = torch.bmm(dy, x.transpose(1, 2)).sum(dim=0)
        if ctx.needs_input_grad[1]:
            dx = torch.matmul(w.t(), dy)
        return dw, dx


def head_logits(head, x: torch.Tensor) -> torch.Tensor:
    """cnn_refiner (tracker_head.py:47-58): [B, 1, h, w] -> [B, 1, h, w], as two matrix products over the whole batch:
        hidden[16] = relu(W1[16 x 9] . neighbourhood(x) + b1)
        P[9]       = W2[9 x 16] . hidden          (per-cell projection onto the nine taps of the second conv)
        z(r, c)    = b2 + sum_tap P[tap](r + dy - 1, c + dx - 1)
    -- the formulation of the inference kernels (csrc/track_mfma.hip, refine_head_kernel)."""
    c0, c2 = head.cnn_refiner[0], head.cnn_refiner[2]
    b, _, h, w = x.shape
    w1 = normalized_weight(c0.weight).reshape(c0.out_channels, 9)
    w2 = normalized_weight(c2.weight).reshape(c2.in_channels, 9)
    hid = _TapContract.apply(w1, _shifted_planes(x[:, 0]).reshape(b, 9, h * w)).reshape(b, c0.out_channels, h, w)
    if c0.bias is not None:
        hid = hid + c0.bias[None, :, None, None]
    planes = _TapContract.apply(w2.t(), torch.relu(hid).reshape(b, c2.in_channels, h * w)).reshape(b, 9, h, w)
    planes = F.pad(planes, (1, 1, 1, 1))
    z = sum(planes[:, 3 * dy + dx, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    if c2.bias is not None:
        z = z + c2.bias[0]
    return z[:, None]


def soft_argmax(p: torch.Tensor, peak: torch.Tensor, patch: int, stride: int, radius: float) -> torch.Tensor:
    """tracker_head.py:68-98: p [B, h, w] (softmax over the map), peak [B] flat arg-max of the cost volume -> [B, 2]
    pixel (x, y): centre of mass of p inside the disk of `radius` px around the peak cell; a disk whose mass is below
    1e-8 is replaced by the uniform distribution over the disk."""
    b, h, w = p.shape
    ys = torch.arange(h, device=p.device, dtype=torch.float32) * stride + patch // 2
    xs = torch.arange(w, device=p.device, dtype=torch.float32) * stride + patch // 2
    pr = torch.div(peak, w, rounding_mode="floor")
    pc = peak - pr * w
    dy = ys[None, :] - (pr * stride + patch // 2).to(torch.float32)[:, None]   # [B, h]
    dx = xs[None, :] - (pc * stride + patch // 2).to(torch.float32)[:, None]   # [B, w]
    mask = torch.sqrt(dy[:, :, None] ** 2 + dx[:, None, :] ** 2) <= radius
    q = p * mask
    s = q.sum(dim=(1, 2))
    zero = s < EPS
    uni = (1.0 / mask.sum(dim=(1, 2)).to(p.dtype))[:, None, None]
    q = torch.where(zero[:, None, None], (q + uni) * mask, q)
    s = torch.where(zero, q.sum(dim=(1, 2)), s)
    px = (q.sum(dim=1) * xs[None, :]).sum(dim=1) / s
    py = (q.sum(dim=2) * ys[None, :]).sum(dim=1) / s
    return torch.stack([px, py], dim=1)


class _HeadFused(torch.autograd.Function):
    """TrackerHead.forward on head_exact_kernel and its LOCAL backward (csrc/track_exact.hip, dtk_head_forward_train /
    dtk_head_backward): the autograd-traced form below is ~60 kernels forward and ~120 backward over [B, 16, h, w] tensors;
    this is one launch each way.  `packed` = w1n[16, 9] | b1[16] | w2n[16, 9] | b2 built by torch from the NORMALISED weights,
    so the normalisation W / sum W (conv_norm.py:34-44) and its gradient stay with autograd."""

    @staticmethod
    def forward(ctx, maps, packed, geom):
        from . import ops
        maps = maps.contiguous()
        packed = packed.contiguous()
        out, stats = ops.head_forward_train(geom, packed, maps, normalized=True)
        ctx.save_for_backward(maps, packed, stats)
        ctx.geom = geom
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, gout, _gstats):
        from . import ops
        maps, packed, stats = ctx.saved_tensors
        dmaps, dpacked = ops.head_backward(ctx.geom, packed, maps, stats, gout.contiguous(), normalized=True)
        return dmaps, dpacked, None


HEAD_FALLBACK_EXACT = bool(os.environ.get("DTK_HEAD_FALLBACK_EXACT"))
USE_FUSED_HEAD = True  # device tensors: the head of the training step on the hand-written forward / backward kernels


def _head_forward_fused(head, cost: torch.Tensor):
    """The fused route of head_forward, or None when it does not apply (geometry; a fallback under HEAD_FALLBACK_EXACT)."""
    b, _, h, w = cost.shape
    packed_geom = _head_packed(head, 1, 32, h, w, cost)
    if packed_geom is None:
        return None
    packed, geom = packed_geom
    out, stats = _HeadFused.apply(cost.reshape(b, h * w), packed, geom)
    # A map whose zero-mass fallback fired (tracker_head.py:86-94) has a dense gradient of total size <= 1e-8 |dq|; the
    # kernel returns its part on the disk and drops the rest (csrc/track_exact.hip).  HEAD_FALLBACK_EXACT reads the
    # statistics on the host (a device synchronisation per call) and sends such a batch through the traced route.
    if HEAD_FALLBACK_EXACT and bool((stats[:, 3] < EPS).any()):
        return None
    return out


_PACKED = {}


def _packed_frames(frames: torch.Tensor):
    """Token-major copy + norms of the batch's frame embeddings (dtk_pack_features), shared by the calls of one iteration
    (the prediction, the cycle-consistency passes) while the tensor is the same object at the same version."""
    import weakref
    from . import ops
    ref = _PACKED.get("ref")
    if ref is None or ref() is not frames or _PACKED["version"] != frames._version:  # identity, not address: the allocator
        feat, norms = ops.pack_features(frames.detach().contiguous())                # reuses addresses across iterations
        _PACKED.update(ref=weakref.ref(frames), version=frames._version, feat=feat, norms=norms)
    return _PACKED["feat"], _PACKED["norms"]


class _TrackFused(torch.autograd.Function):
    """Tracker.get_point_predictions_from_embeddings (tracker.py:158-180) of the training step as kernels both ways:
        forward   dtk_corr_maps (relu'd cosine map of every source against ITS target frame, sources ordered by target frame)
                  -> dtk_head_forward_train
        backward  dtk_head_backward (local: the map gradient lives on the 15 x 15 window around the arg-max)
                  -> dtk_corr_window_backward (the cosine gradient on those <= 225 cells only) -> dtk_unpack_features.
    The traced form multiplies every source with every frame of the batch and keeps one map (tracker.py:159-160), and its
    backward is two dense products with a gradient that is 99.7 % zeros: 3.5 ms of GEMM + ~6 ms of [B, 8 h w] element-wise
    kernels per iteration."""

    @staticmethod
    def forward(ctx, src, frames, tgt, packed, geom):
        from . import ops
        feat, norms = _packed_frames(frames)
        order = torch.argsort(tgt, stable=True)          # corr_exact tiles 64 sources: one or two target frames per tile
        emb = src.detach().index_select(0, order).contiguous()
        tgt_s = tgt.index_select(0, order).to(torch.int32).contiguous()
        maps = ops.corr_maps(geom, feat, norms, emb, tgt_s, relu=True).reshape(emb.shape[0], -1)
        packed = packed.contiguous()
        out_s, stats = ops.head_forward_train(geom, packed, maps, normalized=True)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.shape[0], device=order.device)
        ctx.save_for_backward(emb, tgt_s, maps, stats, packed, feat, norms, order, inv)
        ctx.geom, ctx.frames_shape = geom, frames.shape
        ctx.sink = getattr(frames, "_dtk_sink", None)
        return out_s.index_select(0, inv)

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        emb, tgt_s, maps, stats, packed, feat, norms, order, inv = ctx.saved_tensors
        g = ctx.geom
        dmaps, dpacked = ops.head_backward(g, packed, maps, stats, gout.index_select(0, order).contiguous(), normalized=True)
        n, c, h, w = ctx.frames_shape
        if ctx.sink is not None:   # accumulate into the iteration's shared buffer (same token-major layout as `feat`)
            demb = ops.corr_window_backward(g, feat, norms, emb, tgt_s, maps, dmaps, stats, ctx.sink.get(n, h, w, c, feat))
            return demb.index_select(0, inv), None, None, dpacked, None
        dfeat = torch.zeros_like(feat)
        demb = ops.corr_window_backward(g, feat, norms, emb, tgt_s, maps, dmaps, stats, dfeat)
        return demb.index_select(0, inv), ops.unpack_features(dfeat, h, w), None, dpacked, None


USE_FUSED_TRACK = True  # device tensors: correlation + head of the training step on _TrackFused


def track_points(head, src: torch.Tensor, frames: torch.Tensor, tgt: torch.Tensor):
    """tracker.py:171-180 for the training step: source embeddings [B, C], the batch's frame embeddings [n, C, h, w], target
    indices [B] -> [B, 2] normalised positions.  Device tensors take _TrackFused; host tensors (the CPU parity tests) and
    geometries the kernels do not cover take the traced statement."""
    if USE_FUSED_TRACK and not HEAD_FALLBACK_EXACT and src.is_cuda and src.dtype == torch.float32 and src.shape[0] > 0:
        packed_geom = _head_packed(head, frames.shape[0], frames.shape[1], frames.shape[2], frames.shape[3], src)
        if packed_geom is not None and frames.shape[1] <= 1024 and frames.shape[1] % 16 == 0:
            return _TrackFused.apply(src, frames, tgt, *packed_geom)
    return head_forward(head, torch.relu(cosine_maps(src, frames, tgt))[:, None])


_HEAD_PACKED = {}


def new_iteration() -> None:
    """Drop what the calls of ONE training iteration share (the packed head parameters: three tracker passes use one copy, i.e. one
    set of ~20 small launches and one gradient edge instead of three).  Tracker._forward_train calls this first."""
    _HEAD_PACKED.clear()


def _head_packed(head, n, c, h, w, like):
    """(packed normalised head parameters with autograd into the weights, geometry) or None if the kernels do not apply.  Shared by
    the tracker passes of an iteration (new_iteration) while the parameters are the same objects at the same versions."""
    c0, c2 = head.cnn_refiner[0], head.cnn_refiner[2]
    key = (id(head), n, c, h, w, torch.is_grad_enabled(), c0.weight._version, c2.weight._version,
           None if c0.bias is None else c0.bias._version, None if c2.bias is None else c2.bias._version)
    hit = _HEAD_PACKED.get(key)
    if hit is not None:
        return hit
    out = _head_packed_compute(head, n, c, h, w, like)
    _HEAD_PACKED.clear()
    _HEAD_PACKED[key] = out
    return out


def _head_packed_compute(head, n, c, h, w, like):
    from ._lib import make_geom
    c0, c2 = head.cnn_refiner[0], head.cnn_refiner[2]
    if c0.out_channels != 16 or c2.in_channels != 16 or float(head.argmax_radius) / float(head.step_h) > 5.0:
        return None
    geom = make_geom(n, c, head.video_h, head.video_w, head.patch_size, head.step_h, float(head.argmax_radius))
    if (geom.ph, geom.pw) != (h, w):
        return None
    b1 = c0.bias if c0.bias is not None else like.new_zeros(16)
    b2 = c2.bias if c2.bias is not None else like.new_zeros(1)
    packed = torch.cat([normalized_weight(c0.weight).reshape(-1), b1.reshape(-1), normalized_weight(c2.weight).reshape(-1),
                        b2.reshape(-1)])
    return packed, geom


def head_forward(head, cost: torch.Tensor) -> torch.Tensor:
    """TrackerHead.forward (tracker_head.py:107-121): cost [B, 1, h, w] >= 0 -> [B, 2] in [-1, 1]."""
    b, _, h, w = cost.shape
    if USE_FUSED_HEAD and cost.is_cuda and cost.dtype == torch.float32 and b > 0:
        out = _head_forward_fused(head, cost)
        if out is not None:
            return out
    peak = cost[:, 0].reshape(b, h * w).argmax(dim=1)
    p = torch.softmax(head_logits(head, cost).reshape(b, h * w), dim=1).reshape(b, h, w)
    xy = soft_argmax(p, peak, head.patch_size, head.step_h, float(head.argmax_radius))
    scale = torch.tensor([head.video_w - 1, head.video_h - 1], device=xy.device, dtype=xy.dtype)
    return 2.0 * xy / scale - 1.0


# ---- the k largest per row, for long rows --------------------------------------------------------------------------------------------
TOPK_CHUNK, TOPK_DIRECT, TOPK_SLICES = 2048, 2500, 768


def _topk_one_block(x: torch.Tensor, k: int):
    """torch.topk(x, k, dim=1) kept on its one-block-per-slice kernel: rows no longer than TOPK_DIRECT, at most TOPK_SLICES rows per call
    (ATen picks the multi-block path from (slices, slice length): fewer than 800 slices of fewer than 3000 elements never take it; the
    cycle term's 4 x 406 504 keys are 796 slices of 2048 -- one launch per level)."""
    if x.shape[0] <= TOPK_SLICES:
        return torch.topk(x, k, dim=1)
    parts = [torch.topk(p, k, dim=1) for p in x.split(TOPK_SLICES)]
    return torch.cat([p.values for p in parts]), torch.cat([p.indices for p in parts])


def topk_rows(x: torch.Tensor, k: int):
    """torch.topk(x, k, dim=1) (values, indices) for a 2-D tensor whose rows may be long.  Few long rows send torch.topk to its
    multi-block radix path, which does not survive a captured graph on this stack (its second replay reads state the first one left
    behind: a memory fault in trainer.GraphedIteration, found by bisection with scripts/dev/graph_bisect.py).  Long rows go in
    levels instead -- the k largest of every TOPK_CHUNK-wide piece, then the k largest of those candidates: the same set, every level on
    the one-block-per-slice kernel.  Ties between equal values may resolve to a different index than torch.topk's (the callers' values
    are continuous random keys)."""
    rows, n = x.shape
    if n <= TOPK_DIRECT:
        return _topk_one_block(x, k)
    if 2 * k > TOPK_CHUNK:      # (every level must at least halve the row)
        return torch.topk(x, k, dim=1)
    groups = (n + TOPK_CHUNK - 1) // TOPK_CHUNK
    pad = groups * TOPK_CHUNK - n
    if pad:
        x = torch.cat([x, torch.full((rows, pad), float("-inf"), dtype=x.dtype, device=x.device)], dim=1)
    first = _topk_one_block(x.view(rows * groups, TOPK_CHUNK), k)
    cand_val = first[0].view(rows, groups * k)
    base = (torch.arange(groups, device=x.device) * TOPK_CHUNK)[None, :, None]
    cand_idx = (first[1].view(rows, groups, k) + base).view(rows, groups * k)
    second = topk_rows(cand_val, k)
    return second[0], cand_idx.gather(1, second[1])


# ---- fused Adam (round 5; csrc/train.hip: dtk_adam_step) ----------------------------------------------------------------------------
def _adam_args(optimizer, advance: bool = True):
    """The argument block of dtk_adam_step for the optimizer's current state (-> (AdamArgs, tensors to keep alive) or (None, [])
    when no parameter has a gradient).  `advance`: count this call as a step (torch.optim.Adam's lazy state initialisation and
    `state["step"] += 1`).  Raises for options the kernel does not implement -- BEFORE any state is touched (a
    NotImplementedError raised half-way through the walk left the step counters of the tensors already visited advanced with no
    update applied)."""
    from ._lib import ADAM_MAX_GROUPS, ADAM_MAX_TENSORS, AdamArgs
    groups = optimizer.param_groups
    if len(groups) > ADAM_MAX_GROUPS:
        raise NotImplementedError(f"fused Adam: {len(groups)} parameter groups (max {ADAM_MAX_GROUPS})")
    live = [p for grp in groups for p in grp["params"] if p.grad is not None]
    if len(live) > ADAM_MAX_TENSORS:
        raise NotImplementedError(f"fused Adam: more than {ADAM_MAX_TENSORS} parameter tensors")
    g0 = (float(groups[0]["betas"][0]), float(groups[0]["betas"][1]), float(groups[0]["eps"]))
    for grp in groups:
        if grp.get("weight_decay", 0) != 0 or grp.get("amsgrad", False) or grp.get("maximize", False):
            raise NotImplementedError("fused Adam: weight_decay / amsgrad / maximize are not implemented")
        if (float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"])) != g0:
            raise NotImplementedError("fused Adam: per-group betas / eps")
    for p in live:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32):
            raise NotImplementedError("fused Adam: contiguous fp32 device parameters only")
    a = AdamArgs()
    a.beta1, a.beta2, a.eps = g0
    n = 0
    keep = []
    for gi, grp in enumerate(groups):
        a.lr[gi] = float(grp["lr"])
        for p in grp["params"]:
            if p.grad is None:
                continue
            st = optimizer.state[p]
            if len(st) == 0:   # torch.optim.Adam's lazy state initialisation
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if advance:
                st["step"] += 1
            s_now = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
            g = p.grad.contiguous()
            keep.append(g)
            a.param[n], a.grad[n] = p.data_ptr(), g.data_ptr()
            a.exp_avg[n], a.exp_avg_sq[n] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.numel[n], a.group[n], a.step[n] = p.numel(), gi, max(s_now, 1)
            n += 1
    if n == 0:
        return None, []
    a.n_tensors = n
    return a, keep


def _bump_versions(optimizer) -> None:
    """The kernel writes the parameters through raw pointers: tell autograd (and everything keyed on `_version`, e.g. the packed head
    parameters shared by an iteration's tracker passes) that they changed, as an in-place torch op would."""
    bump = getattr(torch.autograd.graph, "increment_version", None)
    for grp in optimizer.param_groups:
        for p in grp["params"]:
            if p.grad is not None:
                if bump is not None:
                    bump(p)
                else:
                    p.data.add_(0)


def fused_adam_step(optimizer) -> None:
    """One step of a torch.optim.Adam instance (dino_tracker.py:110-115: default betas / eps, no weight decay, no amsgrad, two
    parameter groups whose learning rates the LambdaLR of optimization/schedulers.py:4-8 rewrites) as ONE kernel launch over all
    parameter tensors.  The optimizer object stays torch's: its state dict keeps torch's keys (`step`, `exp_avg`, `exp_avg_sq`), so
    checkpoints and the scheduler work unchanged.  Raises for options the kernel does not implement."""
    import ctypes

    from ._lib import check, lib
    from . import ops
    a, _keep = _adam_args(optimizer)
    if a is None:
        return
    with torch.no_grad():
        check(lib().dtk_adam_step(ctypes.byref(a), ops._stream()))
    _bump_versions(optimizer)


class GraphAdam:
    """The fused Adam step of a CAPTURED iteration (trainer.GraphedIteration).  A launch baked into a graph reads its two per-tensor
    scalars (step size, 1 / sqrt(bias correction 2)) from `self.scalars` on the device; `refresh(params)` advances the step counts of
    the tensors that launch updates, forms the scalars from the optimizer's CURRENT learning rates exactly as dtk_adam_step does
    (dtk_adam_scalars: the arithmetic lives in one place) and queues their upload on the current stream -- call it before every
    replay.  The optimizer object and its state dict stay torch's."""

    def __init__(self, optimizer, device):
        from ._lib import ADAM_MAX_TENSORS
        self.optimizer = optimizer
        self.scalars = torch.zeros(2 * ADAM_MAX_TENSORS, dtype=torch.float32, device=device)
        self._host = [torch.zeros(2 * ADAM_MAX_TENSORS, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._events = [None] * 4
        self._next = 0

    def launch(self):
        """Inside a capture, after backward: the launch.  Returns (parameters it updates, in its order; their gradient tensors --
        the addresses are baked into the launch, so the caller keeps them alive as long as the graph)."""
        import ctypes

        from ._lib import check, lib
        from . import ops
        a, keep = _adam_args(self.optimizer, advance=False)
        if a is None:
            raise RuntimeError("GraphAdam: no parameter has a gradient")
        params = [p for grp in self.optimizer.param_groups for p in grp["params"] if p.grad is not None]
        with torch.no_grad():
            check(lib().dtk_adam_step_dev(ctypes.byref(a), ctypes.c_void_p(self.scalars.data_ptr()), ops._stream()))
        _bump_versions(self.optimizer)
        return params, keep

    def refresh(self, params):
        import ctypes

        from ._lib import AdamArgs, check, lib
        opt = self.optimizer
        a = AdamArgs()
        g0 = opt.param_groups[0]
        a.beta1, a.beta2, a.eps = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"])
        group_of = {id(p): gi for gi, grp in enumerate(opt.param_groups) for p in grp["params"]}
        for gi, grp in enumerate(opt.param_groups):
            a.lr[gi] = float(grp["lr"])
        for n, p in enumerate(params):
            st = opt.state[p]
            st["step"] += 1
            a.group[n] = group_of[id(p)]
            a.step[n] = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])
        a.n_tensors = len(params)
        i = self._next
        self._next = (i + 1) % len(self._host)
        if self._events[i] is not None:
            self._events[i].synchronize()
        host = self._host[i]
        check(lib().dtk_adam_scalars(ctypes.byref(a), ctypes.cast(host.data_ptr(), ctypes.POINTER(ctypes.c_float))))
        self.scalars.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[i] = ev


def install_fused_adam(optimizer):
    """Replace `optimizer.step` of a plain torch.optim.Adam by the fused kernel (same object: the scheduler and the checkpoint code of
    the reference keep working).  DTK_TRAIN_ADAM=torch keeps torch's own step.  Returns the optimizer."""
    import os
    if os.environ.get("DTK_TRAIN_ADAM", "fused") != "fused" or type(optimizer) is not torch.optim.Adam:
        return optimizer
    if any(g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False) or g.get("capturable", False)
           for g in optimizer.param_groups):
        return optimizer
    from ._lib import ADAM_MAX_GROUPS, ADAM_MAX_TENSORS
    params = [p for g in optimizer.param_groups for p in g["params"]]
    # what the kernel cannot take keeps torch's own step (checked here, once, instead of failing at the first optimizer.step())
    if any(not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) for p in params) or len(params) > ADAM_MAX_TENSORS or \
            len(optimizer.param_groups) > ADAM_MAX_GROUPS:
        return optimizer
    g0 = optimizer.param_groups[0]
    if any((g["betas"], g["eps"]) != (g0["betas"], g0["eps"]) for g in optimizer.param_groups):
        return optimizer

    def step(closure=None):
        loss = closure() if closure is not None else None
        fused_adam_step(optimizer)
        optimizer._opt_called = True        # (what torch's LR schedulers look at to check the step / scheduler call order)
        return loss
    step._wrapped_by_lr_sched = True        # an LRScheduler built on this optimizer wrapped the original step: keep its bookkeeping quiet
    optimizer.step = step
    optimizer._dtk_fused = True
    return optimizer
