"""Parameter containers with the reference's module names / state-dict keys, whose forward passes run in HIP.

  TrackerHead      models/networks/tracker_head.py:35-121   keys cnn_refiner.{0,2}.{weight,bias}
  NormalizedConv2d models/networks/conv_norm.py:7-46
  DeltaDINO        models/networks/delta_dino.py:7-61       keys layers.{0,4,8,12}.*, layers.{1,5,9,13}.*,
                                                            layers.{3,7,11}.filt
Checkpoints written by the reference load unchanged (`load_state_dict`), and vice versa.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from ._lib import make_geom


class NormalizedConv2d(nn.Module):
    """Holds weight [out,in,k,k] and bias [out]; the W / sum(W) normalisation (conv_norm.py:34-46) and the
    convolution itself happen inside the tracker kernels (dtk_head_prepare / dtk_track)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        """conv_norm.py:42-46 on the device (dtk_normalized_conv2d).  The tracker path does not come through here: both
        layers of TrackerHead.cnn_refiner are fused into the head kernels."""
        if self.training and torch.is_grad_enabled():  # test-time training: autograd through the normalisation
            from . import train_ops
            return torch.nn.functional.conv2d(x, train_ops.normalized_weight(self.weight), self.bias, stride=self.stride,
                                              padding=self.padding)
        if self.stride != 1 or self.padding != self.kernel_size // 2:
            raise NotImplementedError("NormalizedConv2d on the device: stride 1, padding k // 2 (the reference's use)")
        return ops.normalized_conv2d(x.detach().to(torch.float32).contiguous(), self.weight.detach().contiguous(),
                                     None if self.bias is None else self.bias.detach().contiguous())


class TrackerHead(nn.Module):
    def __init__(self, use_cnn_refiner=True, in_channels=1, hidden_channels=16, out_channels=1, kernel_size=3,
                 stride=1, patch_size=14, step_h=14, step_w=14, argmax_radius=35, video_h=480, video_w=640):
        super().__init__()
        if not use_cnn_refiner or (in_channels, hidden_channels, out_channels, kernel_size, stride) != (1, 16, 1, 3, 1):
            raise NotImplementedError("the HIP tracker head implements the reference configuration 1->16->1, 3x3")
        if step_h != step_w:
            raise NotImplementedError("anisotropic token stride")
        padding = kernel_size // 2
        self.cnn_refiner = nn.Sequential(
            NormalizedConv2d(in_channels, hidden_channels, kernel_size, stride, padding=padding),
            nn.ReLU(inplace=True),
            NormalizedConv2d(hidden_channels, out_channels, kernel_size, stride, padding=padding),
        )
        self.argmax_radius = argmax_radius
        self.patch_size, self.step_h, self.step_w = patch_size, step_h, step_w
        self.video_h, self.video_w = video_h, video_w
        self._packed = None
        self._packed_key = None

    def packed_params(self, device) -> torch.Tensor:
        """Normalised parameters in the kernels' layout; re-packed whenever a parameter changed."""
        ps = [self.cnn_refiner[0].weight, self.cnn_refiner[0].bias, self.cnn_refiner[2].weight, self.cnn_refiner[2].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps) + (str(device),)
        if self._packed is None or key != self._packed_key:
            self._packed = ops.head_prepare(self.state_dict(), device)
            self._packed_key = key
        return self._packed

    def geom(self, T: int = 1, C: int = 4):
        return make_geom(T, C, self.video_h, self.video_w, self.patch_size, self.step_h, float(self.argmax_radius))

    def forward(self, cost_volume):
        """cost_volume [B,1,h,w] (already ReLU'd, tracker.py:173) -> [B,2] normalised (x,y) (tracker_head.py:107-121)."""
        if self.training and torch.is_grad_enabled():  # test-time training (train_ops: autograd supplies the backward)
            from . import train_ops
            return train_ops.head_forward(self, cost_volume)
        b, c, h, w = cost_volume.shape
        g = self.geom()
        if (h, w) != (g.ph, g.pw) or c != 1:
            raise RuntimeError(f"cost volume {tuple(cost_volume.shape)} does not match the {g.ph}x{g.pw} token grid")
        maps = cost_volume.detach().reshape(b, h * w).to(torch.float32).contiguous()
        return ops.head_forward(g, self.packed_params(maps.device), maps, normalized=True)


class _BlurPoolParams(nn.Module):
    """Carries the `filt` buffer of antialiased_cnns.BlurPool so reference checkpoints load (layers.{3,7,11}.filt)."""

    def __init__(self, channels):
        super().__init__()
        a = torch.tensor([1.0, 3.0, 3.0, 1.0])
        self.register_buffer("filt", (a[:, None] * a[None, :] / 64.0)[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x):
        from . import train_ops
        return train_ops.blurpool(x, self.filt)


class DeltaDINO(nn.Module):
    def __init__(self, channels=[3, 64, 128, 256, 1024], dilations=[1, 1, 1, 2], kernel_size=5, down_stride=2,
                 padding_mode="reflect", downsample_layers=[True, True, True, False], vit_stride=7):
        super().__init__()
        if (list(dilations), kernel_size, down_stride, padding_mode, list(downsample_layers)) != (
                [1, 1, 1, 2], 5, 2, "reflect", [True, True, True, False]) or len(channels) != 5:
            raise NotImplementedError("the HIP Delta-DINO implements the reference architecture only")
        self.channels = list(channels)
        self.downsample_layers, self.vit_stride, self.down_stride = downsample_layers, vit_stride, down_stride
        layers = []
        for i in range(4):
            last = i == 3
            dil = dilations[i]
            pad = (kernel_size + (kernel_size - 1) * (dil - 1)) // 2
            conv = nn.Conv2d(channels[i], channels[i + 1], kernel_size, stride=1, dilation=dil, padding=pad,
                             padding_mode=padding_mode)
            if last:  # delta_dino.py:33-35: zero init => residual starts at 0
                nn.init.zeros_(conv.weight)
                nn.init.zeros_(conv.bias)
            layers.append(conv)
            layers.append(nn.BatchNorm2d(channels[i + 1]))
            if last:
                layers[-1].weight.data.fill_(0.05)
            else:
                layers.append(nn.ReLU())
            if downsample_layers[i]:
                layers.append(_BlurPoolParams(channels[i + 1]))
        self.layers = nn.ModuleList(layers)

    def get_total_stride(self):
        return self.down_stride ** sum(self.downsample_layers)

    def forward(self, x, vit_features):
        """delta_dino.py:53-61: frames x [B,3,H,W] in [0,1], vit_features [B,C,h,w] (used for its grid size only, as in
        the reference) -> the residual [B,C,h,w] = align_cnn_vit_features(CNN(x)) (models/utils.py:7-45, patch 14)."""
        if self.training:  # batch-statistics BatchNorm + autograd graph: test-time training
            from . import train_ops
            return train_ops.delta_dino_residual(self, x, vit_features.shape[-2], vit_features.shape[-1])
        from .delta_dino import residual_frames
        return residual_frames(self, x, vit_features)
