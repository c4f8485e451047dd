"""TEST INFRASTRUCTURE ONLY -- stand-in for the un-vendored `antialiased_cnns` package.

The reference imports `antialiased_cnns.BlurPool` (/root/reference/models/networks/delta_dino.py:3,44;
requirements.txt:5, unpinned) but the package is not installed here and there is no network.
This restates its published algorithm (Zhang, "Making Convolutional Networks Shift-Invariant Again",
adobe/antialiased-cnns `BlurPool`): reflect-pad (left 1, right 2, top 1, bottom 2) for filt_size=4,
then a depthwise conv with outer([1,3,3,1])/64 at `stride`.  PARITY UNPINNED: no copy of the original
source exists in this container to diff against; only the buffer name/shape (`filt`, [C,1,4,4]) is
pinned by the reference's checkpoint keys.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class BlurPool(nn.Module):
    def __init__(self, channels, pad_type="reflect", filt_size=4, stride=2, pad_off=0):
        super().__init__()
        assert pad_type in ("reflect", "refl"), "only the reflect variant is restated"
        self.filt_size = filt_size
        self.stride = stride
        self.channels = channels
        lo = int(1.0 * (filt_size - 1) / 2)
        hi = int(np.ceil(1.0 * (filt_size - 1) / 2))
        self.pad_sizes = [lo + pad_off, hi + pad_off, lo + pad_off, hi + pad_off]
        rows = {1: [1.0], 2: [1.0, 1.0], 3: [1.0, 2.0, 1.0], 4: [1.0, 3.0, 3.0, 1.0],
                5: [1.0, 4.0, 6.0, 4.0, 1.0]}[filt_size]
        a = torch.tensor(rows)
        filt = a[:, None] * a[None, :]
        filt = filt / filt.sum()
        self.register_buffer("filt", filt[None, None].repeat(channels, 1, 1, 1))
        self.pad = nn.ReflectionPad2d(self.pad_sizes)

    def forward(self, inp):
        return F.conv2d(self.pad(inp), self.filt, stride=self.stride, groups=inp.shape[1])
