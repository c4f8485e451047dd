"""RangeNormalizer -- same contract as the reference's data/dataset.py:5-53 (px / frame index <-> [dst0, dst1]).
Tiny affine host-side glue around the kernels; unlike the reference its default device is 'cpu' so that
CPU-only plumbing (dino_tracker.py:29 constructs one before any model exists) does not need a GPU."""
import torch


class RangeNormalizer(torch.nn.Module):
    def __init__(self, shapes: tuple, device="cpu"):
        super().__init__()
        self.register_buffer("normalizer", torch.tensor(shapes).float().to(device) - 1)

    def forward(self, x, dst=(0, 1), dims=[0, 1, 2]):
        out = x.clone()
        out[:, dims] = x[:, dims] / self.normalizer[dims]
        out[:, dims] = (dst[1] - dst[0]) * out[:, dims] + dst[0]
        return out

    def unnormalize(self, normalized_x: torch.Tensor, src=(0, 1), dims=[0, 1, 2]):
        x = normalized_x.clone()
        x[:, dims] = (normalized_x[:, dims] - src[0]) / (src[1] - src[0])
        x[:, dims] = x[:, dims] * self.normalizer[dims]
        return x
