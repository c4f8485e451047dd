"""TEST INFRASTRUCTURE ONLY -- torchvision.transforms subset used by the reference
(data/data_utils.py:100 ToTensor; utils.py:46 Normalize)."""
import numpy as np
import torch


class ToTensor:
    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32)


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32)
        self.std = torch.tensor(std, dtype=torch.float32)

    def __call__(self, x):
        m = self.mean.to(x.device).view(-1, 1, 1)
        s = self.std.to(x.device).view(-1, 1, 1)
        return (x - m) / s


class ToPILImage:
    def __call__(self, x):
        raise NotImplementedError("stub")
