"""Host-side image helpers with the reference's semantics (util/util.py:139-163, 253-260)."""
from __future__ import annotations

import math

import numpy as np
import torch


def tensor2img(tensor: torch.Tensor, min_max=(0, 1)) -> np.ndarray:
    """[.., H, W] float tensor -> uint8 HxW (clamp, scale, round-half-even).  Device tensors are
    converted by the HIP kernel; CPU tensors by numpy (same arithmetic)."""
    t = tensor.squeeze().float()
    if t.is_cuda and min_max == (0, 1):
        from . import ops
        return ops.tensor2img_u8(t).cpu().numpy()
    t = t.cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    return (t.numpy() * 255.0).round().astype(np.uint8)


def calculate_psnr(img1: np.ndarray, img2: np.ndarray) -> float:
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))
