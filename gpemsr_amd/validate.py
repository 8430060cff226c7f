"""Validation pass of the stage-3 training loop (R:train_stage3.py:199-317): every validation sample is super-resolved in four
quadrant crops ("We crop patches during inference time to prevent insufficient memory"), converted with the reference's
``tensor2img`` and scored with its ``calculate_psnr`` on the uint8 images; with several ranks sample ``idx`` goes to rank
``idx % world`` and the per-sample PSNR vector is summed onto rank 0 (``dist.reduce``, :254) before the mean.

Host logic only -- the model call is the device path.  ``model`` is anything with ``model(x[1,N,1,h,w]) -> (sr[1,1,sh,sw], _)``.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .imgutil import calculate_psnr, tensor2img


def sr_by_quadrants(model: Callable, LQ: torch.Tensor, scale: int, device) -> np.ndarray:
    """R:train_stage3.py:222-250: LQ [1,N,C,H,W] -> uint8 SR image [sH, sW] assembled from the four H/2 x W/2 crops."""
    B, N, C, H, W = LQ.shape
    assert B == 1
    hs, ws = H // 2, W // 2
    SR = np.zeros((H * scale, W * scale), dtype=np.uint8)
    for (y0, y1), (x0, x1) in (((0, hs), (0, ws)), ((0, hs), (ws, W)), ((hs, H), (0, ws)), ((hs, H), (ws, W))):
        sr, _ = model(LQ[:, :, :, y0:y1, x0:x1].to(device))
        SR[y0 * scale:y1 * scale, x0 * scale:x1 * scale] = tensor2img(sr.detach().cpu())
    return SR


def validate_psnr(model: Callable, val_set: Sequence, scale: int, device, rank: int = 0, world: int = 1,
                  save_dir: Optional[str] = None, save_first: int = 20, save_img: Optional[Callable] = None) -> Optional[float]:
    """Mean PSNR over ``val_set`` (items ``{'GT': [1,sH,sW], 'LQ': [N,1,H,W]}``, the reference dataset's layout).  Returns the
    mean on rank 0 and None on the other ranks (the reference logs on rank 0 only)."""
    n = len(val_set)
    psnr = torch.zeros(n, dtype=torch.float32, device=device if world > 1 and torch.distributed.get_backend() == "nccl" else "cpu")
    gt_dir = fake_dir = None
    if save_dir is not None:
        gt_dir, fake_dir = os.path.join(save_dir, "gt"), os.path.join(save_dir, "fake_reference")
        if rank == 0:
            os.makedirs(gt_dir, exist_ok=True)
            os.makedirs(fake_dir, exist_ok=True)
        if world > 1:
            torch.distributed.barrier()
    with torch.no_grad():
        for idx in range(rank, n, world):
            item = val_set[idx]
            gt, LQ = item["GT"].unsqueeze(0), item["LQ"].unsqueeze(0)
            SR = sr_by_quadrants(model, LQ, scale, device)
            GT = tensor2img(gt)
            psnr[idx] = float(calculate_psnr(GT, SR))
            if save_dir is not None and idx < save_first and save_img is not None:
                save_img(GT, os.path.join(gt_dir, f"{idx}.png"))
                save_img(SR, os.path.join(fake_dir, f"{idx}.png"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.reduce(psnr, 0)
    return float(psnr.mean().item()) if rank == 0 else None
