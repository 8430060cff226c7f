"""Samples for the stage-3 training / validation loop, in the layout the reference's dataset hands to its loop
(R:data/CREMI_dataset.py:52-107: ``{'LQ': float32 [N,1,h,w], 'GT': float32 [1,sh,sw]}`` in [0,1]).

Two sources: ``FolderVolumes`` reads the reference's folder contract (``dataroot/<volume>/<k>.png`` for GT and LQ, the N-slice LQ
window centred on the GT slice, a missing -- damaged -- LQ slice replaced by the nearest earlier one, R:data/CREMI_dataset.py:112-125;
training: aligned random crop, flips / transposition) and ``SyntheticCrops`` gives seeded smooth EM-like crops when the data is absent
(the authors' CREMI crops are not redistributable).  Deliberately small: datasets are outside SURVEY section 8's hot path."""
from __future__ import annotations

import os
import random
from typing import Dict, List

import numpy as np
import torch

from .synth import synth_lr_tiles


def _read_gray(path: str) -> np.ndarray:
    try:
        import cv2
        img = cv2.imread(path, cv2.IMREAD_UNCHANGED)
    except ImportError:
        from PIL import Image
        img = np.array(Image.open(path))
    img = img.astype(np.float32) / 255.0
    return img[..., 0] if img.ndim == 3 else img


class FolderVolumes(torch.utils.data.Dataset):
    def __init__(self, opt: Dict, scale: int, train: bool):
        self.opt, self.scale, self.train = opt, scale, train
        self.N = int(opt["N_frames"])
        half = (self.N - 1) // 2
        self.items: List[tuple] = []
        for vol in sorted(os.listdir(opt["dataroot_GT"])):
            d = os.path.join(opt["dataroot_GT"], vol)
            ks = sorted(int(f[:-4]) for f in os.listdir(d) if f.endswith(".png"))
            for k in ks[2 * half:len(ks) - 2 * half]:            # the margin the reference leaves at both ends of a volume
                self.items.append((vol, k))

    def __len__(self):
        return len(self.items)

    def _lq_path(self, vol: str, k: int) -> str:
        d = os.path.join(self.opt["dataroot_LQ"], vol)
        while k >= 0 and not os.path.exists(os.path.join(d, f"{k}.png")):
            k -= 1                                               # damaged slice: the nearest undamaged one before it
        return os.path.join(d, f"{k}.png")

    def __getitem__(self, i):
        vol, k = self.items[i]
        half = (self.N - 1) // 2
        gt = _read_gray(os.path.join(self.opt["dataroot_GT"], vol, f"{k}.png"))
        lq = np.stack([_read_gray(self._lq_path(vol, k + o)) for o in range(-half, half + 1)])
        if self.train:
            s, g = self.scale, int(self.opt["GT_size"])
            ls = g // s
            y = random.randint(0, max(0, lq.shape[1] - ls)); x = random.randint(0, max(0, lq.shape[2] - ls))
            lq, gt = lq[:, y:y + ls, x:x + ls], gt[y * s:y * s + g, x * s:x * s + g]
            # R:data/util.py:166-170 (`augment`): the vertical flip is gated on use_rot, not use_flip; a random number is drawn only
            # when its flag is set (short-circuit `and`); order hflip, vflip, rot90
            hflip = bool(self.opt.get("use_flip")) and random.random() < 0.5
            vflip = bool(self.opt.get("use_rot")) and random.random() < 0.5
            rot90 = bool(self.opt.get("use_rot")) and random.random() < 0.5
            if hflip:
                lq, gt = lq[:, :, ::-1], gt[:, ::-1]
            if vflip:
                lq, gt = lq[:, ::-1, :], gt[::-1, :]
            if rot90:
                lq, gt = lq.transpose(0, 2, 1), gt.T
        return {"LQ": torch.from_numpy(np.ascontiguousarray(lq)).float().unsqueeze(1), "GT": torch.from_numpy(np.ascontiguousarray(gt)).float().unsqueeze(0)}


class SyntheticCrops(torch.utils.data.Dataset):
    """Seeded smooth crops: LQ [N,1,l,l]; GT = bicubic-free stand-in (bilinear x scale of the centre slice + detail noise)."""

    def __init__(self, n: int, scale: int, lq_size: int, n_frames: int = 5, seed: int = 0):
        self.n, self.scale, self.l, self.N, self.seed = n, scale, lq_size, n_frames, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        lq = synth_lr_tiles(1, self.N, self.l, self.l, seed=self.seed + i, kind="smooth")[0]
        g = torch.Generator().manual_seed(self.seed + 7919 * (i + 1))
        up = torch.nn.functional.interpolate(lq[self.N // 2:self.N // 2 + 1], scale_factor=self.scale, mode="bilinear", align_corners=False)[0]
        gt = (up + 0.05 * torch.randn(up.shape, generator=g)).clamp(0, 1)
        return {"LQ": lq, "GT": gt}


def make_dataset(ds_opt: Dict, scale: int, train: bool, synthetic_if_missing: bool, n_synth: int = 64, seed: int = 0):
    if os.path.isdir(str(ds_opt.get("dataroot_GT"))) and os.path.isdir(str(ds_opt.get("dataroot_LQ"))):
        return FolderVolumes(ds_opt, scale, train)
    if not synthetic_if_missing:
        raise FileNotFoundError(f"dataset folders {ds_opt.get('dataroot_GT')} / {ds_opt.get('dataroot_LQ')} not found")
    lq = int(ds_opt.get("LQ_size") or 32) if train else 2 * int(ds_opt.get("LQ_size") or 16)
    return SyntheticCrops(n_synth if train else 4, scale, lq, int(ds_opt.get("N_frames", 5)), seed)
