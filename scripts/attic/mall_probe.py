#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache help a producer -> consumer pair of HBM-bound layers if the pair is run per group of images instead of
layer by layer over the whole batch?  conv -> ReLU -> conv (+ skip) of a 64-channel ResidualBlockNoBN on the bf16 path (weights-resident kernel),
80 images of 512^2 (33.5 MB per image and tensor) and 16 images of 1024^2 (134 MB).  python3 scripts/mall_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def layer():
    w = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / 24
    pc = pack_conv(w, torch.rand(64, generator=g) * 0.1, dev)
    pc.wb = pack_conv_bf16(w, dev, pc.splits)
    return pc


for (n, hw) in ((80, 512), (16, 1024)):
    p1, p2 = layer(), layer()
    x = ops.new_act(n, hw, hw, 64, device=dev, bf16=True)
    x.buf.view(torch.int16).random_(0, 16000, generator=None)
    t = ops.new_act(n, hw, hw, 64, device=dev, bf16=True)
    y = ops.new_act(n, hw, hw, 64, device=dev, bf16=True)
    for grp in (n, 8, 4, 2, 1):
        if grp > n:
            continue

        def run():
            for i0 in range(0, n, grp):
                xi, ti, yi = x.images(i0, grp), t.images(i0, grp), y.images(i0, grp)
                ops.conv2d([xi], p1, ops.ACT_RELU, out=ti, precision="bf16")
                ops.conv2d([ti], p2, ops.ACT_NONE, residual=xi, out=yi, precision="bf16")
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        gb = 5 * n * hw * hw * 64 * 2 / 1e9                     # x, t (write + read), x again, y
        print(f"{n} x {hw}^2, groups of {grp:2d} images ({grp * hw * hw * 128 / 2**20:.0f} MiB per tensor and group): {ms:.3f} ms per block = {gb / ms:.2f} TB/s algorithmic", flush=True)
