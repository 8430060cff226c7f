#!/usr/bin/env python3
"""GPU box: the DCN of the bf16 path at the forward's sizes -- one kernel (gpemsr_dcn_conv_bf16) against columns + 1x1 product.
    python3 scripts/dcn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv_bf16, pack_dcn, pack_dcn_rows_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
wt = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) * 0.05
b = torch.rand(64, generator=g) - 0.5
pc = pack_dcn(wt, b, dev)
pc.wb = pack_conv_bf16(wt.permute(0, 2, 3, 1).reshape(64, -1, 1, 1), dev)
pc.wrows = pack_dcn_rows_bf16(wt, dev)
print(f"{'level':24s} {'arm':>22s} {'ms':>8s} {'GB/s (om + x + out)':>20s}")
for n, h in ((80, 128), (80, 64), (80, 32)):
    x = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, h, 64, generator=g) * 2 - 1).to(dev)))
    om = ops.from_nhwc(((torch.rand(n, h, h, 216, generator=g) * 2 - 1) * 2).to(dev))
    arms = {"fused": lambda: ops.dcn_conv_bf16(x, om, pc, 2), "columns + product": lambda: ops.conv2d([ops.dcn_columns(x, om, 8)], pc, 2, precision="bf16")}
    times = {a: [] for a in arms}
    for fn in arms.values():
        fn()
    torch.cuda.synchronize()
    for _ in range(5):
        for a, fn in arms.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record(); torch.cuda.synchronize()
            times[a].append(s.elapsed_time(e) / 5)
    nb = n * h * h * (216 * 4 + 128 + 128)
    for a in arms:
        ms = sorted(times[a])[2]
        print(f"{n} x {h}^2{'':14s} {a:>22s} {ms:8.3f} {nb / 1e9 / (ms * 1e-3):20.0f}", flush=True)
