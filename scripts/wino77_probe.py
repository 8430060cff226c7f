#!/usr/bin/env python3
"""Timing of the 2-D Winograd form F(2x2,7x7) (csrc/conv7_wino2d.hip) against the 1-D row form F(2,7) (csrc/conv7_wino.hip) on SpyNet's layer shapes
of the fp32 path (80 frames at the pyramid's two largest levels).  With GPEMSR_LIB_PATH pointing at a variant library the 2-D column is that variant's.
    python3 scripts/wino77_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_winograd7, pack_winograd77  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timed(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (n, cin, cout, h, w) in ((80, 32, 64, 512, 512), (80, 64, 32, 512, 512), (80, 32, 64, 256, 256), (80, 64, 32, 256, 256), (80, 64, 32, 64, 64)):
    wt = (torch.rand(cout, cin, 7, 7, generator=g) * 2 - 1) / (cin * 49) ** 0.5
    pc = pack_conv(wt, torch.rand(cout), dev)
    pc.wino7 = pack_winograd7(wt, dev)
    x = ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev))
    out = ops.new_act(n, h, w, cout, device=dev)
    fl = 2.0 * n * h * w * cin * cout * 49
    t1 = timed(lambda: ops.conv2d([x], pc, 1, out=out))
    pc.wino77 = pack_winograd77(wt, dev)
    t2 = timed(lambda: ops.conv2d([x], pc, 1, out=out))
    print(f"{cin}->{cout} @{h}x{w} x{n}: F(2,7) {t1:.3f} ms ({fl * 8 / 14 / t1 / 1e9:.1f} TF executed), F(2x2,7x7) {t2:.3f} ms ({fl * 64 / 196 / t2 / 1e9:.1f} executed, "
          f"{fl / t2 / 1e9:.1f} algorithmic)", flush=True)
    del x, out
    torch.cuda.empty_cache()
