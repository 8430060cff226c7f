#!/usr/bin/env python3
"""Timing of the F(4x4,3x3) Winograd form (csrc/conv_wino4.hip) against F(2x2) on the many-channel layer shapes of the fp32 path.  With
GPEMSR_LIB_PATH pointing at a library from scripts/build_wino_probe_lib.sh the F(4x4) column is that variant's (single phases compiled out).
    python3 scripts/wino4_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_winograd, pack_winograd4  # noqa: E402

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


SHAPES = ((80, 512, 512, 64, 64), (80, 256, 256, 128, 128), (16, 128, 128, 256, 256), (16, 192, 64, 512, 512))
if "--c64" in sys.argv:                                   # the 64-channel layers (8 chunks per tile): where a tile's fixed cost shows
    SHAPES = ((16, 64, 64, 1024, 1024), (80, 64, 64, 128, 128))
for (n, cin, cout, h, w) in SHAPES:
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    pc = pack_conv(wt, torch.rand(cout), dev)
    pc.wino = pack_winograd(wt, dev)
    x = ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev))
    out = ops.new_act(n, h, w, cout, device=dev)
    fl = 2.0 * n * h * w * cin * cout * 9
    t2 = timed(lambda: ops.conv2d([x], pc, 0, out=out, winograd=True))
    pc.wino4 = pack_winograd4(wt, dev)
    t4 = timed(lambda: ops.conv2d([x], pc, 0, out=out, winograd=True))
    line = (f"{cin}->{cout} @{h}x{w} x{n}: F(2x2) {t2:.3f} ms ({fl * 16 / 36 / t2 / 1e9:.1f} TF executed), F(4x4) {t4:.3f} ms ({fl / 4 / t4 / 1e9:.1f} executed, "
            f"{fl / t4 / 1e9:.1f} algorithmic)")
    print(line, flush=True)
