#!/usr/bin/env python3
"""Timing of the Winograd form of gpemsr_conv2d against the direct form on the fp32 path's layer shapes (GPEMSR_WINO_DBG variants are
timing experiments: results are wrong on purpose; the hooks exist only in a library whose conv_wino.hip was compiled with
-DGPEMSR_WINO_PROBE -- the shipped build ignores the variable).  python3 scripts/wino_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_winograd  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (n, cin, cout, h, w) in ((80, 512, 512, 64, 64), (80, 256, 256, 128, 128), (80, 64, 64, 512, 512), (16, 64, 64, 1024, 1024)):
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    pc = pack_conv(wt, torch.rand(cout), dev)
    pc.wino = pack_winograd(wt, dev)
    x = ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev))
    out = ops.new_act(n, h, w, cout, device=dev)
    res = {}
    for name, kw in (("direct", {}), ("winograd", {"winograd": True})):
        for _ in range(2):
            ops.conv2d([x], pc, 1, out=out, **kw)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            ops.conv2d([x], pc, 1, out=out, **kw)
        e.record(); torch.cuda.synchronize()
        res[name] = s.elapsed_time(e) / 3
    fl = 2.0 * n * h * w * cin * cout * 9
    print(f"{cin}->{cout} @{h}x{w} x{n}: direct {res['direct']:.3f} ms ({fl / res['direct'] / 1e9:.1f} TF), winograd {res['winograd']:.3f} ms "
          f"({fl / res['winograd'] / 1e9:.1f} TF algorithmic, {fl * 16 / 36 / res['winograd'] / 1e9:.1f} executed)", flush=True)
