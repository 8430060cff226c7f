#!/usr/bin/env python3
"""One shape of gpemsr_conv2d_bf16, launched a few times (target of `rocprofv3 --pmc ...`): python3 scripts/pmc_conv16.py cin cout k n h w [variant]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

cin, cout, k, n, h, w = [int(v) for v in sys.argv[1:7]]
variant = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
pc = pack_conv(wt, torch.rand(cout, generator=g) - 0.5, dev)
pc.wb = pack_conv_bf16(wt, dev)
x = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev)))
for _ in range(5):
    ops.conv2d([x], pc, 1, precision="bf16", variant=variant, force_mfma=True)
torch.cuda.synchronize()
