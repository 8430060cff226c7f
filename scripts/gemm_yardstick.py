#!/usr/bin/env python3
"""GPU box: the 1x1 / Linear products of the bf16 path (gpemsr_conv2d_bf16, GEMM form) next to hipBLASLt through torch.matmul on the same
operands -- a yardstick for what a plain bf16 GEMM of that shape reaches on this device (the product path does not call it).
    python3 scripts/gemm_yardstick.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
SHAPES = [("1x1 512->512 @64^2 x80", 80, 64, 64, 512, 512), ("1x1 512->1024 @64^2 x80", 80, 64, 64, 512, 1024), ("1x1 1536->1024 @64^2 x80", 80, 64, 64, 1536, 1024),
          ("1x1 256->512 @128^2 x80 (channel_up)", 80, 128, 128, 256, 512), ("1x1 576->64 @128^2 x80 (dcn)", 80, 128, 128, 576, 64), ("1x1 320->64 @128^2 x16", 16, 128, 128, 320, 64)]
print(f"{'shape':42s} {'arm':>12s} {'ms':>8s} {'TFLOP/s':>9s}")
for name, n, h, w, cin, cout in SHAPES:
    wt = (torch.rand(cout, cin, 1, 1, generator=g) * 2 - 1) / cin ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    pc = pack_conv(wt, b, dev)
    pc.wb = pack_conv_bf16(wt, dev)
    x16 = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev)))
    A = x16.torch().reshape(-1, cin)
    Wt = wt.reshape(cout, cin).to(dev).to(torch.bfloat16)
    bb = b.to(dev).to(torch.bfloat16)
    flops = 2.0 * n * h * w * cin * cout
    arms = {"ours": lambda: ops.conv2d([x16], pc, 0, precision="bf16", force_mfma=True),
            "hipblaslt": lambda: torch.addmm(bb, A, Wt.t())}
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
    for a in arms:
        ms = sorted(times[a])[2]
        print(f"{name:42s} {a:>12s} {ms:8.3f} {flops / (ms * 1e-3) / 1e12:9.1f}", flush=True)
    del x16, A, pc
    torch.cuda.empty_cache()
