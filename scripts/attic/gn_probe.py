#!/usr/bin/env python3
"""Times gpemsr_groupnorm_apply_bf16 (+ its statistics pass) on the two largest GroupNorm tensors of the 8x forward at batch 16."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import ops
dev = torch.device("cuda", 0)
for (n, h, w, c) in ((80, 512, 512, 64), (80, 256, 256, 128), (80, 64, 64, 512)):
    x = ops.cast_bf16(ops.from_nhwc(torch.rand(n, h, w, c, device=dev)))
    g, b = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev)
    for _ in range(2):
        y = ops.groupnorm_relu(x, g, b, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = ops.groupnorm_relu(x, g, b, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = n * h * w * c * 2 * 3 / 1e9          # stats read + apply read + apply write
    print(f"{n}x{h}x{w}x{c}: {ms:.3f} ms for stats + apply, {gb / ms:.2f} TB/s over 3 passes")
    del x, y
