#!/usr/bin/env python3
"""GPU box: gpemsr_conv2d_wgrad micro-benchmark (algorithmic TFLOP/s incl. the partial-sum reduce) on training shapes."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd import ops
dev = torch.device("cuda", 0)
SHAPES = [  # n, h, w, cin, cout, k, stride
    (8, 64, 64, 512, 512, 3, 1), (8, 128, 128, 256, 256, 3, 1), (40, 32, 32, 64, 64, 3, 1), (8, 256, 256, 64, 64, 3, 1),
    (40, 128, 128, 128, 64, 3, 1), (40, 32, 32, 320, 64, 1, 1), (40, 64, 64, 64, 64, 3, 2), (8, 32, 32, 64, 256, 3, 1),
]
for n, h, w, cin, cout, k, s in SHAPES:
    p = k // 2
    oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = ops.from_nhwc(torch.randn(n, h, w, cin, device=dev))
    dz = ops.from_nhwc(torch.randn(n, oh, ow, cout, device=dev))
    dw = torch.zeros(cout, cin, k, k, device=dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, dz, k, s, dw, cin, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad(x, dz, k, s, dw, cin, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * n * oh * ow * cin * cout * k * k
    print(f"n{n} {h}x{w} {cin}->{cout} k{k} s{s}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
