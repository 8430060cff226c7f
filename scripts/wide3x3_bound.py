#!/usr/bin/env python3
"""GPU box: an upper bound for VERDICT r4 item 1(b) -- the wide 3x3 bf16 layers on a `gemm_direct`-style kernel (256-cout wave tile, pixel
fragments straight from global memory, only the weights through an LDS ring) -- without building it: the same multiply count as a GEMM with
K = 9 cin on `gemm_direct_bf16_kernel` (no tap shifts, no border masks, no halo re-reads: everything a real 3x3 form would add is left out),
next to the ring kernel on the real 3x3 layer and hipBLASLt on that GEMM.
    python3 scripts/wide3x3_bound.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / reps)
    return sorted(ts)[2]


print(f"{'layer':34s} {'arm':>44s} {'ms':>8s} {'TFLOP/s':>9s}")
for name, n, h, w, cin, cout in (("3x3 512->512 @64^2 x80", 80, 64, 64, 512, 512), ("3x3 256->256 @128^2 x80", 80, 128, 128, 256, 256),
                                 ("3x3 512->512 @64^2 x16", 16, 64, 64, 512, 512)):
    flops = 2.0 * n * h * w * cin * cout * 9
    w3 = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    pc3 = pack_conv(w3, b, dev)
    pc3.wb = pack_conv_bf16(w3, dev)
    x16 = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev)))
    t_ring = timed(lambda: ops.conv2d([x16], pc3, 0, precision="bf16"))
    print(f"{name:34s} {'ring kernel, the real 3x3 layer':>44s} {t_ring:8.3f} {flops / (t_ring * 1e-3) / 1e12:9.1f}", flush=True)
    del x16
    k9 = 9 * cin
    w1 = (torch.rand(cout, k9, 1, 1, generator=g) * 2 - 1) / k9 ** 0.5
    pc1 = pack_conv(w1, b, dev)
    pc1.wb = pack_conv_bf16(w1, dev)
    xg = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, k9, generator=g) * 2 - 1).to(dev)))
    t_dir = timed(lambda: ops.conv2d([xg], pc1, 0, precision="bf16", force_mfma=True))
    print(f"{name:34s} {'gemm_direct, GEMM with K = 9 cin (bound)':>44s} {t_dir:8.3f} {flops / (t_dir * 1e-3) / 1e12:9.1f}", flush=True)
    os.environ["GPEMSR_GEMM_DIRECT"] = "0"
    t_rg = timed(lambda: ops.conv2d([xg], pc1, 0, precision="bf16", force_mfma=True))
    os.environ["GPEMSR_GEMM_DIRECT"] = "1"
    print(f"{name:34s} {'ring kernel, the same GEMM':>44s} {t_rg:8.3f} {flops / (t_rg * 1e-3) / 1e12:9.1f}", flush=True)
    A = xg.torch().reshape(-1, k9)
    Wt = w1.reshape(cout, k9).to(dev).to(torch.bfloat16)
    bb = b.to(dev).to(torch.bfloat16)
    t_bl = timed(lambda: torch.addmm(bb, A, Wt.t()))
    print(f"{name:34s} {'hipBLASLt, the same GEMM':>44s} {t_bl:8.3f} {flops / (t_bl * 1e-3) / 1e12:9.1f}", flush=True)
    del xg, A
    torch.cuda.empty_cache()
