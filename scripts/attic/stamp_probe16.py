#!/usr/bin/env python3
"""Diagnostic build only (GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_stamp.so, built with -DGP16_STAMP): where wave 0 of a
persistent conv_bf16 workgroup spends its cycles, summed over its tiles (s_memtime buckets; shares matter, not lengths):
0 set-up + prologue issue | 1 first wait | 2 stage compute (LDS reads + MFMA) | 3 end-of-stage wait + barrier |
4 refill issue | 5 epilogue."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import _abi, ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16, pack_convT, pack_convT_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (n, cin, cout, k, h, w, bias) in ((80, 128, 64, -3, 256, 256, True), (80, 256, 128, -3, 128, 128, True), (80, 64, 64, -3, 256, 256, True), (80, 128, 128, 3, 256, 256, True), (80, 256, 256, 3, 128, 128, True), (80, 512, 512, 3, 64, 64, True), (80, 512, 512, 1, 64, 64, True), (16, 32, 64, 7, 512, 512, True), (16, 32, 16, 7, 512, 512, True)):
    convT = k < 0
    k = abs(k)
    wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
    if convT:
        wT = wt.permute(1, 0, 2, 3).contiguous()
        pc = pack_convT(wT, torch.rand(cout), dev)
        pc.wb = pack_convT_bf16(wT, dev)
    else:
        pc = pack_conv(wt, torch.rand(cout) if bias else None, dev)
        pc.wb = pack_conv_bf16(wt, dev)
    x = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev)))
    for _ in range(3):
        out = ops.conv2d([x], pc, 1, precision="bf16", force_mfma=True)
    torch.cuda.synchronize()
    lib = _abi.load()
    nb = 256
    buf = (C.c_ulonglong * (8 * nb))()
    lib.gpemsr_debug_read_xstamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.gpemsr_debug_read_xstamps(buf, nb) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
    tiles = st[:, 6]
    tot = st[:, :6].sum(axis=1)
    names = ["setup+prologue issue", "first wait", "stage compute", "stage wait+barrier", "refill issue", "epilogue"]
    import time as _t
    torch.cuda.synchronize(); t0 = _t.perf_counter()
    for _ in range(5):
        out = ops.conv2d([x], pc, 1, precision="bf16", force_mfma=True)
    torch.cuda.synchronize(); ms = (_t.perf_counter() - t0) / 5 * 1e3
    fl = 2.0 * n * h * w * cin * cout * (9 if convT else k * k)
    print(f"{'convT ' if convT else ''}{cin}->{cout} k{k} @{h}x{w} x{n}: {ms:.3f} ms = {fl / ms / 1e9:.0f} TFLOP/s; tiles per workgroup {tiles.mean():.1f}; cycles per tile {tot.mean() / tiles.mean():.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:24s} {st[:, i].mean() / tiles.mean():9.0f} cycles/tile  {100 * st[:, i].sum() / tot.sum():5.1f} %")
