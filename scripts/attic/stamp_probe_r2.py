#!/usr/bin/env python3
"""Diagnostic build only (bash scripts/build_stamp_lib.sh; GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_stamp.so): cycles of
conv64_resident2_kernel per tile, wave 0 (group 0) and loader wave 8:
   multiplying wave: chunk compute | barrier after a chunk | epilogue | barrier after the epilogue
   loader wave     : issue | wait for the DMA to land | barrier"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import _abi, ops  # noqa: E402
from gpemsr_amd.ops import ACT_NONE, ACT_RELU  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
lib = _abi.load()
lib.gpemsr_debug_read_xstamps.argtypes = [C.c_void_p, C.c_int]
VAR = {"relu-nomfma": 101, "relu-noepi": 102, "relu-noload": 103, "relu-noload-noepi": 104}
for (n, h, w, mode) in ((80, 512, 512, "relu"), (80, 512, 512, "relu-nomfma"), (80, 512, 512, "relu-noepi"), (80, 512, 512, "relu-noload"), (80, 512, 512, "relu-noload-noepi"), (80, 512, 512, "residual"), (80, 512, 512, "gn"), (80, 128, 128, "relu")):
    wt = (torch.rand(64, 64, 3, 3, generator=g) * 2 - 1) / (64 * 9) ** 0.5
    pc = pack_conv(wt, torch.rand(64), dev)
    pc.wb = pack_conv_bf16(wt, dev)
    x = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, 64, generator=g) * 2 - 1).to(dev)))
    kw = {}
    if mode in VAR:
        kw["variant"] = VAR[mode]
    if mode == "residual":
        kw["residual"] = x
    if mode == "gn":
        kw["gn_stats"] = True
    act = ACT_RELU if mode.startswith("relu") else ACT_NONE
    for _ in range(2):
        out = ops.conv2d([x], pc, act, precision="bf16", **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = ops.conv2d([x], pc, act, precision="bf16", **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    nb = 2048
    buf = (C.c_ulonglong * (8 * nb))()
    assert lib.gpemsr_debug_read_xstamps(buf, nb) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
    m, l = st[:256], st[1024:1280]
    tiles = m[:, 6].mean() / 2          # tiles of group 0 per workgroup
    flops = 2 * 9 * 64 * 64 * n * h * w
    print(f"64->64 3x3 @{h}x{w} x{n} {mode}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s, {(2 * n * h * w * 128 + (n * h * w * 128 if mode == 'residual' else 0)) / ms / 1e9:.2f} TB/s; tiles per group {tiles:.1f}")
    names = {2: "chunk compute (x2 per tile)", 3: "barrier after chunk", 5: "epilogue", 4: "barrier after epilogue"}
    tot = sum(m[:, k].mean() for k in names)
    for k, nm in names.items():
        print(f"   mult wave 0   {nm:30s} {m[:, k].mean() / tiles:9.0f} cycles/tile {100 * m[:, k].mean() / tot:5.1f} %")
    iv = l[:, 6].mean()
    ltot = sum(l[:, k].mean() for k in range(3))
    for k, nm in enumerate(("issue", "wait for DMA to land", "barrier")):
        print(f"   loader wave 8 {nm:30s} {l[:, k].mean() / iv:9.0f} cycles/interval {100 * l[:, k].mean() / ltot:5.1f} %")
