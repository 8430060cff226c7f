#!/usr/bin/env python3
"""Diagnostic build only (GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_stamp.so, -DGP16_STAMP): where a conv_bf16 workgroup spends
its lifetime.  s_memtime stamps: 0 entry, 1 prologue issued, 2 first stage ready, 3 main loop done, 4 epilogue done."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import _abi, ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (n, cin, cout, k, h, w) in ((80, 64, 64, 3, 128, 128), (80, 256, 256, 3, 128, 128), (80, 512, 512, 1, 64, 64)):
    wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
    pc = pack_conv(wt, None, dev)
    pc.wb = pack_conv_bf16(wt, dev)
    x = ops.cast_bf16(ops.from_nhwc((torch.rand(n, h, w, cin, generator=g) * 2 - 1).to(dev)))
    for _ in range(3):
        out = ops.conv2d([x], pc, 1, precision="bf16", force_mfma=True)
    torch.cuda.synchronize()
    lib = _abi.load()
    nb = min(65536, n * ((h + 7) // 8) * ((w + 31) // 32) * ((cout + 127) // 128 if cout > 64 else 1))
    buf = (C.c_ulonglong * (8 * nb))()
    lib.gpemsr_debug_read_xstamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.gpemsr_debug_read_xstamps(buf, nb) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
    d = np.diff(st[:, :5], axis=1)
    life = st[:, 4] - st[:, 0]
    span = st[:, 4].max() - st[:, 0].min()
    print(f"{cin}->{cout} k{k} @{h}x{w} x{n}: {nb} workgroups; kernel span {span} cycles (memtime @100MHz? see below)")
    print("  mean cycles: setup+issue %.0f | wait first stage %.0f | main loop %.0f | epilogue %.0f | lifetime %.0f" %
          (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), life.mean()))
    print("  median      : %.0f | %.0f | %.0f | %.0f | %.0f" % tuple(np.median(np.concatenate([d, life[:, None]], axis=1), axis=0)))
    print("  sum of lifetimes / span = %.1f workgroups resident on average (512 slots)" % (life.sum() / span))
