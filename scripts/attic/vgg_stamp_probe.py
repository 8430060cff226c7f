#!/usr/bin/env python3
"""Diagnostic build only: GPEMSR_LIB_PATH=gpemsr_amd/lib/libgpemsr_vstamp.so (vgg_mask.hip compiled with -DGPVGG_STAMP).  Where the waves of
vgg_mask2_kernel spend their cycles (s_memtime buckets per wave, summed over the workgroup's life; per item = / 2560 at batch 16).
  producers (waves 8-15): 0 window fetch issue | 4 wait for the loads | 1 convert + conv1_1 + writes | 2 barrier wait | 3 other
  multipliers (waves 0-7): 0 conv1_2 MFMA loop | 1 feature / patch-sum epilogue | 2 barrier wait | 3 other"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd import _abi, ops
from gpemsr_amd.packing import pack_conv_bf16
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
n, h, w, s = 80, 128, 128, 8
ref = torch.rand(n, 1, h * s, w * s, generator=g).to(dev); lr = torch.rand(n, 1, h, w, generator=g).to(dev)
w1 = (torch.rand(64, 3, 3, 3, generator=g) - 0.5).to(dev); b1 = torch.rand(64, generator=g).to(dev)
w2 = (torch.rand(64, 64, 3, 3, generator=g) - 0.5) / 24; b2 = torch.rand(64, generator=g).to(dev)
w2b = pack_conv_bf16(w2, dev)
A = lambda t: ops.Act(t.reshape(-1), t.shape[0], t.shape[2], t.shape[3], 1, 1, 0)
w1s = w1.sum(1).reshape(64, 9).contiguous()
up = ops.bilinear(A(lr), h * s, w * s)
for _ in range(3):
    out = ops.vgg_mask_bf16(A(ref), up, 1, w1s, b1, w2b, b2) if os.environ.get("GPEMSR_VGG_UPLR", "1") != "0" else ops.vgg_mask_bf16(A(ref), A(lr), s, w1s, b1, w2b, b2)
torch.cuda.synchronize()
lib = _abi.load()
buf = (C.c_ulonglong * (256 * 16 * 8))()
lib.gpemsr_debug_read_vstamps.argtypes = [C.c_void_p]
assert lib.gpemsr_debug_read_vstamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 8).astype(np.float64)
items = 4 * 80 * 32 * 64 / 256
for role, sl, names in (("multiplier waves 0-7", slice(0, 8), ["MFMA loop", "epilogue", "barrier wait", "other"]),
                        ("producer waves 8-15", slice(8, 16), ["fetch issue", "convert+conv1_1+write", "barrier wait", "other", "load wait"])):
    x = st[:, sl, :len(names)]
    tot = x.sum(axis=2).mean()
    print(f"{role}: {tot / items:.0f} cycles per item")
    for i, nm in enumerate(names):
        print(f"   {nm:24s} mean {x[:, :, i].mean() / items:8.0f}  per wave: " + " ".join(f"{v / items:6.0f}" for v in x[:, :, i].mean(axis=0)))
