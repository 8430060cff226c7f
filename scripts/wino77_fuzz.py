#!/usr/bin/env python3
"""GPU box: screen of the 2-D Winograd form F(2x2,7x7) (csrc/conv7_wino2d.hip) for races and addressing mistakes -- random shapes (ragged tiles,
one to eight chunks, one to three cout blocks, fewer tiles than workgroups and many tiles per persistent workgroup, strided outputs) against the
direct kernel, and run-to-run bit-stability.
    python3 scripts/wino77_fuzz.py [cases]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_winograd77  # noqa: E402

dev = torch.device("cuda", 0)
rng = random.Random(11)
g = torch.Generator().manual_seed(11)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(ncases):
    cin = 8 * rng.randint(1, 8)
    cout = rng.choice([32, 32, 64, 64, 96, 16, 16, 48])
    big = rng.random() < 0.25
    n = rng.randint(1, 6 if big else 3)
    h, w = (rng.randint(60, 200), rng.randint(60, 260)) if big else (rng.randint(6, 70), rng.randint(12, 140))
    if 3 * h * w < 2 * (-(-h // 8) * 8) * (-(-w // 16) * 16):
        continue
    act = rng.choice([0, 1, 2])
    x = torch.rand(n, h, w, cin, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 7, 7, generator=g) * 2 - 1) / (cin * 49) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    pc = pack_conv(wt, b, dev)
    xa = ops.from_nhwc(x.to(dev))
    ref = None
    if cout in (16, 32, 64):
        ref = ops.conv2d([xa], pc, act, direct7=True).nchw().clone()
    else:                                                    # (the direct 7x7 kernel stops at 64 couts: pieces)
        halves, step = [], (32 if cout % 32 == 0 else 16)
        for lo in range(0, cout, step):
            pch = pack_conv(wt[lo:lo + step], b[lo:lo + step], dev)
            halves.append(ops.conv2d([xa], pch, act, direct7=True).nchw())
        ref = torch.cat(halves, 1)
    pc.wino77 = pack_winograd77(wt, dev)
    pad = rng.choice([0, 0, 16, 24])
    off = 8 if pad else 0
    out = ops.Act(torch.full((n, h, w, cout + pad), 3.0, device=dev), n, h, w, cout, cout + pad, off)
    assert ops.winograd77_ok([xa], pc, out=out, act=act)
    ops.conv2d([xa], pc, act, out=out)
    first = out.nchw().clone()
    err = float((first - ref).abs().max() / ref.abs().max())
    untouched = pad == 0 or (float((out.buf[..., :off] - 3.0).abs().max()) == 0.0 and float((out.buf[..., off + cout:] - 3.0).abs().max()) == 0.0)
    stable = True
    for _ in range(3):
        again = ops.Act(torch.full((n, h, w, cout + pad), 3.0, device=dev), n, h, w, cout, cout + pad, off)
        ops.conv2d([xa], pc, act, out=again)
        stable = stable and torch.equal(again.nchw(), first)
    ok = err < 4e-5 and stable and untouched
    bad += not ok
    tiles = n * -(-h // 8) * -(-w // 16) * (cout // (32 if cout % 32 == 0 else 16))
    print(f"{'ok ' if ok else 'BAD'} n={n} cin={cin} cout={cout} {h}x{w} act={act} ld={cout + pad}+{off} tiles={tiles} err {err:.2e} stable {stable} slice-only {untouched}", flush=True)
print(f"{bad} bad cases", flush=True)
sys.exit(1 if bad else 0)
