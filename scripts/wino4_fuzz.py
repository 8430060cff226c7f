#!/usr/bin/env python3
"""GPU box: screen of the F(4x4) Winograd form (csrc/conv_wino4.hip) for races and addressing mistakes -- random shapes (ragged tiles, several
sources, every epilogue flavour) against the direct kernel and run-to-run bit-stability; then the whole fp32 forward repeated bit for bit.
    python3 scripts/wino4_fuzz.py [cases]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.config import build_model, load_options  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_winograd, pack_winograd4  # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles  # noqa: E402

dev = torch.device("cuda", 0)
rng = random.Random(7)
g = torch.Generator().manual_seed(7)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(ncases):
    nsrc = rng.choice([1, 1, 1, 2, 3])
    cins = [8 * rng.randint(1, 24 if nsrc == 1 else 8) for _ in range(nsrc)]
    cout = rng.choice([64, 64, 128, 192, 256, 72, 216])
    n, h, w = rng.randint(1, 4), rng.randint(5, 70), rng.randint(9, 140)
    act = rng.choice([0, 1, 2])
    flavour = rng.choice(["plain", "plain", "res", "resmul", "gn"]) if cout % 64 == 0 else rng.choice(["plain", "res"])
    if flavour == "gn":
        act = 0
    cin = sum(cins)
    x = torch.rand(n, h, w, cin, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    pc = pack_conv(wt, b, dev, tuple(cins))
    if cout % 32 == 0:
        pc.wino = pack_winograd(wt, dev)
    pc.wino4 = pack_winograd4(wt, dev)
    srcs, o = [], 0
    for c in cins:
        srcs.append(ops.from_nhwc(x[..., o:o + c].contiguous().to(dev))); o += c
    res = ops.from_nhwc((torch.rand(n, h, w, cout, generator=g) - 0.5).to(dev)) if flavour in ("res", "resmul") else None
    mul = ops.Act((torch.rand(n * h * w, generator=g) + 0.5).to(dev), n, h, w, 1, 1, 0) if flavour == "resmul" else None
    kw = dict(residual=res, pixmul=mul)
    if flavour == "gn":
        kw = dict(gn_stats=True)
    got = ops.conv2d(srcs, pc, act, winograd=True, **kw)
    first = got.nchw().clone()
    sums = got.gn[0].clone() if got.gn is not None else None
    ref = ops.conv2d(srcs, pc, act, **({} if flavour == "gn" else kw)).nchw()
    err = float((first - ref).abs().max() / ref.abs().max())
    stable = True
    for _ in range(3):
        again = ops.conv2d(srcs, pc, act, winograd=True, **kw)
        stable = stable and torch.equal(again.nchw(), first) and (sums is None or torch.equal(again.gn[0], sums))
    ok = err < 6e-5 and stable
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} n={n} cins={cins} cout={cout} {h}x{w} act={act} {flavour:7s} err {err:.2e} stable {stable}", flush=True)
print(f"{ncases - bad} / {ncases} cases ok", flush=True)

opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
m = build_model(opt, load_prior_files=False, precision="fp32").eval().to(dev)
xb = synth_lr_tiles(16, 5, 128, 128, seed=1000, kind="uniform").to(dev)
first = m(xb)[0].clone()
same = 0
for r in range(8):
    same += int(torch.equal(m(xb)[0], first))
torch.cuda.synchronize()
print(f"whole fp32 forward (16 windows): {same} / 8 repeats bit-identical", flush=True)
sys.exit(1 if bad or same != 8 else 0)
