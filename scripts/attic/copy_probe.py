#!/usr/bin/env python3
"""Which Python lines of the bf16 forward issue device copies / fills through torch (they show up as __amd_rocclr_copyBuffer and
FillFunctor launches between the library's kernels)?  One profiled forward, aten::copy_ / fill_ / zero_ grouped by caller line."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
opt = load_options(os.path.join(root, "option", "output_GPEMSR_x8.yml"))
model = build_model(opt, load_prior_files=False, precision=prec).eval().to(torch.device("cuda", 0))
x = synth_lr_tiles(16, 5, 128, 128, seed=0, kind="uniform").cuda()
model(x); model(x); torch.cuda.synchronize()

counts = collections.Counter()
def hook(name, orig):
    def f(*a, **k):
        st = traceback.extract_stack(limit=6)
        where = " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st[:-1]) if "gpemsr_amd" in s.filename or "bench" in s.filename)
        counts[(name, where)] += 1
        return orig(*a, **k)
    return f
for nm in ("zeros", "empty", "full", "tensor", "zeros_like"):
    setattr(torch, nm, hook("torch." + nm, getattr(torch, nm)))
for nm in ("copy_", "clone", "contiguous", "zero_", "fill_", "to"):
    setattr(torch.Tensor, nm, hook("Tensor." + nm, getattr(torch.Tensor, nm)))
model(x); torch.cuda.synchronize()
for (name, where), c in counts.most_common(40):
    print(f"{c:5d}  {name:18s} {where}")
