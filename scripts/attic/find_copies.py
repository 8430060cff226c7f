#!/usr/bin/env python3
"""GPU box: which Python lines launch device-to-device copies (aten::copy_ and friends) during one bf16 forward of 16 windows."""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
x = synth_lr_tiles(16, 5, 128, 128, seed=1000, kind="uniform").to(dev)
m = build_model(opt, load_prior_files=False, precision="bf16").eval().to(dev)
m(x); torch.cuda.synchronize()
counts = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        st = [f for f in traceback.extract_stack() if "gpemsr_amd" in f.filename]
        where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno}" if st else "?"
        counts[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    m(x)
torch.cuda.synchronize()
for (name, where), n in counts.most_common(40):
    print(f"{n:5d}  {name:45s} {where}")
