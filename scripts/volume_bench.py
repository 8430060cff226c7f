#!/usr/bin/env python3
"""GPU box: volume-mode throughput (T consecutive slices, sliding 5-slice windows, per-slice cache) for one precision.
usage: python3 scripts/volume_bench.py [precision] [T]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
model = build_model(opt, load_prior_files=False, precision=prec).eval().to(dev)
fr = synth_lr_tiles(1, T, 128, 128, seed=77, kind="smooth")[0].to(dev)
win = torch.tensor([[min(max(c + o, 0), T - 1) for o in (-2, -1, 0, 1, 2)] for c in range(T)], dtype=torch.int32, device=dev)
model.forward_volume(fr, win); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): model.forward_volume(fr, win)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"{prec} volume T={T}: {1e3*dt:.1f} ms  {T*1.048576/dt:.1f} MP/s")
