#!/usr/bin/env python3
"""GPU box: forward ms/step for different frame_chunk / tile_chunk settings (batch 16 of 128^2 windows).
    python3 scripts/chunk_sweep.py [precision]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
x = synth_lr_tiles(16, 5, 128, 128, seed=1000, kind="uniform").to(dev)
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model = build_model(opt, load_prior_files=False, precision=prec).eval().to(dev)
for fc, tc in ((80, 16), (40, 16), (20, 16), (10, 16), (80, 8), (40, 8), (80, 16)):
    model._chunks = (fc, tc); model._engine = None
    model(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"frame_chunk {fc:3d} tile_chunk {tc:3d}: {1e3*dt:7.1f} ms/step  {16*1.048576/dt:6.2f} MP/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    torch.cuda.reset_peak_memory_stats()
