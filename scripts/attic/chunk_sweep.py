#!/usr/bin/env python3
"""Step time of the bench workload (16 windows of 5 x 1 x 128 x 128 -> 1024^2) against the batching granularity of the two halves of the forward
(`frame_chunk` slices / `tile_chunk` windows per launch group).  python3 scripts/chunk_sweep.py [bf16|fp32]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpemsr_amd.config import build_model, load_options  # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
opt = load_options(os.path.join(root, "option", "output_GPEMSR_x8.yml"))
dev = torch.device("cuda", 0)
x = synth_lr_tiles(16, 5, 128, 128, seed=1, kind="smooth").to(dev)
model = build_model(opt, load_prior_files=False, precision=prec).eval().to(dev)
for fc, tc in ((80, 16), (40, 16), (40, 8), (20, 8), (80, 8), (16, 16)):
    model._chunks = (fc, tc)
    model._engine = None
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5 if prec == "bf16" else 2
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
    print(f"{prec}: frame_chunk {fc:3d} tile_chunk {tc:3d}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    torch.cuda.reset_peak_memory_stats()
