#!/usr/bin/env python3
"""Step time of the 8x bf16 forward (16 tiles of 5x128x128) against the frame-chunk size of the per-frame half: smaller chunks keep the
layer-to-layer tensors inside the 256 MB Infinity Cache at the price of more, smaller launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
opt = load_options(os.path.join(root, "option", "output_GPEMSR_x8.yml"))
model = build_model(opt, load_prior_files=False, precision=prec).eval().to(torch.device("cuda", 0))
x = synth_lr_tiles(16, 5, 128, 128, seed=0, kind="uniform").cuda()
for fc in (80, 40, 20, 16, 10, 5):
    model._chunks, model._engine = (fc, 16), None
    for _ in range(2):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    print(f"{prec} frame_chunk {fc:3d}: {(time.perf_counter() - t0) / 5 * 1e3:8.2f} ms/step", flush=True)
