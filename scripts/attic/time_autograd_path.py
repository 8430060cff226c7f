#!/usr/bin/env python3
"""GPU box: ms/step of the zero-change torch.autograd path (reference statements + torch Adam) next to Stage3Trainer, batch 8."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench_train import TRAIN_OPT
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.contextual import ContextualLoss
from gpemsr_amd.synth import synth_lr_tiles
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
model = build_model(opt, load_prior_files=False).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
optim = torch.optim.Adam(params, lr=4e-4, betas=(0.9, 0.99))
LR = synth_lr_tiles(8, 5, 32, 32, seed=1, kind="smooth").to(dev)
GT = torch.rand(8, 1, 256, 256).to(dev)
def step():
    optim.zero_grad()
    SR, ref_img = model(LR)
    rec = torch.nn.L1Loss()(GT, SR)
    b, c, h, w = SR.size(); t = ref_img.size(1)
    srb = SR[:, None].expand(-1, -1, 3, -1, -1).expand(-1, t, -1, -1, -1).reshape(b * t, 3, h, w)
    rfb = ref_img.expand(-1, -1, 3, -1, -1).reshape(b * t, 3, h, w)
    ref_loss, _ = ContextualLoss(model.vgg)(srb, rfb)
    (rec + 0.001 * ref_loss).backward()
    optim.step()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"autograd path: {1e3*dt:.1f} ms/step = {8/dt:.1f} samples/s (batch 8)")
