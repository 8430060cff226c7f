#!/usr/bin/env python3
"""GPU box: host-enqueue vs device time of the phases of one training step (batch 8, LR 32)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench_train import TRAIN_OPT
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
from gpemsr_amd.train import Stage3Trainer
from gpemsr_amd import ops
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision=(sys.argv[1] if len(sys.argv) > 1 else "fp32")).to(dev), TRAIN_OPT, dev)
LR = synth_lr_tiles(8, 5, 32, 32, seed=1, kind="smooth").to(dev)
GT = torch.rand(8, 1, 256, 256).to(dev)
tr.step(LR, GT); torch.cuda.synchronize()
def phase(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:28s} host {1e3*(t1-t0):8.2f} ms   total {1e3*(t2-t0):8.2f} ms"); return r
for it in range(2):
    eng = tr.eng
    def fwd():
        tr.flat_g.zero_(); eng.tape = []
        return eng.forward_train(LR)
    out, ref_img = phase("forward (tape on)", fwd)
    gt = GT.contiguous()
    phase("l1 loss", lambda: ops.l1_loss(out.buf, gt, 1.0, out.grad().buf))
    ref_act = ops.Act(ref_img, 40, 256, 256, 1, 1, 0)
    phase("contextual fwd", lambda: tr._contextual(out, ref_act, 5, 0.001))
    n = len(eng.tape)
    def bwd():
        for fn in reversed(eng.tape): fn()
    phase(f"backward ({n} records)", bwd)
    eng.tape = None
    phase("adam", lambda: ops.adam_step(tr.flat_p, tr.flat_g, tr.flat_m, tr.flat_v, 1e-4, 0.9, 0.99, 1e-8, 0.0, 2))
    phase("refresh_weights (repack)", eng.refresh_weights)
