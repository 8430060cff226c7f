#!/usr/bin/env python3
"""GPU box: one training step against tests/golden/train_x8.npz with a per-tensor gradient error listing."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gen_golden_train import TRAIN_OPT, projection
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.train import Stage3Trainer

d = np.load(os.path.join(ROOT, "tests/golden/train_x8.npz"))
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
model = build_model(opt, load_prior_files=False).to(dev)
tr = Stage3Trainer(model, TRAIN_OPT, dev)
LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
idx = torch.from_numpy(d["code_idx"]).to(dev)
flow = torch.from_numpy(d['flow']).to(dev)
rec, ref = tr.forward_backward(LR, GT, idx, flow)
torch.cuda.synchronize()
print("rec", rec.item(), d["rec_loss_1"], "ref", ref.item(), d["ref_loss_1"])
sr = tr.last_sr.cpu().numpy()
print("SR err", np.abs(sr - d["SR"]).max() / np.abs(d["SR"]).max())
names = [str(n) for n in d["grad_names"]]
stats = d["grad_stats"]
bad = 0
for i, k in enumerate(names):
    base, leaf = k.rsplit(".", 1)
    g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
    got = (g.norm().item(), g.sum().item(), (g * projection(k, g.numel())).sum().item())
    want = stats[i]
    scale = max(want[0], 1e-12)
    err = max(abs(got[0] - want[0]), abs(got[2] - want[2])) / scale
    flag = "" if err < 1e-3 else "   <<<<"
    bad += err >= 1e-3
    if flag or "-v" in sys.argv:
        print(f"{k:60s} norm {got[0]:.5e} / {want[0]:.5e}   proj {got[2]:+.5e} / {want[2]:+.5e}  err {err:.2e}{flag}")
print("bad tensors:", bad, "of", len(names))

# ---- against the CPU oracle's autograd on the same inputs (full tensors) ----
import yaml
from oracle import gpemsr_oracle as orc
from gpemsr_amd.arch import param_specs
from gpemsr_amd.synth import synth_state_dict
torch.set_num_threads(16)
o = yaml.safe_load(open(os.path.join(ROOT, "option/output_GPEMSR_x8.yml")))
kw = {k: v for k, v in o["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
sd = synth_state_dict(param_specs(scale=8, **kw), seed=0)
for k in names:
    sd[k] = sd[k].clone().requires_grad_(True)
out, refi = orc.gpemsr_forward(sd, LR.cpu(), scale=8, forced_idx=idx.cpu().long(), forced_flow=flow.cpu())
rec_o, ref_o, _ = orc.stage3_losses(sd, out, refi.detach(), GT.cpu())
(rec_o * TRAIN_OPT["rec_loss_factor"] + TRAIN_OPT["ref_loss_factor"] * ref_o).backward()
errs = []
for k in names:
    if sd[k].grad is None:
        continue
    base, leaf = k.rsplit(".", 1)
    g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
    w = sd[k].grad.reshape(-1).double()
    errs.append(((g - w).norm().item() / max(w.norm().item(), 1e-30), k))
errs.sort(reverse=True)
print("HIP vs oracle autograd, ||dg|| / ||g||, worst:", [(f"{e:.1e}", k) for e, k in errs[:10]])
print("median", np.median([e for e, _ in errs]), "count > 1e-3:", sum(e > 1e-3 for e, _ in errs))

for k in ["ThreeDA.spatial_attn3.weight", "ThreeDA.spatial_attn2.bias", "ThreeDA.spatial_attn_l2.bias", "ThreeDA.spatial_attn5.bias", "ThreeDA.spatial_attn_add2.bias", "ThreeDA.conv2D_fusion_3.bias"]:
    base, leaf = k.rsplit(".", 1)
    g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
    w = sd[k].grad.reshape(-1).double()
    dd = (g - w).abs()
    print(k, "rel", (dd.norm() / w.norm()).item(), "max|d|", dd.max().item(), "max|w|", w.abs().max().item(), "n>10%max", int((dd > 0.1 * dd.max()).sum()), "of", dd.numel())
