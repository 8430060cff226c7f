"""Dev helper (GPU box): stage-by-stage HIP vs CPU-oracle comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
from oracle import gpemsr_oracle as orc

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
opt = load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{scale}.yml"))
model = build_model(opt, load_prior_files=False).eval().cuda()
x = synth_lr_tiles(B, 5, lr, lr, seed=9, kind="smooth")
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
otr, tr = {}, {}
with torch.no_grad():
    want, want_ref = orc.gpemsr_forward(sd, x, scale=scale, trace=otr)
out, ref = model(x.cuda(), forced_code_idx=otr["code_idx"].cuda(), trace=tr)
torch.cuda.synchronize()
def rel(a, b): return float((a.cpu() - b).abs().max() / b.abs().max().clamp_min(1e-12))
for k in ("L1_fea", "logits", "mask_cos", "L1_fused", "aligned", "fused", "recon", "up1", "up_last", "hr"):
    g = tr[k]; g = torch.cat(g) if isinstance(g, list) else g
    print(f"{k:10s} rel err {rel(g.reshape(otr[k].shape), otr[k]):.3e}")
print(f"{'ref_img':10s} rel err {rel(ref, want_ref):.3e}")
print(f"{'out':10s} rel err {rel(out, want):.3e}")
