"""Dev helper (GPU box): stage-by-stage HIP vs CPU-oracle comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
from oracle import gpemsr_oracle as orc

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
precision = sys.argv[4] if len(sys.argv) > 4 else "fp32"
opt = load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{scale}.yml"))
model = build_model(opt, load_prior_files=False, precision=precision).eval().cuda()
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

import numpy as np
from gpemsr_amd.imgutil import tensor2img, calculate_psnr
u8, w8 = tensor2img(out[0]), tensor2img(want[0])
base = tensor2img(torch.nn.functional.interpolate(x[0:1, 2], scale_factor=scale, mode="bilinear", align_corners=False))
print("u8 max diff", int(np.abs(u8.astype(int) - w8.astype(int)).max()), "dPSNR vs base %.5f" % (calculate_psnr(u8, base) - calculate_psnr(w8, base)),
      "PSNR(ours, oracle) %.2f" % calculate_psnr(u8, w8))
out_free, _ = model(x.cuda())
tr2 = {}; model(x.cuda(), trace=tr2)
idx = torch.cat(tr2["code_idx"]).cpu()
print("free-running index agreement", float((idx == otr["code_idx"].to(idx.dtype)).float().mean()), "out rel err free-running %.3e" % rel(out_free, want))
