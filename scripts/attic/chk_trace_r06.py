import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
x = synth_lr_tiles(16, 5, 128, 128, seed=1000, kind="uniform").to(dev)
for fold in ("0", "1"):
    os.environ["GPEMSR_FOLD_GN32"] = fold
    m = build_model(opt, load_prior_files=False, precision="fp32").eval().to(dev)
    o1 = m(x)[0].clone()
    o2 = m(x)[0].clone()
    tr = {}
    o3 = m(x, trace=tr)[0].clone()
    idx = torch.cat(tr["code_idx"])
    o4 = m(x, forced_code_idx=idx)[0].clone()
    s = float(o1.abs().max())
    print(f"fold={fold}: free vs free {float((o1-o2).abs().max())/s:.2e}; free vs traced {float((o1-o3).abs().max())/s:.2e}; traced vs forced(own idx) {float((o3-o4).abs().max())/s:.2e}", flush=True)
    if fold == "0":
        base, base_idx = o1, idx
    else:
        print(f"fold 1 vs 0: free {float((o1-base).abs().max())/s:.2e}; idx equal {float((idx==base_idx).float().mean())*100:.4f} %; cells differing {int((idx!=base_idx).sum())}")
