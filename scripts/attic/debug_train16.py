import os, sys, numpy as np, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from train_constants import TRAIN_OPT
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.train import Stage3Trainer
d = np.load(os.path.join(ROOT,'tests/golden/train_x8.npz'))
dev = torch.device('cuda',0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
res={}
for prec in ("fp32","bf16x3","bf16"):
    tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision=prec).to(dev), TRAIN_OPT, dev)
    calls=[]
    if tr.eng._frozen16 is not None:
        orig=tr.eng._frozen16.ref_extract
        def wrap(*a,**k):
            calls.append(1); return orig(*a,**k)
        tr.eng._frozen16.ref_extract=wrap
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    res[prec]=(tr.last_sr.clone(), rec.item(), ref.item(), tr.flat_g.clone())
    print(prec, 'frozen16 calls', len(calls), rec.item(), ref.item(), float(d["rec_loss_1"]), float(d["ref_loss_1"]))
for p in ("bf16x3","bf16"):
    a=res[p][0]; b=res["fp32"][0]
    print(p, 'SR rel diff vs fp32', float((a-b).abs().max()/b.abs().max()), 'grad rel diff', float((res[p][3]-res["fp32"][3]).norm()/res["fp32"][3].norm()))
