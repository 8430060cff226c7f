import os, sys, cProfile, pstats, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from train_constants import TRAIN_OPT
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.train import Stage3Trainer
from gpemsr_amd.synth import synth_lr_tiles
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
dev = torch.device('cuda', 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision=prec).to(dev), TRAIN_OPT, dev)
LR = synth_lr_tiles(8, 5, 32, 32, seed=1, kind="uniform").to(dev)
GT = torch.rand(8, 1, 256, 256, device=dev)
for _ in range(2): tr.step(LR, GT)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): tr.step(LR, GT)
torch.cuda.synchronize()
pr.disable()
print(prec, "ms/step", 1e3 * (time.perf_counter() - t0) / 3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
