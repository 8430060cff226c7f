#!/usr/bin/env python3
"""GPU box: A/B of environment switches on the forward (batch 16 of 128^2 windows), variants interleaved in ONE process (guide rule 24).
    python3 scripts/ab_bench.py bf16 GPEMSR_FOLD_GN=0 GPEMSR_FOLD_GN=1 [--rounds 3] [--steps 4]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd.config import build_model, load_options
from gpemsr_amd.synth import synth_lr_tiles
argv = sys.argv[1:]
rounds = int(argv[argv.index("--rounds") + 1]) if "--rounds" in argv else 3
steps = int(argv[argv.index("--steps") + 1]) if "--steps" in argv else 4
for f in ("--rounds", "--steps"):
    if f in argv:
        i = argv.index(f); del argv[i:i + 2]
args = argv
prec, variants = args[0], args[1:]
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
x = synth_lr_tiles(16, 5, 128, 128, seed=1000, kind="uniform").to(dev)
def setenv(v):
    for kv in v.split(","):
        k, val = kv.split("=")
        os.environ[k] = val
models = []
for v in variants:
    setenv(v)
    m = build_model(opt, load_prior_files=False, precision=prec).eval().to(dev)
    m(x); torch.cuda.synchronize()          # engine construction reads the switches
    models.append(m)
outs = []
for r in range(rounds):
    for v, m in zip(variants, models):
        setenv(v)                           # (switches read at call time)
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = m(x)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"round {r} {v:40s} {1e3 * dt:8.2f} ms/step {16 * 1.048576 / dt:8.2f} MP/s", flush=True)
        if r == 0:
            outs.append(o.clone())
for v, o in zip(variants[1:], outs[1:]):
    print(f"max |out[{v}] - out[{variants[0]}]| / max|out| = {float((o - outs[0]).abs().max() / outs[0].abs().max()):.3e}")
# a VQ flip (one codebook cell picked differently) and a numeric error look the same in the line above: tell them apart -- code-index
# agreement of every variant with the first one, and the output delta with the FIRST variant's indices teacher-forced (all windows)
idx, forced = [], []
for v, m in zip(variants, models):
    setenv(v)
    tr = {}
    m(x, trace=tr)
    idx.append(torch.cat(tr["code_idx"]))
for v, m, i in zip(variants, models, idx):
    setenv(v)
    forced.append(m(x, forced_code_idx=idx[0])[0].clone())
    if v != variants[0]:
        print(f"{v}: code indices differ from [{variants[0]}] in {int((i != idx[0]).sum())} of {i.numel()} cells; "
              f"teacher-forced max |out - out[{variants[0]}]| / max|out| = {float((forced[-1] - forced[0]).abs().max() / forced[0].abs().max()):.3e}")
