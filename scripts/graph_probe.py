#!/usr/bin/env python3
"""GPU box: does the forward / the training step run as ONE hipGraph (torch.cuda.CUDAGraph capture of the launch stream), and what does
replaying it buy over launching the ~300 / ~2,000 kernels from Python?   python3 scripts/graph_probe.py [infer|train|both]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpemsr_amd.config import build_model, load_options  # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
dev = torch.device("cuda", 0)
opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if what in ("infer", "both"):
    for prec, B in (("bf16", 16), ("fp32", 16)):
        x = synth_lr_tiles(B, 5, 128, 128, seed=1000, kind="uniform").to(dev)
        m = build_model(opt, load_prior_files=False, precision=prec).eval().to(dev)
        for _ in range(2):
            out_e = m(x)[0]
        torch.cuda.synchronize()
        ms_e = timeit(lambda: m(x), 4)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g):
                out_g = m(x)[0]
            torch.cuda.synchronize()
            ms_g = timeit(g.replay, 4)
            err = float((out_g - out_e).abs().max())
            print(f"infer {prec}: eager {ms_e:.2f} ms/step, graph replay {ms_g:.2f} ms/step, max |graph - eager| = {err:.3e}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"infer {prec}: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
        del m, g
        torch.cuda.empty_cache()

if what in ("train", "both"):
    import bench_train
    from gpemsr_amd.train import Stage3Trainer
    from gpemsr_amd.dist import average_gradients
    from gpemsr_amd import ops
    model = build_model(opt, load_prior_files=False, precision="fp32").to(dev)
    tr = Stage3Trainer(model, bench_train.TRAIN_OPT, dev, world=1)
    LR = synth_lr_tiles(8, 5, 32, 32, seed=2000, kind="smooth").to(dev)
    GT = torch.rand(8, 1, 256, 256, generator=torch.Generator().manual_seed(3000)).to(dev)
    for _ in range(2):
        o = tr.step(LR, GT)
    ms_e = timeit(lambda: tr.step(LR, GT), 5)
    print(f"train: eager step {ms_e:.2f} ms = {8e3 / ms_e:.1f} samples/s; losses {float(o['rec_loss']):.6f} {float(o['ref_loss']):.6f}", flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            rec, ref = tr.forward_backward(LR, GT)
            average_gradients(tr.flat_g, tr.world)
        torch.cuda.synchronize()

        def gstep():
            g.replay()
            tr.step_count += 1
            b1, b2, eps, wd = tr.adam_hparams()
            ops.adam_step(tr.flat_p, tr.flat_g, tr.flat_m, tr.flat_v, tr.lr, b1, b2, eps, wd, tr.step_count)
            tr.lr = tr.sched.step()
            tr.eng.refresh_weights()
        ms_g = timeit(gstep, 5)
        print(f"train: graph replay + Adam {ms_g:.2f} ms = {8e3 / ms_g:.1f} samples/s; losses {float(rec):.6f} {float(ref):.6f}", flush=True)
        # parity of one more step against the eager step from the same state
        import copy
        p0, m0, v0 = tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone()
        sc, lr0, sched0 = tr.step_count, tr.lr, copy.deepcopy(tr.sched)
        gstep(); torch.cuda.synchronize()
        pg = tr.flat_p.clone(); lg = (float(rec), float(ref))
        tr.flat_p.copy_(p0); tr.flat_m.copy_(m0); tr.flat_v.copy_(v0); tr.step_count, tr.lr, tr.sched = sc, lr0, sched0
        tr.eng.refresh_weights()
        oe = tr.step(LR, GT); torch.cuda.synchronize()
        print(f"train: graph vs eager after one step from the same state: losses {lg} vs {(float(oe['rec_loss']), float(oe['ref_loss']))}, "
              f"max |dp| = {float((tr.flat_p - pg).abs().max()):.3e}", flush=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        print(f"train: capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
