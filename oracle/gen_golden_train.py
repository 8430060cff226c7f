#!/usr/bin/env python3
"""Golden vectors for the stage-3 TRAINING step, emitted by the UNMODIFIED reference (TEST INFRASTRUCTURE ONLY; runs only
where /root/reference is mounted).

What is executed is the reference's own code: ``model/GPEMSR.py`` (imported as is behind oracle/ref_shims, like
gen_golden.py), ``model/contextual.py::ContextualLoss``, ``model/lr_scheduler.py::CosineAnnealingLR_Restart`` and
``torch.optim.Adam``, driven by the statements of ``train_EMSR_onestep`` (train_stage3.py:343-366; the function itself
cannot be imported because train_stage3.py imports cv2/tensorboard at module level, so its ten lines are re-enacted here
against the reference objects).  Two consecutive steps are run so that the optimizer state, the scheduler and the
re-packing of updated weights are all pinned.

Writes tests/golden/train_x8.npz (and train_x16.npz with --scale16): inputs (LR, GT), the code indices of the frozen prior (teacher forcing) and its SpyNet flows (same purpose: a constant input that is ill-conditioned in fp32), both loss
values per step, per-parameter gradient statistics of step 1 (L2 norm, sum, and a seeded random projection) for every
trainable tensor, the full gradients of a few small tensors, and parameter values after each step for the same few.
    python oracle/gen_golden_train.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                                      # noqa: E402
from gpemsr_amd.arch import param_specs                      # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles, synth_state_dict  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from train_constants import FULL, TRAIN_OPT, hash_name, projection   # noqa: E402,F401  (the tests own these constants)


def main(scale: int = 8, B: int = 2, lr_size: int = 16):
    torch.set_num_threads(8)
    import yaml
    with open(os.path.join(gg.REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    model, _ = gg.build_reference(scale, sd)
    model.load_state_dict(sd, strict=True)
    from model.contextual import ContextualLoss                          # the reference, unmodified
    import model.lr_scheduler as lr_scheduler                             # the reference, unmodified

    LR = synth_lr_tiles(B, 5, lr_size, lr_size, seed=77 + scale, kind="smooth")
    g = torch.Generator().manual_seed(78)
    GT = torch.rand(B, 1, lr_size * scale, lr_size * scale, generator=g)

    # train_stage3.py:153-178
    optim_params = [v for k, v in model.named_parameters() if v.requires_grad]
    names = [k for k, v in model.named_parameters() if v.requires_grad]
    optimizer = torch.optim.Adam(optim_params, lr=TRAIN_OPT["lr_G"], betas=(TRAIN_OPT["beta1"], TRAIN_OPT["beta2"]), weight_decay=0)
    scheduler = lr_scheduler.CosineAnnealingLR_Restart(optimizer, TRAIN_OPT["T_period"], eta_min=TRAIN_OPT["eta_min"],
                                                       restarts=TRAIN_OPT["restarts"], weights=TRAIN_OPT["restart_weights"])
    cap = {}
    hook = model.refmodel.indexer.register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach().clone()))
    flows = []
    hook2 = model.align_module.spynet.register_forward_hook(lambda m, i, o: flows.append(o.detach().clone()))
    arrs = {"LR": LR.numpy(), "GT": GT.numpy()}
    for step in (1, 2):
        # train_stage3.py:343-366
        model.train()
        optimizer.zero_grad()
        SR, ref_img = model(LR)
        rec_loss = torch.nn.L1Loss()(GT, SR)
        CLoss = ContextualLoss(model.vgg)
        b, c, h, w = SR.size()
        b_ref, t, _, _, _ = ref_img.size()
        sr_frame_batch = SR[:, None].expand(-1, -1, 3, -1, -1).expand(-1, t, -1, -1, -1).reshape(b * t, 3, h, w)
        ref_frame_batch = ref_img.expand(-1, -1, 3, -1, -1).reshape(b_ref * t, 3, h, w)
        ref_loss, u = CLoss(sr_frame_batch, ref_frame_batch)
        loss_total = rec_loss * TRAIN_OPT["rec_loss_factor"] + TRAIN_OPT["ref_loss_factor"] * ref_loss
        loss_total.backward()
        if step == 1:
            logits = cap["logits"]
            arrs["code_idx"] = torch.argmax(logits.reshape(-1, logits.shape[-1]), dim=1).numpy().astype(np.int32)
            top2 = torch.topk(logits.reshape(-1, logits.shape[-1]), 2, dim=1).values
            arrs["min_logit_margin"] = np.float64((top2[:, 0] - top2[:, 1]).min().item())
            arrs["SR"] = SR.detach().numpy()
            arrs["flow"] = torch.stack(flows[0:2 * t:2], dim=1).numpy()       # [B,N,2,4H,4W]; SpyNet is called twice per frame (:99-100)
            stats = np.zeros((len(names), 3), dtype=np.float64)
            for i, (k, p) in enumerate(zip(names, optim_params)):
                if p.grad is None:            # keys the x8 graph never touches (reffea_L4_conv1, ...): Adam skips them
                    stats[i] = (0.0, 0.0, 0.0)
                    continue
                gflat = p.grad.detach().reshape(-1).to(torch.float64)
                stats[i] = (gflat.norm().item(), gflat.sum().item(), (gflat * projection(k, gflat.numel())).sum().item())
            arrs["grad_names"] = np.array(names)
            arrs["grad_stats"] = stats
            for k in FULL:
                arrs["grad__" + k] = dict(zip(names, optim_params))[k].grad.detach().numpy().copy()
        arrs[f"rec_loss_{step}"] = np.float64(rec_loss.item())
        arrs[f"ref_loss_{step}"] = np.float64(ref_loss.item())
        optimizer.step()
        scheduler.step()
        arrs[f"lr_after_{step}"] = np.float64(optimizer.param_groups[0]["lr"])
        for k in FULL:
            arrs[f"param{step}__" + k] = dict(zip(names, optim_params))[k].detach().numpy().copy()
        print(f"step {step}: rec {rec_loss.item():.6f} ref {ref_loss.item():.6f} lr {optimizer.param_groups[0]['lr']:.6e}")
    hook.remove()
    hook2.remove()
    # learning-rate schedules of model/lr_scheduler.py with short periods (restarts, half-weight restart, milestones)
    dummy = [torch.nn.Parameter(torch.zeros(1))]
    o = torch.optim.Adam(dummy, lr=4e-4)
    sc = lr_scheduler.CosineAnnealingLR_Restart(o, [6, 10, 8], eta_min=1e-7, restarts=[6, 16], weights=[1, 0.5])
    seq = []
    for _ in range(30):
        o.step(); sc.step(); seq.append(o.param_groups[0]["lr"])
    arrs["sched_cosine"] = np.array(seq, dtype=np.float64)
    o = torch.optim.Adam(dummy, lr=2e-4)
    sc = lr_scheduler.MultiStepLR_Restart(o, [3, 7, 7, 12], restarts=[9], weights=[0.5], gamma=0.5)
    seq = []
    for _ in range(16):
        o.step(); sc.step(); seq.append(o.param_groups[0]["lr"])
    arrs["sched_multistep"] = np.array(seq, dtype=np.float64)
    path = os.path.join(gg.GOLD, f"train_x{scale}.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes; min logit margin", arrs["min_logit_margin"])
    gn = arrs["grad_stats"][:, 0]
    print("grad norms: min %.3e max %.3e; zero-grad tensors: %s" % (gn.min(), gn.max(), [n for n, v in zip(names, gn) if v == 0.0]))


if __name__ == "__main__":
    if "--scale16" in sys.argv:
        main(16, 1, 16)          # x16: one window of 5 x 16x16 -> 256x256 (every x16-only layer gets a gradient)
    else:
        main()
