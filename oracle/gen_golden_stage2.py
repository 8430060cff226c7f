#!/usr/bin/env python3
"""Golden vectors for the stage-2 (indexer) TRAINING step, emitted by the UNMODIFIED reference (TEST INFRASTRUCTURE ONLY;
runs only where /root/reference is mounted).  ``model/vqgan_indexer.py::lrGenerator8`` and everything it imports are pure
torch, so they are imported as they are; the ten statements of ``train_vqgan_onestep`` (train_stage2.py:351-366; the module
itself imports cv2/tensorboard) are re-enacted against the reference objects with ``torch.optim.Adam`` and the reference's
``CosineAnnealingLR_Restart``.  Two consecutive steps.

Writes tests/golden/stage2_x8.npz: LR / GT inputs, the encoder's code indices (teacher forcing) and their top-2 distance
margin, logits and loss per step, gradient statistics of all trainable (indexer) tensors, full gradients / updated values
of a few.
    python oracle/gen_golden_stage2.py
"""
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_ROOT = "/root/reference/GPEMSR-CREMI/GPEMSR"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
from gen_golden_train import projection                      # noqa: E402
from gpemsr_amd.arch import param_specs                      # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles, synth_state_dict  # noqa: E402

TRAIN_OPT = dict(lr_G=4e-4, beta1=0.9, beta2=0.99, T_period=[40000, 80000, 120000, 120000, 120000],
                 restarts=[40000, 80000, 240000, 360000], restart_weights=[1, 1, 1, 1], eta_min=1e-7)   # option/train_stage2_x8.yml:78-86
FULL = ("indexer.input_layer.0.weight", "indexer.feat_extract.0.block.1.weight", "indexer.feat_extract.0.block.1.bias",
        "indexer.feat_extract.8.gn.weight", "indexer.feat_extract.8.q.bias", "indexer.feat_extract.8.k.bias",
        "indexer.feat_extract.8.v.bias", "indexer.feat_extract.8.proj_out.bias", "indexer.feat_extract.7.downblock.bias",
        "indexer.feat_extract.3.channel_up.bias", "indexer.embedding.bias", "indexer.output_layer.3.bias")


def main(scale: int = 8):
    torch.set_num_threads(8)
    sys.path.insert(0, REF_ROOT)
    from model.vqgan_indexer import lrGenerator8, lrGenerator16          # the reference, unmodified
    import model.lr_scheduler as lr_scheduler
    with open(os.path.join(REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    gen = (lrGenerator8 if scale == 8 else lrGenerator16)(opt["network"]["argref"])
    gen.load_state_dict({k[len("refmodel."):]: v for k, v in sd.items() if k.startswith("refmodel.")}, strict=True)
    # train_stage2.py:152-179
    for part in (gen.encoder, gen.codebook, gen.decoder):
        for p in part.parameters():
            p.requires_grad = False
    names = [k for k, v in gen.named_parameters() if v.requires_grad]
    params = [v for k, v in gen.named_parameters() if v.requires_grad]
    assert all(n.startswith("indexer.") for n in names)
    optimizer = torch.optim.Adam(params, lr=TRAIN_OPT["lr_G"], betas=(TRAIN_OPT["beta1"], TRAIN_OPT["beta2"]), weight_decay=0)
    scheduler = lr_scheduler.CosineAnnealingLR_Restart(optimizer, TRAIN_OPT["T_period"], eta_min=TRAIN_OPT["eta_min"],
                                                       restarts=TRAIN_OPT["restarts"], weights=TRAIN_OPT["restart_weights"])
    B, lr_size = (2, 32) if scale == 8 else (2, 16)
    LR = synth_lr_tiles(B, 1, lr_size, lr_size, seed=91, kind="smooth")[:, 0]                  # [B,1,32,32] / [B,1,16,16]
    GT = synth_lr_tiles(B, 1, lr_size * scale, lr_size * scale, seed=92, kind="smooth")[:, 0]  # [B,1,256,256]
    arrs = {"LR": LR.numpy(), "GT": GT.numpy()}
    for step in (1, 2):
        gen.train()
        optimizer.zero_grad()
        logits, idx = gen(LR, GT)
        loss = torch.nn.CrossEntropyLoss()(logits, idx)
        loss.backward()
        if step == 1:
            with torch.no_grad():                       # margin of the arg-min (model/codebook.py:19-23)
                z = gen.encoder(GT).permute(0, 2, 3, 1).reshape(-1, 512)
                E = gen.codebook.embedding.weight
                d = (z ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * z @ E.t()
                two = torch.topk(-d, 2, dim=1).values
                arrs["min_distance_margin"] = np.float64((two[:, 0] - two[:, 1]).min().item())
            arrs["target_idx"] = idx.numpy().astype(np.int32)
            arrs["logits_1_every4"] = logits.detach().numpy()[::4]
            stats = np.zeros((len(names), 3), dtype=np.float64)
            for i, (k, p) in enumerate(zip(names, params)):
                g = p.grad.detach().reshape(-1).to(torch.float64)
                stats[i] = (g.norm().item(), g.sum().item(), (g * projection(k, g.numel())).sum().item())
            arrs["grad_names"] = np.array(names)
            arrs["grad_stats"] = stats
            for k in FULL:
                if k in names:
                    arrs["grad__" + k] = dict(zip(names, params))[k].grad.detach().numpy().copy()
        arrs[f"loss_{step}"] = np.float64(loss.item())
        optimizer.step()
        scheduler.step()
        arrs[f"lr_after_{step}"] = np.float64(optimizer.param_groups[0]["lr"])
        for k in FULL:
            if k in names:
                arrs[f"param{step}__" + k] = dict(zip(names, params))[k].detach().numpy().copy()
        print(f"step {step}: loss {loss.item():.6f}")
    path = os.path.join(REPO, "tests", "golden", f"stage2_x{scale}.npz")
    np.savez_compressed(path, **arrs)
    gn = arrs["grad_stats"][:, 0]
    print("wrote", path, os.path.getsize(path), "bytes;", len(names), "trainable tensors; grad norms min %.2e max %.2e; margin %.3e" %
          (gn.min(), gn.max(), arrs["min_distance_margin"]))


if __name__ == "__main__":
    main(16 if "--scale16" in sys.argv else 8)
