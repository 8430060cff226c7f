#!/usr/bin/env python3
"""Golden vectors for the stage-1 (VQGAN) GENERATOR-PHASE training step, emitted by the UNMODIFIED reference (TEST INFRASTRUCTURE
ONLY; runs only where /root/reference is mounted).

Stage 1 trains ``model/vqgan.py::Generator`` (Encoder -> Codebook -> Decoder).  For ``current_step <= gan_start`` (the first 40,000
steps with option/train_stage1.yml) ``train_vqgan_onestep`` (train_stage1.py:313-326) is generator-only:
    decoded, _, q_loss = generator(imgs);  vq_loss = rec_loss_factor * L1(imgs, decoded) + codebook_loss_factor * q_loss
    vq_loss.backward(); optimizer_G.step(); scheduler_G.step()
Those statements are re-enacted here against the reference objects (the module itself imports cv2 / tensorboard) with
torch.optim.Adam and the reference's CosineAnnealingLR_Restart, two consecutive steps.  The adversarial phase (PatchGAN
discriminator, R1 penalty) is NOT covered: the HIP side does not implement it.

Writes tests/golden/stage1_gen.npz: the image batch, the code indices and their top-2 distance margin (teacher forcing: arg-min over
1024 codes is discontinuous), rec / codebook / total loss per step, the decoded image of step 1, gradient statistics (L2 norm, sum,
seeded projection) of every generator tensor, the full gradients and the values after each step of a few.
    python oracle/gen_golden_stage1.py
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
sys.path.insert(0, os.path.join(REPO, "tests"))
from train_constants import projection                       # noqa: E402
from gpemsr_amd.arch import param_specs                      # noqa: E402
from gpemsr_amd.synth import synth_lr_tiles, synth_state_dict  # noqa: E402

FULL = ("encoder.input_layer.0.bias", "encoder.feat_extract.0.block.1.weight", "encoder.output_layer.3.bias",
        "decoder.input_layer.0.bias", "decoder.feat_extract.0.gn.weight", "decoder.feat_extract.2.upblock.bias", "decoder.output_layer.bias")


def main():
    torch.set_num_threads(8)
    sys.path.insert(0, REF_ROOT)
    from model.vqgan import Generator                         # the reference, unmodified
    import model.lr_scheduler as lr_scheduler
    with open(os.path.join(REF_ROOT, "option/train_stage1.yml"), encoding="utf-8") as f:
        opt1 = yaml.safe_load(f)
    with open(os.path.join(REF_ROOT, "option/output_GPEMSR_x8.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    T = opt1["train"]
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    gen = Generator(opt1["network"]["Generator"])
    mine = {k[len("refmodel."):]: v for k, v in sd.items() if k.startswith(("refmodel.encoder.", "refmodel.codebook.", "refmodel.decoder."))}
    gen.load_state_dict(mine, strict=True)                   # the x8 prior of the stage-3 model IS this generator (same argref blocks)
    names = [k for k, v in gen.named_parameters()]
    params = [v for k, v in gen.named_parameters()]
    optimizer = torch.optim.Adam(params, lr=T["lr_G"], betas=(T["beta1"], T["beta2"]), weight_decay=0)      # train_stage1.py:164-170
    scheduler = lr_scheduler.CosineAnnealingLR_Restart(optimizer, T["T_period"], eta_min=T["eta_min"], restarts=T["restarts"],
                                                       weights=T["restart_weights"])
    imgs = synth_lr_tiles(2, 1, 128, 128, seed=171, kind="smooth")[:, 0]                # [2,1,128,128] -> latent 8x8 (64 tokens per image)
    arrs = {"imgs": imgs.numpy()}
    cap = {}
    hook = gen.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("z", o.detach().clone()))
    for step in (1, 2):
        gen.train()
        optimizer.zero_grad()
        decoded, idx, q_loss = gen(imgs)
        rec_loss = torch.nn.L1Loss()(imgs, decoded)
        vq_loss = T["rec_loss_factor"] * rec_loss + T["codebook_loss_factor"] * q_loss
        vq_loss.backward()
        if step == 1:
            z = cap["z"].permute(0, 2, 3, 1).reshape(-1, cap["z"].shape[1])
            E = gen.codebook.embedding.weight.detach()
            d = (z ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * z @ E.t()
            top2 = torch.topk(-d, 2, dim=1).values
            arrs["code_idx"] = idx.numpy().astype(np.int32)
            arrs["code_margin"] = (top2[:, 0] - top2[:, 1]).numpy()
            arrs["decoded"] = decoded.detach().numpy()
            arrs["z"] = cap["z"].numpy()
            stats = np.zeros((len(names), 3), dtype=np.float64)
            for i, (k, p) in enumerate(zip(names, params)):
                g = p.grad.detach().reshape(-1).to(torch.float64)
                stats[i] = (g.norm().item(), g.sum().item(), (g * projection(k, g.numel())).sum().item())
            arrs["grad_names"] = np.array(names)
            arrs["grad_stats"] = stats
            for k in FULL:
                arrs["grad__" + k] = dict(zip(names, params))[k].grad.detach().numpy().copy()
        arrs[f"rec_loss_{step}"] = np.float64(rec_loss.item())
        arrs[f"q_loss_{step}"] = np.float64(q_loss.item())
        arrs[f"vq_loss_{step}"] = np.float64(vq_loss.item())
        arrs[f"code_idx_{step}"] = idx.numpy().astype(np.int32)
        optimizer.step()
        scheduler.step()
        arrs[f"lr_after_{step}"] = np.float64(optimizer.param_groups[0]["lr"])
        for k in FULL:
            arrs[f"param{step}__" + k] = dict(zip(names, params))[k].detach().numpy().copy()
        print(f"step {step}: rec {rec_loss.item():.6f} q {q_loss.item():.6f} total {vq_loss.item():.6f} lr {optimizer.param_groups[0]['lr']:.6e}; "
              f"latent {tuple(cap['z'].shape)}, distinct codes {len(set(idx.tolist()))}")
    hook.remove()
    arrs["train_opt"] = np.array([T["lr_G"], T["beta1"], T["beta2"], T["rec_loss_factor"], T["codebook_loss_factor"], opt1["network"]["Generator"]["Codebook"]["beta"]])
    path = os.path.join(REPO, "tests", "golden", "stage1_gen.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes; min code margin", float(arrs["code_margin"].min()))


if __name__ == "__main__":
    main()
