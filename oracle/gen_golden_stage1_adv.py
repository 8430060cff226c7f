#!/usr/bin/env python3
"""Golden vectors for the ADVERSARIAL phase of stage-1 (VQGAN) training, emitted by the UNMODIFIED reference (TEST INFRASTRUCTURE ONLY;
runs only where /root/reference is mounted).

The reference's own ``train_stage1.train_vqgan_onestep`` (R:train_stage1.py:291-357) is imported (behind oracle/ref_shims for cv2 /
torchvision) and CALLED for two consecutive steps with ``current_step > gan_start``: step 15 (generator step with the GAN term,
discriminator step) and step 16 (the same plus the R1 penalty: 16 % net_d_reg_every == 0, R:train_stage1.py:339-345,360-372).  Generator =
``model.vqgan.Generator`` with this repository's synthetic x8 prior, discriminator = ``model.discriminator.Discriminator`` loaded
(strict=True) with the seeded initial weights of ``gpemsr_amd.discriminator.Discriminator``; torch.optim.Adam and the reference's
CosineAnnealingLR_Restart for both, as R:train_stage1.py:156-178 builds them.  The option block is option/train_stage1.yml's ``train:``
with gan_start = 0 (so that both steps are adversarial), logger_freq = 1 (the losses are only ever LOGGED: they are parsed from the log
records) and r1_reg_weight = 10 (the yaml's 1e-4 makes the R1 gradient 1e-4 of the discriminator loss's: invisible in a comparison).  A second, fresh run does step 16 alone
(keys with the suffix "16f"): after one Adam step (a sign-like update of every element) two implementations' weights differ by up to
2 * lr wherever a gradient is rounding noise, so only a step from identical weights compares sharply.

Writes tests/golden/stage1_adv.npz: images, per step the code indices the generator chose (teacher forcing), the logged losses, gradient
statistics (L2 norm, sum, seeded projection) of every generator and discriminator tensor as left in ``.grad`` by the call, full
gradients and post-step values of a few tensors, learning rates.
    python oracle/gen_golden_stage1_adv.py
"""
import logging
import os
import re
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
from gpemsr_amd.discriminator import Discriminator as SeededD  # noqa: E402   (initial weights only; nothing is computed with it)

G_FULL = ("encoder.input_layer.0.bias", "decoder.output_layer.weight", "decoder.feat_extract.2.upblock.bias", "encoder.output_layer.3.bias")
D_FULL = ("model.0.weight", "model.0.bias", "model.11.weight", "model.11.bias")


def main():
    torch.set_num_threads(8)
    sys.path.insert(0, os.path.join(HERE, "ref_shims"))
    sys.path.insert(0, REF_ROOT)
    import train_stage1 as ref                                # the reference script, unmodified
    from model.vqgan import Generator
    from model.discriminator import Discriminator
    import model.lr_scheduler as lr_scheduler
    with open(os.path.join(REF_ROOT, "option/train_stage1.yml"), encoding="utf-8") as f:
        opt1 = yaml.safe_load(f)
    with open(os.path.join(REF_ROOT, "option/output_GPEMSR_x8.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    T = dict(opt1["train"])
    T.update(gan_start=0, logger_freq=1, r1_reg_weight=10.0)
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    dargs = opt1["network"]["Discriminator"]
    imgs = synth_lr_tiles(2, 1, 128, 128, seed=271, kind="smooth")[:, 0]
    arrs = {"imgs": imgs.numpy(), "disc_args": np.array([dargs["im_channel"], dargs["num_filters_last"], dargs["n_layers"]])}
    records = []

    class Cap(logging.Handler):
        def emit(self, r):
            records.append(r.getMessage())
    logging.getLogger("base").addHandler(Cap())
    logging.getLogger("base").setLevel(logging.INFO)
    num = r"([-+0-9.eE]+)"

    def run(steps, suffix):
        """fresh generator / discriminator / optimizers / schedulers, then the reference's step function for ``steps``."""
        gen = Generator(opt1["network"]["Generator"])
        gen.load_state_dict({k[len("refmodel."):]: v for k, v in sd.items()
                             if k.startswith(("refmodel.encoder.", "refmodel.codebook.", "refmodel.decoder."))}, strict=True)
        disc = Discriminator(dargs)
        disc.load_state_dict(SeededD(dargs, init_seed=0).state_dict(), strict=True)
        gn, gp = [k for k, _ in gen.named_parameters()], [v for _, v in gen.named_parameters()]
        dn, dp = [k for k, _ in disc.named_parameters()], [v for _, v in disc.named_parameters()]
        opt_g = torch.optim.Adam(gp, lr=T["lr_G"], betas=(T["beta1"], T["beta2"]), weight_decay=0)
        opt_d = torch.optim.Adam(dp, lr=T["lr_D"], betas=(T["beta1"], T["beta2"]), weight_decay=0)
        sch = lambda o: lr_scheduler.CosineAnnealingLR_Restart(o, T["T_period"], eta_min=T["eta_min"], restarts=T["restarts"], weights=T["restart_weights"])  # noqa: E731
        sch_g, sch_d = sch(opt_g), sch(opt_d)
        cap = {}
        hook = gen.register_forward_hook(lambda m, i, o: cap.__setitem__("out", (o[0].detach().clone(), o[1].detach().clone())))
        for step in steps:
            tag = f"{step}{suffix}"
            del records[:]
            ref.train_vqgan_onestep(gen, disc, T, step, imgs.clone(), torch.device("cpu"), opt_g, opt_d, sch_g, sch_d)
            text = "\n".join(records)
            m = re.search(r"reconstruction_loss:" + num + r",\s*gan_loss:" + num + r",codebook_feat_loss:" + num, text)
            arrs[f"rec_loss_{tag}"], arrs[f"g_loss_{tag}"], arrs[f"q_loss_{tag}"] = (np.float64(m.group(i)) for i in (1, 2, 3))
            m = re.search(r"d_loss_real:" + num + r",d_loss_fake:" + num, text)
            arrs[f"d_loss_real_{tag}"], arrs[f"d_loss_fake_{tag}"] = np.float64(m.group(1)), np.float64(m.group(2))
            m = re.search(r"R1_regularization:" + num, text)
            if m:
                arrs[f"r1_{tag}"] = np.float64(m.group(1))
            arrs[f"decoded_{tag}"] = cap["out"][0].numpy()
            arrs[f"code_idx_{tag}"] = cap["out"][1].numpy().astype(np.int32)
            for kind, names, params, full in (("g", gn, gp, G_FULL), ("d", dn, dp, D_FULL)):
                stats = np.zeros((len(names), 3), dtype=np.float64)
                for i, (k, p) in enumerate(zip(names, params)):
                    g = p.grad.detach().reshape(-1).to(torch.float64)
                    stats[i] = (g.norm().item(), g.sum().item(), (g * projection(k, g.numel())).sum().item())
                arrs[f"{kind}_grad_stats_{tag}"] = stats
                byname = dict(zip(names, params))
                for k in full:
                    arrs[f"{kind}_grad_{tag}__{k}"] = byname[k].grad.detach().numpy().copy()
                    arrs[f"{kind}_param_{tag}__{k}"] = byname[k].detach().numpy().copy()
            arrs[f"lr_g_after_{tag}"] = np.float64(opt_g.param_groups[0]["lr"])
            arrs[f"lr_d_after_{tag}"] = np.float64(opt_d.param_groups[0]["lr"])
            print(f"step {tag}: rec {arrs[f'rec_loss_{tag}']:.6f} gan {arrs[f'g_loss_{tag}']:.6f} q {arrs[f'q_loss_{tag}']:.6f} d_real "
                  f"{arrs[f'd_loss_real_{tag}']:.6f} d_fake {arrs[f'd_loss_fake_{tag}']:.6f} r1 {arrs.get(f'r1_{tag}', float('nan'))}")
        hook.remove()
        return gn, dn

    gn, dn = run((15, 16), "")          # two consecutive steps: optimizer / scheduler continuity (step 16 starts from step 15's update)
    run((16,), "f")                     # step 16 (with the R1 penalty) from the FRESH weights: a sharp comparison of the R1 gradient
    arrs["g_names"], arrs["d_names"] = np.array(gn), np.array(dn)
    arrs["train_opt"] = np.array([T["lr_G"], T["lr_D"], T["beta1"], T["beta2"], T["rec_loss_factor"], T["codebook_loss_factor"], T["gan_loss_factor"],
                                  T["r1_reg_weight"], T["net_d_reg_every"], T["generator_update_rate"], opt1["network"]["Generator"]["Codebook"]["beta"]])
    path = os.path.join(REPO, "tests", "golden", "stage1_adv.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
