#!/usr/bin/env python3
"""Golden vectors for the stage-3 contextual loss (forward), emitted by the UNMODIFIED reference:
``model/contextual.py`` (pure torch, imported as is) and ``model/VGG.py`` behind the torchvision shim of
oracle/ref_shims (third-party layer list: parity unpinned for the VGG part, pinned for the CX arithmetic).

Runs only where /root/reference is mounted; writes tests/golden/cx_x8.npz (inputs + expected outputs only).
    python oracle/gen_golden_cx.py
"""
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                                      # noqa: E402  (build_reference, synthetic weights)
from gpemsr_amd.arch import param_specs                      # noqa: E402
from gpemsr_amd.synth import synth_state_dict               # noqa: E402


def main():
    torch.set_num_threads(8)
    scale = 8
    with open(os.path.join(gg.REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    model, _ = gg.build_reference(scale, sd)
    model.load_state_dict(sd, strict=True)
    from model.contextual import ContextualLoss, contextual_loss       # the reference, unmodified
    g = torch.Generator().manual_seed(4242)
    arrs = {}
    with torch.no_grad():
        # (1) the CX arithmetic alone, on random feature maps (x and y of different spatial size, as the code allows)
        fx = torch.randn(2, 32, 8, 8, generator=g)
        fy = torch.randn(2, 32, 8, 4, generator=g) * 0.7 + 0.1
        loss, c = contextual_loss(fx, fy, band_width=0.5, loss_type='cosine')
        arrs.update(f_x=fx.numpy(), f_y=fy.numpy(), f_loss=np.float64(loss.item()), f_c=c.numpy())
        loss2, c2 = contextual_loss(fx, fx.flip(0) * 0.5 + 0.2, band_width=0.1, loss_type='cosine')
        arrs.update(f_loss_bw01=np.float64(loss2.item()), f_c_bw01=c2.numpy())
        # (2) ContextualLoss(model.vgg) on 3-channel images, relu3_4 (the training step's configuration)
        crit = ContextualLoss(model.vgg)
        x3 = torch.rand(2, 3, 32, 32, generator=g)
        y3 = torch.rand(2, 3, 32, 32, generator=g)
        m, s = crit.vgg_mean, crit.vgg_std
        taps = model.vgg((x3 - m) / s)
        loss3, c3 = crit(x3, y3)
        arrs.update(i_x=x3.numpy(), i_y=y3.numpy(), i_loss=np.float64(loss3.item()), i_c=c3.numpy())
        for name in taps._fields:
            arrs["i_x_" + name] = getattr(taps, name).numpy()
        # (3) the loss half of train_EMSR_onestep (train_stage3.py:349-359) on given SR / ref_img / GT
        sr = torch.rand(1, 1, 32, 32, generator=g)
        ref_img = torch.rand(1, 5, 1, 32, 32, generator=g)
        gt = torch.rand(1, 1, 32, 32, generator=g)
        rec = torch.nn.L1Loss()(gt, sr)
        b, _, h, w = sr.size()
        t = ref_img.size(1)
        sr_b = sr[:, None].expand(-1, -1, 3, -1, -1).expand(-1, t, -1, -1, -1).reshape(b * t, 3, h, w)
        ref_b = ref_img.expand(-1, -1, 3, -1, -1).reshape(b * t, 3, h, w)
        ref_loss, u = crit(sr_b, ref_b)
        arrs.update(t_sr=sr.numpy(), t_ref_img=ref_img.numpy(), t_gt=gt.numpy(), t_rec_loss=np.float64(rec.item()),
                    t_ref_loss=np.float64(ref_loss.item()), t_u=u.numpy())
    path = os.path.join(gg.GOLD, "cx_x8.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in arrs.items()})
    print("losses:", arrs["f_loss"], arrs["f_loss_bw01"], arrs["i_loss"], arrs["t_rec_loss"], arrs["t_ref_loss"])


if __name__ == "__main__":
    main()
