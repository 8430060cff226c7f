"""The reference's training statements, verbatim, on the drop-in module (gpemsr_amd/autograd.py): ``model.train()``,
``SR, ref_img = model(LR)``, ``ContextualLoss(model.vgg)``, ``loss_total.backward()``, ``torch.optim.Adam.step()`` --
train_stage3.py:343-366 -- with torch.autograd driving the HIP tape.  Gradients in ``p.grad`` and the losses of two steps
against the vectors of the UNMODIFIED reference (tests/golden/train_x8.npz; code indices are free-running here, as in the
script: the golden's minimum logit margin is 2e-3, far above fp32 noise)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_reference_training_statements_run_unchanged(golden_dir):
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd import GPEMSR                                     # the one changed import
    from gpemsr_amd.config import load_options
    from gpemsr_amd.contextual import ContextualLoss                  # reference: from model.contextual import ContextualLoss
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    device = torch.device("cuda", 0)
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    net = opt["network"]
    model = GPEMSR(ref_path_G=None, ref_path_Indexer=None, argref=net["argref"], nf=net["nf"], nframes=net["nframes"], groups=net["groups"],
                   front_RBs=net["front_RBs"], back_RBs=net["back_RBs"], w_ref=net["w_ref"], ref_fusion_feat_RBs=net["ref_fusion_feat_RBs"],
                   align_mode=net["align_mode"], fusion_mode=net["fusion_mode"], mode=net["mode"], scale=opt["scale"]).to(device)
    # train_stage3.py:153-163
    optim_params_G = [v for k, v in model.named_parameters() if v.requires_grad]
    names = [k for k, v in model.named_parameters() if v.requires_grad]
    assert names == [str(n) for n in d["grad_names"]]
    optimizer_G = torch.optim.Adam(optim_params_G, lr=TRAIN_OPT["lr_G"], betas=(TRAIN_OPT["beta1"], TRAIN_OPT["beta2"]), weight_decay=0)
    LR, GT = torch.from_numpy(d["LR"]), torch.from_numpy(d["GT"])
    losses = []
    for step in (1, 2):
        # ---- train_EMSR_onestep, train_stage3.py:343-366, statement for statement ----
        model.train()
        GT = GT.to(device)
        LR = LR.to(device)
        optimizer_G.zero_grad()
        SR, ref_img = model(LR)
        L1_loss = torch.nn.L1Loss().to(device)
        rec_loss = L1_loss(GT, SR)
        CLoss = ContextualLoss(model.vgg).to(device)
        b, c, h, w = SR.size()
        b_ref, t, _, _, _ = ref_img.size()
        sr_frame_batch = SR[:, None].expand(-1, -1, 3, -1, -1).expand(-1, t, -1, -1, -1).reshape(b * t, 3, h, w)
        ref_frame_batch = ref_img.expand(-1, -1, 3, -1, -1).reshape(b_ref * t, 3, h, w)
        ref_loss, u = CLoss(sr_frame_batch, ref_frame_batch)
        loss_total = rec_loss * TRAIN_OPT['rec_loss_factor'] + TRAIN_OPT['ref_loss_factor'] * ref_loss
        loss_total.backward()
        if step == 1:
            assert abs(rec_loss.item() - float(d["rec_loss_1"])) <= 1e-5 * float(d["rec_loss_1"])
            assert abs(ref_loss.item() - float(d["ref_loss_1"])) <= 2e-5 * float(d["ref_loss_1"])
            errs = {}
            for i, (k, p) in enumerate(zip(names, optim_params_G)):
                want = d["grad_stats"][i]
                if want[0] == 0.0:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                    continue
                g = p.grad.detach().reshape(-1).double().cpu()
                errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
            print("autograd path, gradient parity worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:3], "median %.1e" % np.median(list(errs.values())))
            trunk = ("recon_trunk.", "upconv", "HRconv", "conv_last")
            for k, e in errs.items():
                assert e <= (1e-3 if k.startswith(trunk) else 6e-2), f"{k}: {e:.2e}"
            assert np.median(list(errs.values())) <= 3e-3
        optimizer_G.step()
        losses.append((rec_loss.item(), ref_loss.item()))
    print("losses", losses, "reference", (float(d["rec_loss_2"]), float(d["ref_loss_2"])))
    assert abs(losses[1][0] - float(d["rec_loss_2"])) <= 2e-3 * float(d["rec_loss_2"])
    assert abs(losses[1][1] - float(d["ref_loss_2"])) <= 2e-3 * float(d["ref_loss_2"])
    # validation afterwards (train_stage3.py: model.eval() + torch.no_grad()) takes the inference path
    model.eval()
    with torch.no_grad():
        out, _ = model(LR)
    assert out.shape == SR.shape and not out.requires_grad


def test_reference_stage2_statements_run_unchanged(golden_dir):
    """train_stage2.py:120-179 + train_vqgan_onestep (:351-366) verbatim on gpemsr_amd.vqgan_indexer.lrGenerator8: torch's
    CrossEntropyLoss and Adam, autograd through the HIP tape; losses of two steps and gradients vs tests/golden/stage2_x8.npz."""
    from gen_golden_stage2 import TRAIN_OPT
    from train_constants import projection
    from gpemsr_amd.arch import param_specs
    from gpemsr_amd.config import load_options
    from gpemsr_amd.synth import synth_state_dict
    from gpemsr_amd.vqgan_indexer import lrGenerator8             # reference: from model.vqgan_indexer import VQGAN_Indexer8 / lrGenerator8
    d = np.load(os.path.join(golden_dir, "stage2_x8.npz"))
    device = torch.device("cuda", 0)
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    lrgenerator = lrGenerator8(opt["network"]["argref"])
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    lrgenerator.load_state_dict({k[len("refmodel."):]: v for k, v in sd.items() if k.startswith("refmodel.")}, strict=True)
    lrgenerator = lrgenerator.to(device)
    # train_stage2.py:152-179
    for part in (lrgenerator.encoder, lrgenerator.codebook, lrgenerator.decoder):
        for k, v in part.named_parameters():
            v.requires_grad = False
    optim_params_G = [v for k, v in lrgenerator.named_parameters() if v.requires_grad]
    names = [k for k, v in lrgenerator.named_parameters() if v.requires_grad]
    assert sorted(names) == sorted(str(n) for n in d["grad_names"])
    optimizer_G = torch.optim.Adam(optim_params_G, lr=TRAIN_OPT["lr_G"], betas=(TRAIN_OPT["beta1"], TRAIN_OPT["beta2"]), weight_decay=0)
    img_LR, img_GT = torch.from_numpy(d["LR"]), torch.from_numpy(d["GT"])
    want_stats = {str(n): d["grad_stats"][i] for i, n in enumerate(d["grad_names"])}
    losses = []
    for step in (1, 2):
        # ---- train_vqgan_onestep, train_stage2.py:351-362 ----
        lrgenerator.train()
        img_GT = img_GT.to(device)
        img_LR = img_LR.to(device)
        optimizer_G.zero_grad()
        logits, gtcodebook_indices = lrgenerator(img_LR, img_GT)
        CELoss = torch.nn.CrossEntropyLoss().to(device)
        loss_total = CELoss(logits, gtcodebook_indices)
        loss_total.backward()
        if step == 1:
            assert torch.equal(gtcodebook_indices.cpu().to(torch.int32), torch.from_numpy(d["target_idx"]))
            errs = {}
            for k, p in zip(names, optim_params_G):
                want = want_stats[k]
                g = p.grad.detach().reshape(-1).double().cpu()
                if k.endswith(".k.bias"):
                    assert g.norm().item() <= 1e-5
                    continue
                errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
            print("stage-2 autograd path, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:3], "median %.1e" % np.median(list(errs.values())))
            assert max(errs.values()) <= 2e-2 and np.median(list(errs.values())) <= 1e-3
        optimizer_G.step()
        losses.append(loss_total.item())
    print("losses", losses, "reference", float(d["loss_1"]), float(d["loss_2"]))
    assert abs(losses[0] - float(d["loss_1"])) <= 1e-5 * float(d["loss_1"])
    assert abs(losses[1] - float(d["loss_2"])) <= 5e-3 * float(d["loss_2"])
    # inference-side methods of the generator object (no autograd)
    lrgenerator.eval()
    with torch.no_grad():
        feats = lrgenerator.ref_extract(img_LR)
        logits2, idx2 = lrgenerator(img_LR, img_GT)
    assert [tuple(f.shape[1:]) for f in feats] == [(512, 16, 16), (256, 32, 32), (128, 64, 64), (64, 128, 128), (1, 256, 256)]
    assert logits2.shape == logits.shape and torch.equal(idx2, gtcodebook_indices)
