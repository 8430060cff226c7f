#!/usr/bin/env python3
"""Generate golden vectors by importing the UNMODIFIED reference model.

TEST INFRASTRUCTURE ONLY; runs only in a container where /root/reference is
mounted (never on the GPU box).  Usage:  python oracle/gen_golden.py

What it does
  1. puts ``oracle/ref_shims`` (restatements of the basicsr/torchvision/cv2
     symbols the reference imports) and the reference root on ``sys.path``;
  2. redirects the reference's hard-coded ``torch.load`` paths
     (model/VGG.py:11, model/GPEMSR.py:65, the YAML's ref_path_G /
     ref_path_Indexer) to synthetic state dicts;
  3. builds ``model.GPEMSR.GPEMSR`` from the reference's own
     ``option/output_GPEMSR_x{8,16}.yml`` and loads the synthetic stage-3 state
     dict with ``strict=True`` (this pins the key/shape inventory of
     ``gpemsr_amd/arch.py``);
  4. runs the reference forward on small seeded tiles, records inputs,
     outputs and hooked intermediates into ``tests/golden/*.npz`` and a
     key/shape manifest into ``tests/golden/state_manifest_x{8,16}.json``;
  5. cross-checks ``oracle/gpemsr_oracle.py`` against every recorded tensor.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF_ROOT = "/root/reference/GPEMSR-CREMI/GPEMSR"
GOLD = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
from gpemsr_amd.arch import param_specs            # noqa: E402
from gpemsr_amd.synth import synth_state_dict, synth_lr_tiles   # noqa: E402
from oracle import gpemsr_oracle as orc           # noqa: E402

MAX_FULL = 200_000      # tensors larger than this are stored strided
SUB_TARGET = 60_000


def store(arrs: dict, name: str, t: torch.Tensor):
    a = t.detach().cpu().numpy().copy()      # copy: the reference's tensor2img clamps its input IN PLACE
    if a.size > MAX_FULL:
        stride = int(np.ceil(a.size / SUB_TARGET))
        arrs[name + "__sub"] = np.ascontiguousarray(a.reshape(-1)[::stride])
        arrs[name + "__stride"] = np.array([stride, a.size], dtype=np.int64)
    else:
        arrs[name] = a


def build_reference(scale: int, sd_synth):
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, os.path.join(HERE, "ref_shims"))
    real_load = torch.load

    def fake_load(path, *a, **k):
        base = os.path.basename(str(path))
        if base.startswith("vgg19"):
            import torchvision.models.vgg as vgg
            return vgg.vgg19().state_dict()
        if base.startswith("spynet"):
            pre = "align_module.spynet."
            return {"params": {k[len(pre):]: v for k, v in sd_synth.items()
                               if k.startswith(pre) and not k.endswith((".mean", ".std"))}}
        if base.startswith("stage1"):
            return {}                                     # loaded strict=False (model/GPEMSR.py:275,283)
        if base.startswith("stage2"):
            pre = "refmodel.indexer."
            return {k[len(pre):]: v for k, v in sd_synth.items() if k.startswith(pre)}
        return real_load(path, *a, **k)

    torch.load = fake_load
    try:
        from model.GPEMSR import GPEMSR                   # the reference, unmodified
        with open(os.path.join(REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
            opt = yaml.safe_load(f)
        net = opt["network"]
        model = GPEMSR(ref_path_G=net["ref_path_G"], ref_path_Indexer=net["ref_path_Indexer"], argref=net["argref"],
                       nf=net["nf"], nframes=net["nframes"], groups=net["groups"], front_RBs=net["front_RBs"],
                       back_RBs=net["back_RBs"], w_ref=net["w_ref"], ref_fusion_feat_RBs=net["ref_fusion_feat_RBs"],
                       align_mode=net["align_mode"], fusion_mode=net["fusion_mode"], mode=net["mode"],
                       scale=opt["scale"])
    finally:
        torch.load = real_load
    model.eval()
    return model, opt


def run_case(scale: int, lr: int, batch: int, kind: str, tag: str, model, sd):
    x = synth_lr_tiles(batch, 5, lr, lr, seed=1234 + scale + lr, kind=kind)
    cap = {}
    hooks = []

    def grab(name, which="out", multi=False):
        def fn(mod, inp, out):
            t = (out if which == "out" else inp[0]).detach().clone()
            if multi:
                cap.setdefault(name, []).append(t)
            else:
                cap[name] = t
        return fn
    hooks.append(model.feature_extraction.register_forward_hook(grab("L1_fea")))
    hooks.append(model.refmodel.indexer.register_forward_hook(grab("logits")))
    hooks.append(model.refmaskconv1.register_forward_hook(grab("mask_cos", "in")))
    hooks.append(model.reduce_dim_conv.register_forward_hook(grab("L1_fused")))
    hooks.append(model.align_module.register_forward_hook(grab("aligned", multi=True)))
    hooks.append(model.ThreeDA.register_forward_hook(grab("fused")))
    hooks.append(model.align_module.spynet.register_forward_hook(grab("flow", multi=True)))
    t0 = time.time()
    with torch.no_grad():
        out, ref_img = model(x)
    dt = time.time() - t0
    for h in hooks:
        h.remove()
    cap["aligned"] = torch.stack(cap["aligned"], dim=1)
    cap["flow"] = torch.stack(cap["flow"][0::2], dim=1)       # SpyNet is called twice per frame (:99-100)
    cap["code_idx"] = torch.argmax(cap["logits"].reshape(-1, cap["logits"].shape[-1]), dim=1)

    # the oracle on the same input
    tr = {}
    with torch.no_grad():
        o_out, o_ref = orc.gpemsr_forward(sd, x, scale=scale, trace=tr)
    top2 = torch.topk(cap["logits"].reshape(-1, cap["logits"].shape[-1]), 2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    rep = {"case": tag, "ref_forward_s": round(dt, 2),
           "idx_agree": float((tr["code_idx"] == cap["code_idx"]).float().mean()),
           "min_logit_margin": float(margin.min())}
    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    rep["out"] = rel(o_out, out)
    rep["ref_img"] = rel(o_ref, ref_img)
    for k in ("L1_fea", "logits", "mask_cos", "L1_fused", "aligned", "fused"):
        rep[k] = rel(tr[k], cap[k])
    print(json.dumps(rep))

    arrs = {"x": x.numpy(), "scale": np.array(scale), "margin_min": np.array(float(margin.min()))}
    store(arrs, "out", out)
    store(arrs, "ref_img", ref_img)
    arrs["code_idx"] = cap["code_idx"].numpy().astype(np.int32)
    arrs["logit_margin"] = margin.numpy()
    for k in ("L1_fea", "logits", "mask_cos", "L1_fused", "aligned", "fused", "flow"):
        store(arrs, k, cap[k])
    # image-space golden (util/util.py:139-163,253-260) through the reference's own helpers
    import util.util as rutil
    img = rutil.tensor2img(out[0:1].clone())
    arrs["out_u8"] = img
    arrs["psnr_vs_base"] = np.array(rutil.calculate_psnr(
        img, rutil.tensor2img(torch.nn.functional.interpolate(x[0:1, 2], scale_factor=scale, mode="bilinear",
                                                                align_corners=False))))
    path = os.path.join(GOLD, f"{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")
    return rep


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    reports = []
    for scale in (8, 16):
        with open(os.path.join(REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
            opt = yaml.safe_load(f)
        kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
        specs = param_specs(scale=opt["scale"], **kw)
        sd = synth_state_dict(specs, seed=0)
        model, _ = build_reference(scale, sd)
        ref_sd = model.state_dict()
        missing = sorted(set(ref_sd) - set(sd)); extra = sorted(set(sd) - set(ref_sd))
        assert not missing and not extra, (missing[:5], extra[:5])
        for k, v in ref_sd.items():
            assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
        model.load_state_dict(sd, strict=True)              # output_GPEMSR.py:52
        trainable = {k for k, p in model.named_parameters() if p.requires_grad}
        mine = {k for k, s in specs.items() if s.trainable}
        assert trainable == mine, (sorted(trainable ^ mine)[:10])
        manifest = {"scale": scale, "n_tensors": len(ref_sd),
                    "n_params": int(sum(p.numel() for p in model.parameters())),
                    "n_trainable": int(sum(p.numel() for p in model.parameters() if p.requires_grad)),
                    "keys": [[k, list(v.shape)] for k, v in ref_sd.items()]}
        with open(os.path.join(GOLD, f"state_manifest_x{scale}.json"), "w") as f:
            json.dump(manifest, f)
        print(f"x{scale}: {manifest['n_tensors']} tensors, {manifest['n_params']} params, "
              f"{manifest['n_trainable']} trainable")
        cases = [(16, 1, "uniform"), (32, 1, "smooth")] if scale == 8 else [(16, 1, "smooth")]
        for lr, b, kind in cases:
            reports.append(run_case(scale, lr, b, kind, f"x{scale}_lr{lr}_b{b}_{kind}", model, sd))
        for m in [m for m in list(sys.modules) if m.split(".")[0] in ("model", "util", "data")]:
            del sys.modules[m]
    with open(os.path.join(GOLD, "gen_report.json"), "w") as f:
        json.dump(reports, f, indent=1)


if __name__ == "__main__":
    main()
