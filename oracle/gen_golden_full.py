#!/usr/bin/env python3
"""Golden vectors at the BASELINE sizes and for a whole volume, from the UNMODIFIED reference model.

TEST INFRASTRUCTURE ONLY; runs only where /root/reference is mounted (never on the GPU box).
Usage:  python oracle/gen_golden_full.py          (about 3 minutes of CPU)

Adds to tests/golden/ (gen_golden.py's files are untouched):
  full_x8_lr128.npz   one 5-frame window of 128x128 LR tiles through the x8 model (BASELINE.json configs[1] tile size),
  full_x16_lr64.npz   one window of 64x64 LR tiles through the x16 model (configs[3] tile size): the input, the reference's
                      code indices (for teacher forcing), strided sub-samples of `out`, `ref_img` and of the hooked
                      intermediates, a strided sub-sample of the uint8 image the reference's own util.tensor2img
                      produces, and its PSNR against the bilinear base (util.calculate_psnr);
  vol_x8_t7_lr16.npz  a 7-slice volume of 16x16 LR slices quantised to uint8 (what the PNG reader hands over) through the
                      reference's window scheme (R:output_GPEMSR.py:54-128: [0,0,0,1,2], [0,0,1,2,3], sliding windows,
                      [T-4..T-1,T-1], [T-3..T-1,T-1,T-1]), one reference forward per window as that script does: the frames,
                      every window's fp32 output and code indices, and the uint8 images util.tensor2img would write.
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
sys.path.insert(0, REPO)
from gpemsr_amd.arch import param_specs            # noqa: E402
from gpemsr_amd.synth import synth_state_dict, synth_lr_tiles   # noqa: E402
from oracle.gen_golden import GOLD, REF_ROOT, build_reference, store   # noqa: E402


def hooked_forward(model, x):
    cap, hooks = {}, []

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
    hooks.append(model.ThreeDA.register_forward_hook(grab("fused")))
    with torch.no_grad():
        out, ref_img = model(x)
    for h in hooks:
        h.remove()
    cap["code_idx"] = torch.argmax(cap["logits"].reshape(-1, cap["logits"].shape[-1]), dim=1)
    return out, ref_img, cap


def full_case(scale, lr, model, tag):
    import util.util as rutil
    x = synth_lr_tiles(1, 5, lr, lr, seed=4321 + scale, kind="smooth")
    t0 = time.time()
    out, ref_img, cap = hooked_forward(model, x)
    dt = time.time() - t0
    arrs = {"x": x.numpy(), "scale": np.array(scale)}
    for k, t in (("out", out), ("ref_img", ref_img), ("L1_fea", cap["L1_fea"]), ("logits", cap["logits"]), ("mask_cos", cap["mask_cos"]),
                 ("L1_fused", cap["L1_fused"]), ("fused", cap["fused"])):
        store(arrs, k, t)
    arrs["code_idx"] = cap["code_idx"].numpy().astype(np.int32)
    top2 = torch.topk(cap["logits"].reshape(-1, cap["logits"].shape[-1]), 2, dim=1).values
    arrs["logit_margin"] = (top2[:, 0] - top2[:, 1]).numpy()
    img = rutil.tensor2img(out[0:1].clone())
    arrs["out_u8__sub"] = np.ascontiguousarray(img.reshape(-1)[::16])
    arrs["out_u8__stride"] = np.array([16, img.size], dtype=np.int64)
    arrs["psnr_vs_base"] = np.array(rutil.calculate_psnr(
        img, rutil.tensor2img(torch.nn.functional.interpolate(x[0:1, 2], scale_factor=scale, mode="bilinear", align_corners=False))))
    path = os.path.join(GOLD, f"{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB), reference forward {dt:.1f} s")
    return {"case": tag, "ref_forward_s": round(dt, 1), "out_absmax": float(out.abs().max())}


def window_rows(T):
    """R:output_GPEMSR.py:54-128"""
    return ([[0, 0, 0, 1, 2], [0, 0, 1, 2, 3]] + [[i, i + 1, i + 2, i + 3, i + 4] for i in range(T - 4)]
            + [[T - 4, T - 3, T - 2, T - 1, T - 1], [T - 3, T - 2, T - 1, T - 1, T - 1]])


def volume_case(model, tag, T=7, lr=16, scale=8):
    import util.util as rutil
    frames = synth_lr_tiles(1, T, lr, lr, seed=99, kind="smooth")[0]                  # [T,1,lr,lr]
    frames_u8 = (frames[:, 0].numpy() * 255.0).round().astype(np.uint8)
    frames = torch.from_numpy(frames_u8.astype(np.float32) / 255.0).unsqueeze(1)      # what a PNG reader hands over
    rows = window_rows(T)
    outs, imgs, idxs = [], [], []
    for r in rows:
        x = frames[torch.tensor(r)].unsqueeze(0)                                       # [1,5,1,lr,lr]
        out, _, cap = hooked_forward(model, x)
        outs.append(out[0].numpy().copy())
        imgs.append(rutil.tensor2img(out.clone()))
        idxs.append(cap["code_idx"].numpy().astype(np.int32))
    arrs = {"frames_u8": frames_u8, "rows": np.array(rows, dtype=np.int32), "scale": np.array(scale),
            "out": np.stack(outs), "out_u8": np.stack(imgs), "code_idx": np.stack(idxs)}
    path = os.path.join(GOLD, f"{tag}.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")
    return {"case": tag, "windows": len(rows)}


def main():
    torch.set_num_threads(8)
    reports = []
    for scale, lr in ((8, 128), (16, 64)):
        with open(os.path.join(REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
            opt = yaml.safe_load(f)
        kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
        sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
        model, _ = build_reference(scale, sd)
        model.load_state_dict(sd, strict=True)
        reports.append(full_case(scale, lr, model, f"full_x{scale}_lr{lr}"))
        if scale == 8:
            reports.append(volume_case(model, "vol_x8_t7_lr16"))
        for m in [m for m in list(sys.modules) if m.split(".")[0] in ("model", "util", "data")]:
            del sys.modules[m]
    with open(os.path.join(GOLD, "gen_report_full.json"), "w") as f:
        json.dump(reports, f, indent=1)


if __name__ == "__main__":
    main()
