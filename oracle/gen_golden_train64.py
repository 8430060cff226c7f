#!/usr/bin/env python3
"""fp64 run of the reference's stage-3 training step 1 (TEST INFRASTRUCTURE ONLY; runs only where /root/reference is mounted).

Purpose (tests/test_train_gpu.py::test_gradient_distance_to_fp64_is_no_worse_than_the_reference): the fp32 gradients of this
network are ill-conditioned upstream of the reconstruction trunk (kinks of LeakyReLU / max-pool / bilinear taps),
so instead of a wide tolerance against the reference's fp32 numbers the test measures BOTH fp32 implementations -- the
reference's (tests/golden/train_x8.npz) and the HIP one -- against the SAME network evaluated in float64, and requires the HIP
distance to be no worse than the reference's own.

The unmodified reference model is cast to float64 and driven exactly like oracle/gen_golden_train.py drives it, with the frozen
prior's code indices and the SpyNet flows of the fp32 run forced through forward hooks (the GPU tests teacher-force the same
two constants), so all three evaluations differentiate the same function.  Writes tests/golden/train_x8_fp64.npz:
grad_names, grad_stats64 (L2 norm, sum, seeded projection per trainable tensor), rec_loss_1, ref_loss_1, SR64.
    python oracle/gen_golden_train64.py         (about 5 minutes of CPU)
"""
import os
import sys
import time

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                                      # noqa: E402
from gpemsr_amd.arch import param_specs                      # noqa: E402
from gpemsr_amd.synth import synth_state_dict                # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from train_constants import TRAIN_OPT, projection            # noqa: E402


def main(scale: int = 8):
    torch.set_num_threads(8)
    d = np.load(os.path.join(gg.GOLD, f"train_x{scale}.npz"))
    with open(os.path.join(gg.REF_ROOT, f"option/output_GPEMSR_x{scale}.yml"), encoding="utf-8") as f:
        opt = yaml.safe_load(f)
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=opt["scale"], **kw), seed=0)
    model, _ = gg.build_reference(scale, sd)
    model.load_state_dict(sd, strict=True)
    model = model.double()
    from model.contextual import ContextualLoss              # the reference, unmodified
    LR, GT = torch.from_numpy(d["LR"]).double(), torch.from_numpy(d["GT"]).double()
    B, t = LR.shape[0], LR.shape[1]

    # force the fp32 run's code indices (a one-hot logit tensor has the forced argmax) and SpyNet flows
    code = torch.from_numpy(d["code_idx"]).long()
    flows = torch.from_numpy(d["flow"]).double()              # [B,N,2,4H,4W]
    calls = {"spy": 0}

    def indexer_hook(mod, inp, out):
        forced = torch.zeros_like(out).reshape(-1, out.shape[-1])
        forced[torch.arange(forced.shape[0]), code] = 1.0
        return forced.reshape(out.shape)

    def spynet_hook(mod, inp, out):
        # called twice per frame with the SAME arguments (model/GPEMSR.py:99-100); gen_golden_train.py stored the first of each pair
        i = calls["spy"]; calls["spy"] += 1
        return flows[:, i // 2].reshape(out.shape)
    model.refmodel.indexer.register_forward_hook(indexer_hook)
    model.align_module.spynet.register_forward_hook(spynet_hook)
    names = [k for k, v in model.named_parameters() if v.requires_grad]
    params = [v for k, v in model.named_parameters() if v.requires_grad]
    model.train()
    t0 = time.time()
    SR, ref_img = model(LR)
    rec_loss = torch.nn.L1Loss()(GT, SR)
    CLoss = ContextualLoss(model.vgg)
    b, c, h, w = SR.size()
    b_ref, tt, _, _, _ = ref_img.size()
    sr_frame_batch = SR[:, None].expand(-1, -1, 3, -1, -1).expand(-1, tt, -1, -1, -1).reshape(b * tt, 3, h, w)
    ref_frame_batch = ref_img.expand(-1, -1, 3, -1, -1).reshape(b_ref * tt, 3, h, w)
    ref_loss, _ = CLoss(sr_frame_batch, ref_frame_batch)
    (rec_loss * TRAIN_OPT["rec_loss_factor"] + TRAIN_OPT["ref_loss_factor"] * ref_loss).backward()
    stats = np.zeros((len(names), 3), dtype=np.float64)
    for i, (k, p) in enumerate(zip(names, params)):
        if p.grad is None:
            continue
        g = p.grad.detach().reshape(-1)
        stats[i] = (g.norm().item(), g.sum().item(), (g * projection(k, g.numel())).sum().item())
    assert [str(n) for n in d["grad_names"]] == names
    out = {"grad_names": np.array(names), "grad_stats64": stats, "rec_loss_1": np.float64(rec_loss.item()), "ref_loss_1": np.float64(ref_loss.item()),
           "SR64": SR.detach().numpy()}
    path = os.path.join(gg.GOLD, f"train_x{scale}_fp64.npz")
    np.savez_compressed(path, **out)
    ref32 = d["grad_stats"]
    nz = stats[:, 0] > 0
    e = np.maximum(np.abs(ref32[nz, 0] - stats[nz, 0]), np.abs(ref32[nz, 2] - stats[nz, 2])) / stats[nz, 0]
    print(f"wrote {path}; {time.time() - t0:.0f} s; rec {rec_loss.item():.8f} (fp32 {float(d['rec_loss_1']):.8f}) ref {ref_loss.item():.6f} "
          f"(fp32 {float(d['ref_loss_1']):.6f}); reference fp32 vs fp64 gradient distance: median {np.median(e):.2e}, 90th pct {np.percentile(e, 90):.2e}, max {e.max():.2e}")


if __name__ == "__main__":
    main(16 if "--scale16" in sys.argv else 8)
