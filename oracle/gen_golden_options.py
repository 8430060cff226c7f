"""Fixture generator (run only where /root/reference is mounted): the CONTRACT of the reference's stage-3 training option files --
their key sets and `train:` blocks -- plus the learning-rate trace of the reference's OWN scheduler class
(R:model/lr_scheduler.py:36-68, imported unmodified) driven exactly as R:train_stage3.py:153-181 drives it with those blocks.
Emits tests/golden/train_options.json.  Data only: keys, scalar values and sampled learning rates -- no reference text.

    python oracle/gen_golden_options.py [out.json]
"""
import json
import os
import sys

import torch
import yaml

REF = "/root/reference/GPEMSR-CREMI/GPEMSR"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def flat_keys(d, prefix=""):
    out = []
    for k, v in d.items():
        p = f"{prefix}.{k}" if prefix else str(k)
        out.append(p)
        if isinstance(v, dict):
            out += flat_keys(v, p)
    return out


def main():
    sys.path.insert(0, REF)
    from model.lr_scheduler import CosineAnnealingLR_Restart          # the reference's class, unmodified
    doc = {}
    for s in (8, 16):
        opt = yaml.safe_load(open(os.path.join(REF, "option", f"train_stage3_x{s}.yml")))
        tr = opt["train"]
        p = torch.nn.Parameter(torch.zeros(1))
        optim = torch.optim.Adam([p], lr=tr["lr_G"], betas=(tr["beta1"], tr["beta2"]), weight_decay=tr.get("weight_decay_G") or 0)
        sched = CosineAnnealingLR_Restart(optim, tr["T_period"], eta_min=tr["eta_min"], restarts=tr["restarts"], weights=tr["restart_weights"])
        niter = int(tr["niter"])
        # sample points: the first steps, every restart's neighbourhood, every 5000th step, the end
        want = set(range(1, 6)) | {niter}
        for r in tr["restarts"]:
            want |= set(range(r - 2, r + 4))
        want |= set(range(5000, niter + 1, 5000))
        trace = {}
        for step in range(1, niter + 1):
            optim.step()
            sched.step()
            if step in want:
                trace[str(step)] = optim.param_groups[0]["lr"]
        doc[f"x{s}"] = {"keys": sorted(flat_keys(opt)), "train": tr, "val": opt.get("val"), "scale": opt["scale"],
                        "batch_size": opt["datasets"]["train"]["batch_size"], "GT_size": opt["datasets"]["train"]["GT_size"],
                        "LQ_size": opt["datasets"]["train"]["LQ_size"], "lr_trace": trace}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "train_options.json")
    json.dump(doc, open(out, "w"), indent=0, sort_keys=True)
    print("wrote", out, {k: len(v["lr_trace"]) for k, v in doc.items()})


if __name__ == "__main__":
    main()
