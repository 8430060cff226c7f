#!/usr/bin/env python3
"""Stage-3 training entry point -- the boundary of BASELINE configs[4], drop-in for the reference's ``train_stage3.py``:

    python train_stage3.py -opt option/train_stage3_x8.yml            # one GPU
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_stage3.py -opt option/train_stage3_x8.yml --launcher pytorch

Same ``-opt`` / ``--launcher`` arguments and option keys as R:train_stage3.py:29-45; the loop (R:train_stage3.py:186-336) is: one
optimisation step (``gpemsr_amd.train.Stage3Trainer.step`` = R:train_stage3.py:343-374 on the HIP kernels, gradients averaged by one
RCCL all-reduce), validation every ``val.val_freq`` steps (``gpemsr_amd.validate.validate_psnr`` = :199-317), checkpoint every
``save_checkpoint_freq`` steps (``{step}_G.pth`` = ``model.state_dict()`` + ``{step}.state`` with the optimizer / scheduler state in
torch's layout, :183-184, :319-334).  Additive: ``--max-steps`` for smoke runs, ``precision`` / ``synthetic_data_if_missing`` keys.
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import sys
import time

import torch

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)

from gpemsr_amd import dist as gdist                                   # noqa: E402
from gpemsr_amd.config import build_model, dict_to_nonedict, load_options   # noqa: E402


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    ap.add_argument("--launcher", choices=["none", "pytorch"], default="none", help="job launcher")
    ap.add_argument("--local_rank", type=int, default=0)
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many steps (0: train.niter)")
    ap.add_argument("--out", type=str, default="", help="experiment directory (default ./experiments/<name>)")
    args = ap.parse_args(argv)
    opt = dict_to_nonedict(load_options(args.opt))
    rank, world, local = gdist.init_from_env()
    assert torch.cuda.is_available(), "train_stage3.py needs MI355X GPUs (no CPU path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    scale = int(opt["scale"])
    seed = opt["train"]["manual_seed"]
    if seed is not None:
        import random
        random.seed(seed + rank); torch.manual_seed(seed + rank)

    from gpemsr_amd.data import make_dataset
    from gpemsr_amd.train import Stage3Trainer
    from gpemsr_amd.validate import validate_psnr
    synth = bool(opt["synthetic_data_if_missing"])
    tr_opt = dict(opt["datasets"]["train"]); tr_opt["phase"] = "train"
    train_set = make_dataset(tr_opt, scale, True, synth, seed=1000 * rank)
    val_set = make_dataset(dict(opt["datasets"]["val"]), scale, False, synth, seed=999) if opt["datasets"]["val"] else None
    sampler = torch.utils.data.distributed.DistributedSampler(train_set, world, rank, shuffle=bool(tr_opt.get("use_shuffle"))) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_set, batch_size=int(tr_opt["batch_size"]), shuffle=(sampler is None and bool(tr_opt.get("use_shuffle"))),
                                         sampler=sampler, num_workers=min(int(tr_opt.get("n_workers") or 0), 8), drop_last=True, pin_memory=True)

    have_prior = osp.exists(str(opt["network"]["ref_path_G"])) and osp.exists(str(opt["network"]["ref_path_Indexer"]))
    model = build_model(opt, load_prior_files=have_prior).to(dev)
    if opt["pretrain"] and opt["pretrain"]["EMSR"]:
        model.load_state_dict(torch.load(opt["pretrain"]["EMSR"], map_location="cpu"), strict=bool(opt["pretrain"]["strict_load"]))
    trainer = Stage3Trainer(model, dict(opt["train"]), dev, world=world)
    step = int(opt["train"]["current_step"] or 0)
    if opt["pretrain"] and opt["pretrain"]["training_state"]:
        st = torch.load(opt["pretrain"]["training_state"], map_location="cpu")
        if step != st["iter"]:
            raise ValueError("train.current_step does not match the training state's iteration")
        trainer.load_torch_optimizer_state_dict(st["optimizers"]["0"], st["schedulers"]["0"])
    out_dir = args.out or osp.join(ROOT, "experiments", str(opt["name"]))
    if rank == 0:
        os.makedirs(osp.join(out_dir, "models"), exist_ok=True)
        os.makedirs(osp.join(out_dir, "training_state"), exist_ok=True)
    niter = int(args.max_steps or opt["train"]["niter"])
    val_freq = int(opt["val"]["val_freq"]) if opt["val"] and opt["val"]["val_freq"] else 0
    save_freq = int(opt["save_checkpoint_freq"] or 0)
    log_freq = int(opt["train"]["logger_freq"] or 100)
    epoch, t0 = int(opt["train"]["start_epoch"] or 0), time.time()
    while step < niter:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch in loader:
            step += 1
            if step > niter:
                break
            model.train()
            r = trainer.step(batch["LQ"].to(dev, non_blocking=True), batch["GT"].to(dev, non_blocking=True))
            if rank == 0 and (step % log_freq == 0 or step == 1):
                print(f"[train] step {step} lr {r['lr']:.3e} rec_loss {float(r['rec_loss']):.4e} ref_loss {float(r['ref_loss']):.4e} "
                      f"({(time.time() - t0) / max(step - int(opt['train']['current_step'] or 0), 1):.3f} s/step)", flush=True)
            model.eval()
            if val_set is not None and val_freq and step % val_freq == 0:
                from output_GPEMSR import save_img
                psnr = validate_psnr(model, val_set, scale, dev, rank, world, save_dir=osp.join(out_dir, "val", str(step)), save_img=save_img)
                if rank == 0:
                    print(f"# Validation # PSNR: {psnr:.4e},current_step:{step}", flush=True)
            if save_freq and step % save_freq == 0 and rank == 0:
                torch.save({k: v.cpu() for k, v in model.state_dict().items()}, osp.join(out_dir, "models", f"{step}_G.pth"))
                torch.save({"epoch": epoch, "iter": step, "optimizers": {"0": trainer.torch_optimizer_state_dict()},
                            "schedulers": {"0": dict(vars(trainer.sched))}}, osp.join(out_dir, "training_state", f"{step}.state"))
        epoch += 1
    if rank == 0:
        print("End of training.", flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
