"""The training legs of bench.py (kept beside it, outside the package: the CPU-baseline leg below is the only code outside
tests/ and smoke() that touches oracle/).  `bench.py --mode train`: BASELINE.json configs[4] -- the 8x EMSR stage-3 training step (forward + backward incl. the VGG
contextual loss + Adam), batch 8 per GPU of the reference's training crops (LR 32x32 -> 256x256, option/train_stage3_x8.yml),
one process per GPU, one RCCL all-reduce of the flat gradient buffer per step (weak scaling).  Same timing protocol as the
forward bench: W warm-up steps, then exactly K steps between barrier + synchronize pairs, max over ranks."""
from __future__ import annotations

import json
import os
import time

import torch

PEAK_F32_MATRIX_TFLOPS = 157.3
TRAIN_OPT = dict(lr_G=4e-4, beta1=0.9, beta2=0.99, lr_scheme="CosineAnnealingLR_Restart", T_period=[40000, 80000, 120000, 120000, 120000],
                 restarts=[40000, 120000, 240000, 360000], restart_weights=[1, 1, 1, 1], eta_min=1e-7, rec_loss_factor=1,
                 ref_loss_factor=0.001)                     # option/train_stage3_x8.yml:90-108


def train_leg(root: str, scale: int, rank: int, world: int, dev, steps: int = 10, warmup: int = 2, batch: int = 8, lr: int = 32,
              precision: str = "fp32") -> dict:
    """BASELINE configs[4] as a compact leg of the default `python bench.py` line: the stage-3 training step (R:train_stage3.py:343-366:
    forward, L1 + 0.001 x contextual(VGG relu3_4) loss, backward, gradient all-reduce at N > 1, Adam), batch `batch` per GPU of 5x1xlrxlr crops.
    Timed like every leg (K steps between barrier + synchronize pairs, no per-launch events, max over ranks); the executed fraction of the
    f32 matrix peak comes from a second pass with HIP events around every convolution / weight-gradient launch."""
    from gpemsr_amd import ops
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    opt = load_options(os.path.join(root, "option", f"output_GPEMSR_x{scale}.yml"))
    model = build_model(opt, load_prior_files=False, precision=precision).to(dev)
    trainer = Stage3Trainer(model, TRAIN_OPT, dev, world=world)
    LR = synth_lr_tiles(batch, 5, lr, lr, seed=2000 + rank, kind="smooth").to(dev)
    GT = torch.rand(batch, 1, lr * scale, lr * scale, generator=torch.Generator().manual_seed(3000 + rank)).to(dev)
    for _ in range(warmup):
        trainer.step(LR, GT)

    def timed_pass(k):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            o_ = trainer.step(LR, GT)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        d_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d_], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d_ = float(tt.item())
        return d_, o_
    dt, o = timed_pass(steps)
    psteps = max(1, min(steps, 3))
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    try:
        dtp, _ = timed_pass(psteps)
    finally:
        ops.PROFILER = None
    summ = prof.summary()
    fam = [summ[k] for k in ("conv_mfma", "conv_wgrad") if k in summ]
    ms = sum(v["ms"] for v in fam)
    ex = sum(v["executed"] for v in fam)
    fl = sum(v["flops"] for v in fam)
    by_name = {k: v for k, v in prof.summary(by_name=True).items() if v["family"] in ("conv_mfma", "conv_wgrad")}
    dom_name, dom = max(by_name.items(), key=lambda kv: kv[1]["ms"]) if by_name else ("?", {"ms": 0.0, "executed": 0.0, "launches": 0})
    etf = lambda d_: d_["executed"] / (d_["ms"] * 1e-3) / 1e12 if d_["ms"] > 0 else 0.0      # noqa: E731
    res = {"value": round(world * batch * steps / dt, 3), "unit": "training samples/s", "ms_per_step": round(1e3 * dt / steps, 2), "steps": steps,
           "n_gpus": world, "batch_per_gpu": batch, "dtype": "f32" if precision == "fp32" else precision,
           "config": f"{scale}x EMSR stage-3 training step, batch {batch}/GPU of 5x1x{lr}x{lr} -> {lr * scale}^2 crops (BASELINE.json configs[4]; the reference's "
                     "step has no discriminator)",
           "kernel": dom_name, "frac": round(etf(dom) / PEAK_F32_MATRIX_TFLOPS, 4),
           "kernel_avg_launch_us": round(1e3 * dom["ms"] / max(dom["launches"], 1), 2),
           "family_frac": round((ex / (ms * 1e-3) / 1e12 if ms > 0 else 0.0) / PEAK_F32_MATRIX_TFLOPS, 4),
           "effective_frac": round((fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0) / PEAK_F32_MATRIX_TFLOPS, 4),
           "family": "f32 MFMA convolution family of the step: forward + data gradients (conv_mfma) and weight gradients (conv_wgrad)",
           "family_time_share_of_step": round(ms * 1e-3 / dtp, 3), "profiled_pass_ms_per_step": round(1e3 * dtp / psteps, 2),
           "losses_last_step": {"rec": float(o["rec_loss"].item()), "ref": float(o["ref_loss"].item())}}
    del trainer, model
    return res


def run(args, root: str, effective_cores):
    from gpemsr_amd import dist as gdist, ops
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer

    rank, world, local = gdist.init_from_env(backend=getattr(args, "backend", "") or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
    if os.environ.get("GPEMSR_BENCH_SHARE_GPU") == "1":       # rehearsal on a box with fewer GPUs than ranks (--backend gloo)
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    s = args.scale
    opt = load_options(os.path.join(root, "option", f"output_GPEMSR_x{s}.yml"))
    model = build_model(opt, load_prior_files=False, precision=args.precision).to(dev)
    trainer = Stage3Trainer(model, TRAIN_OPT, dev, world=world)
    B = args.train_batch
    lr = args.train_lr
    LR = synth_lr_tiles(B, 5, lr, lr, seed=2000 + rank, kind="smooth").to(dev)
    GT = torch.rand(B, 1, lr * s, lr * s, generator=torch.Generator().manual_seed(3000 + rank)).to(dev)

    for _ in range(args.warmup):
        trainer.step(LR, GT)
    def timed_pass():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o_ = trainer.step(LR, GT)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        d_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d_], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d_ = float(tt.item())
        return d_, o_
    # the timed region carries no per-launch events (a training step is ~2,300 launches: a pair of timing events around each
    # convolution costs the step several per cent); the roofline fields come from a second, profiled pass of the same K steps
    dt, o = timed_pass()
    prof = ops.LaunchProfiler()
    dt_prof = dt
    if not args.no_profile:
        ops.PROFILER = prof
        dt_prof, _ = timed_pass()
        ops.PROFILER = None
    value = world * B * args.steps / dt
    n_params = trainer.n_params
    summ = prof.summary()
    fam = {k: summ.get(k, {"launches": 0, "ms": 0.0, "flops": 0.0}) for k in ("conv_mfma", "conv_wgrad")}
    split = summ.get("conv_split")
    flops = sum(v["flops"] for v in fam.values())
    ms = sum(v["ms"] for v in fam.values())
    launches = sum(v["launches"] for v in fam.values())
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    def tf(d):
        return round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2) if d["ms"] > 0 else 0.0
    roofline = {
        "bound": "mfma", "kernel": "f32 MFMA convolution family of the step: conv_mfma_kernel (forward + data gradients) and wgrad_kernel "
                                   "(weight gradients), v_mfma_f32_32x32x2_f32",
        "achieved": round(achieved, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4),
        "traffic": None,
        "algorithmic_gflop_per_launch": round(flops / 1e9 / max(launches, 1), 3), "launches_per_step": launches // max(args.steps, 1),
        "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
        "forward_and_dgrad_tflops": tf(fam["conv_mfma"]), "wgrad_tflops": tf(fam["conv_wgrad"]),
        "algorithmic_gflop_per_sample": round(flops / 1e9 / (B * args.steps), 1),
        "kernel_time_share_of_step": round(ms * 1e-3 / dt_prof, 3),
        "profiled_pass_ms_per_step": round(1e3 * dt_prof / args.steps, 2),
        "note": "value / ms_per_step: K steps without per-launch events; this block: a second pass of the same K steps with HIP events around "
                "every convolution launch",
    }
    if split is not None:
        roofline["split_bf16_kernel"] = {"achieved_algorithmic_tflops": tf(split), "launches_per_step": split["launches"] // max(args.steps, 1),
                                         "time_share_of_step": round(split["ms"] * 1e-3 / dt, 3)}
    if args.layer_report and rank == 0:
        rows = sorted(prof.summary(by_tag=True).items(), key=lambda kv: -kv[1]["ms"])
        with open(args.layer_report, "w") as f:
            f.write("kernel\ttag\tlaunches\tms_total\tGFLOP\tTFLOP/s\n")
            for (kern, tag), d in rows:
                f.write(f"{kern}\t{tag}\t{d['launches']}\t{d['ms']:.3f}\t{d['flops'] / 1e9:.1f}\t{tf(d)}\n")

    cpu_baseline = None
    losses_fp32 = {"rec": float(o["rec_loss"].item()), "ref": float(o["ref_loss"].item())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # bounded sample: ONE training sample (forward + both losses + backward) through the CPU oracle under torch autograd
        from oracle import gpemsr_oracle as orc
        cores = effective_cores()
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        for k, p in model.named_parameters():
            if p.requires_grad:
                sd[k].requires_grad_(True)
        xc, gc = LR[:1].cpu(), GT[:1].cpu()
        t1 = time.perf_counter()
        out, ref = orc.gpemsr_forward(sd, xc, scale=s)
        rec, refl, _ = orc.stage3_losses(sd, out, ref.detach(), gc)
        (rec * TRAIN_OPT["rec_loss_factor"] + TRAIN_OPT["ref_loss_factor"] * refl).backward()
        cdt = time.perf_counter() - t1
        cpu_baseline = {"value": round(1.0 / cdt, 5), "unit": "training samples/s", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"1 sample [1,5,1,{lr},{lr}] -> {lr * s}^2: forward + L1 + contextual loss + backward ({cdt:.1f} s) of "
                                  "oracle/gpemsr_oracle.py under torch CPU autograd (no optimizer step)"}
    extras = None
    if world == 1 and args.precision == "fp32" and not args.no_extras:
        # the same step with the FORWARD convolutions of the frozen sub-networks (VQGAN prior, VGG, SpyNet: about two thirds of
        # the step's FLOPs, constants of the backward) on the split-bf16 kernel; trainable layers and all gradients stay f32
        del trainer, model
        torch.cuda.empty_cache()
        # exact-fp32 step with the FROZEN 3x3 layers (prior, mask, flow and loss networks: forward and data gradients) in the Winograd
        # F(2x2,3x3) form (train option `winograd_frozen`, off by default: see gpemsr_amd/train.py TrainEngine)
        mw = build_model(opt, load_prior_files=False, precision="fp32").to(dev)
        tw = Stage3Trainer(mw, dict(TRAIN_OPT, winograd_frozen=True), dev, world=world)
        for _ in range(max(args.warmup, 1)):
            tw.step(LR, GT)
        torch.cuda.synchronize()
        tq = time.perf_counter()
        for _ in range(args.steps):
            ow = tw.step(LR, GT)
        torch.cuda.synchronize()
        dw = time.perf_counter() - tq
        extras_w = {"value": round(B * args.steps / dw, 3), "unit": "samples/s", "ms_per_step": round(1e3 * dw / args.steps, 2),
                    "dtype": "f32; frozen 3x3 stride-1 layers (forward and data gradients) in the Winograd forms (F(4x4,3x3) / F(2x2,3x3)) on the f32 "
                             "matrix pipe: median gradient distance to float64 on the two goldens 2.1e-4 / 6.8e-5 (direct forms: 5.9e-4 / 1.5e-5; "
                             "the reference's own: 3.6e-5 / 1.4e-4; profiles/r05_winograd_training_distance.log)",
                    "losses_last_step": {"rec": float(ow["rec_loss"].item()), "ref": float(ow["ref_loss"].item())},
                    "speedup_vs_fp32_step": round((dt / args.steps) / (dw / args.steps), 3)}
        del tw, mw
        torch.cuda.empty_cache()
        m3 = build_model(opt, load_prior_files=False, precision="bf16x3").to(dev)
        t3 = Stage3Trainer(m3, TRAIN_OPT, dev, world=world)
        for _ in range(max(args.warmup, 1)):
            t3.step(LR, GT)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            o3 = t3.step(LR, GT)
        torch.cuda.synchronize()
        d3 = time.perf_counter() - ta
        extras = {"fp32_winograd_frozen": extras_w, "bf16x3_frozen_forward": {
            "value": round(B * args.steps / d3, 3), "unit": "samples/s", "ms_per_step": round(1e3 * d3 / args.steps, 2),
            "dtype": "frozen sub-networks' forward convolutions: fp32 operands split hi+lo bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulate "
                     "(fp32-grade, DESIGN_HISTORY.md §3.3); trainable layers, data and weight gradients: f32 MFMA",
            "losses_last_step": {"rec": float(o3["rec_loss"].item()), "ref": float(o3["ref_loss"].item())},
            "speedup_vs_fp32_step": round((dt / args.steps) / (d3 / args.steps), 3)}}
        # ... and with the frozen sub-networks that carry no gradient (VQGAN prior, VGG mask, SpyNet) on the bf16 DATA PATH of the
        # inference engine (BASELINE configs[2]'s kernels: bf16 activations, fused VGG mask, flash attention); the rest as above
        del t3, m3
        torch.cuda.empty_cache()
        m4 = build_model(opt, load_prior_files=False, precision="bf16").to(dev)
        t4 = Stage3Trainer(m4, TRAIN_OPT, dev, world=world)
        for _ in range(max(args.warmup, 1)):
            t4.step(LR, GT)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(args.steps):
            o4 = t4.step(LR, GT)
        torch.cuda.synchronize()
        d4 = time.perf_counter() - tb
        extras["bf16_frozen_subnetworks"] = {
            "value": round(B * args.steps / d4, 3), "unit": "samples/s", "ms_per_step": round(1e3 * d4 / args.steps, 2),
            "dtype": "VQGAN prior, VGG relu1_2 mask and SpyNet (frozen, no gradient through them): bf16 activations in HBM + bf16 MFMA; trainable "
                     "layers and the loss network: fp32 activations, forward products bf16x3 (fp32-grade); data and weight gradients: f32 MFMA",
            "losses_last_step": {"rec": float(o4["rec_loss"].item()), "ref": float(o4["ref_loss"].item())},
            "speedup_vs_fp32_step": round((dt / args.steps) / (d4 / args.steps), 3)}
        model = m4
    if rank == 0:
        line = {
            "metric": f"stage-3 training samples/sec, {s}x EMSR (LR {lr}x{lr} -> {lr * s}x{lr * s} crops), batch {B}/GPU",
            "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
            "rccl_world": torch.distributed.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "forward convolutions bf16x3 (split hi+lo bf16 MFMA, fp32 accumulate, fp32-grade); gradients f32",
            "data": "synthetic",
            "config": {"workload": f"{s}x EMSR stage-3 training step (train_stage3.py:343-366): forward, L1 + 0.001 x contextual(VGG relu3_4) loss, "
                                   f"backward, Adam; batch {B}/GPU of 5x1x{lr}x{lr} LR crops (BASELINE.json configs[4]; the reference's step has "
                                   "no discriminator)", "batch_per_gpu": B, "global_batch": B * world, "lr": lr, "scale": s,
                       "trainable_parameters": n_params,
                       "weights": "deterministic synthetic init", "parallelism": f"data parallel over {world} GPU(s), one RCCL all-reduce of the "
                       f"flat gradient buffer ({n_params * 4 / 1e6:.1f} MB) per step" if world > 1 else "single GPU"},
            "losses_last_step": losses_fp32,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extras": extras,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def run_stage2(args, root: str, effective_cores):
    """`bench.py --mode train2`: the stage-2 (indexer) training step, train_stage2.py:351-366, at the reference's geometry
    (option/train_stage2_x8.yml: batch 8, GT 1024x1024 -> LR 128x128), same timing protocol."""
    from gpemsr_amd import dist as gdist, ops
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train_stage2 import Stage2Trainer

    rank, world, local = gdist.init_from_env(backend=getattr(args, "backend", "") or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
    if os.environ.get("GPEMSR_BENCH_SHARE_GPU") == "1":       # rehearsal on a box with fewer GPUs than ranks (--backend gloo)
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    s = args.scale
    opt = load_options(os.path.join(root, "option", f"output_GPEMSR_x{s}.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)
    topt = dict(TRAIN_OPT, restarts=[40000, 80000, 240000, 360000])          # option/train_stage2_x8.yml:78-86
    trainer = Stage2Trainer(model, topt, dev, world=world)
    B, lr = args.train_batch, args.stage2_lr
    LR = synth_lr_tiles(B, 1, lr, lr, seed=4000 + rank, kind="smooth")[:, 0].contiguous().to(dev)
    GT = synth_lr_tiles(B, 1, lr * s, lr * s, seed=5000 + rank, kind="smooth")[:, 0].contiguous().to(dev)
    for _ in range(args.warmup):
        trainer.step(LR, GT)
    def timed_pass():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o_ = trainer.step(LR, GT)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        d_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d_], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d_ = float(tt.item())
        return d_, o_
    dt, o = timed_pass()                      # no per-launch events inside the timed region; the roofline block: a second, profiled pass
    prof = ops.LaunchProfiler()
    dt_prof = dt
    if not args.no_profile:
        ops.PROFILER = prof
        dt_prof, _ = timed_pass()
        ops.PROFILER = None
    summ = prof.summary()
    fam = {k: summ.get(k, {"launches": 0, "ms": 0.0, "flops": 0.0}) for k in ("conv_mfma", "conv_wgrad")}
    flops, ms = sum(v["flops"] for v in fam.values()), sum(v["ms"] for v in fam.values())
    launches = sum(v["launches"] for v in fam.values())
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    def tf(d):
        return round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2) if d["ms"] > 0 else 0.0
    if args.layer_report and rank == 0:
        rows = sorted(prof.summary(by_tag=True).items(), key=lambda kv: -kv[1]["ms"])
        with open(args.layer_report, "w") as f:
            f.write("kernel\ttag\tlaunches\tms_total\tGFLOP\tTFLOP/s\n")
            for (kern, tag), d in rows:
                f.write(f"{kern}\t{tag}\t{d['launches']}\t{d['ms']:.3f}\t{d['flops'] / 1e9:.1f}\t{tf(d)}\n")
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import gpemsr_oracle as orc
        torch.set_num_threads(effective_cores())
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        for k in list(sd):
            if k.startswith("refmodel.indexer."):
                sd[k].requires_grad_(True)
        t1 = time.perf_counter()
        loss, _, _ = orc.stage2_loss(sd, LR[:1].cpu(), GT[:1].cpu())
        loss.backward()
        cdt = time.perf_counter() - t1
        cpu_baseline = {"value": round(1.0 / cdt, 5), "unit": "training samples/s", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"1 sample (LR {lr}^2, GT {lr * s}^2): encoder + nearest code + indexer + cross-entropy + backward ({cdt:.1f} s) of "
                                  "oracle/gpemsr_oracle.py under torch CPU autograd"}
    if rank == 0:
        print(json.dumps({
            "metric": f"stage-2 (indexer) training samples/sec, {s}x (GT {lr * s}x{lr * s} -> LR {lr}x{lr}), batch {B}/GPU",
            "value": round(world * B * args.steps / dt, 3), "unit": "samples/s", "n_gpus": world,
            "rccl_world": torch.distributed.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"stage-2 training step (train_stage2.py:351-366): frozen Encoder(GT) + nearest codebook vector -> targets; "
                                   f"Indexer{s}(LR) -> logits; cross-entropy, backward, Adam; batch {B}/GPU (option/train_stage2_x{s}.yml geometry)",
                       "batch_per_gpu": B, "lr": lr, "scale": s, "trainable_parameters": trainer.n_params,
                       "parallelism": f"data parallel over {world} GPU(s), one RCCL all-reduce of the flat gradient buffer" if world > 1 else "single GPU"},
            "loss_last_step": float(o["loss"].item()),
            "roofline": {"bound": "mfma", "kernel": "f32 MFMA convolution family (forward incl. attention GEMMs, data gradients, wgrad_kernel)",
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4), "traffic": None, "launches_per_step": launches // max(args.steps, 1),
                         "forward_and_dgrad_tflops": tf(fam["conv_mfma"]), "wgrad_tflops": tf(fam["conv_wgrad"]),
                         "kernel_time_share_of_step": round(ms * 1e-3 / dt_prof, 3),
                         "profiled_pass_ms_per_step": round(1e3 * dt_prof / max(args.steps, 1), 2),
                         "note": "value / ms_per_step: no per-launch events in the timed region; this block: a second pass with HIP events around every convolution launch"},
            "cpu_baseline": cpu_baseline}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def run_stage1(args, root: str, effective_cores):
    """`bench.py --mode train1`: the stage-1 (VQGAN) training step, train_stage1.py:291-357, at the reference's geometry
    (option/train_stage1.yml: batch 8, 512x512 crops), same timing protocol.  `value` = the ADVERSARIAL phase (current_step > gan_start:
    generator step with the GAN term + discriminator step, R1 penalty every net_d_reg_every = 16 steps; the timed steps are whole
    16-step cycles so that exactly 1/16 of them carry the penalty); the generator phase (the first 40,000 steps) is in `extras`."""
    from gpemsr_amd import dist as gdist, ops
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.discriminator import Discriminator
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train_stage1 import Stage1Trainer

    rank, world, local = gdist.init_from_env(backend=getattr(args, "backend", "") or None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path)"
    if os.environ.get("GPEMSR_BENCH_SHARE_GPU") == "1":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    opt = load_options(os.path.join(root, "option", "output_GPEMSR_x8.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)
    # option/train_stage1.yml train: block
    topt = dict(lr_G=1e-4, lr_D=4e-4, beta1=0.9, beta2=0.99, T_period=[40000, 80000, 120000, 120000, 120000], restarts=[40000, 120000, 240000, 360000],
                restart_weights=[1, 1, 1, 1], eta_min=1e-7, rec_loss_factor=1.0, codebook_loss_factor=1.0, gan_start=40000, gan_loss_factor=0.05,
                generator_update_rate=1, r1_reg_weight=1e-4, net_d_reg_every=16)
    disc = Discriminator(dict(im_channel=1, num_filters_last=64, n_layers=3), init_seed=0).to(dev)
    trainer = Stage1Trainer(model, topt, dev, beta=1.0, world=world, discriminator=disc)
    B, size = args.train_batch, args.stage1_size
    imgs = synth_lr_tiles(B, 1, size, size, seed=6000 + rank, kind="smooth")[:, 0].contiguous().to(dev)

    def timed(first_step, steps):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            o = trainer.step(imgs, current_step=first_step + i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, o

    for i in range(args.warmup):
        trainer.step(imgs, current_step=1 + i)
    gsteps = max(args.steps // 2, 1)
    gdt, go = timed(100, gsteps)                                                         # generator phase
    for i in range(max(args.warmup, 1)):
        trainer.step(imgs, current_step=40016 + i)                                       # warm the discriminator kernels, incl. one R1 step
    cycles = max(1, (args.steps + 15) // 16)
    # the timed region carries no per-launch events (~1,700 launches per step: timing events around every convolution cost this step
    # 25 %); the roofline block comes from a second, profiled pass over the same 16-step cycle
    dt, o = timed(40033, 16 * cycles)                                                    # 40048 % 16 == 0: one R1 step per 16
    prof = ops.LaunchProfiler()
    dt_prof = dt
    if not args.no_profile:
        ops.PROFILER = prof
        dt_prof, _ = timed(40033 + 16 * cycles, 16 * cycles)
        ops.PROFILER = None
    steps = 16 * cycles
    summ = prof.summary()
    fam = {k: summ.get(k, {"launches": 0, "ms": 0.0, "flops": 0.0}) for k in ("conv_mfma", "conv_wgrad")}
    flops, ms = sum(v["flops"] for v in fam.values()), sum(v["ms"] for v in fam.values())
    launches = sum(v["launches"] for v in fam.values())
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0

    def tf(d):
        return round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2) if d["ms"] > 0 else 0.0
    if args.layer_report and rank == 0:
        rows = sorted(prof.summary(by_tag=True).items(), key=lambda kv: -kv[1]["ms"])
        with open(args.layer_report, "w") as f:
            f.write("kernel\ttag\tlaunches\tms_total\tGFLOP\tTFLOP/s\n")
            for (kern, tag), d in rows:
                f.write(f"{kern}\t{tag}\t{d['launches']}\t{d['ms']:.3f}\t{d['flops'] / 1e9:.1f}\t{tf(d)}\n")
    if rank == 0:
        print(json.dumps({
            "metric": f"stage-1 (VQGAN) training samples/sec, adversarial phase, {size}x{size} crops, batch {B}/GPU",
            "value": round(world * B * steps / dt, 3), "unit": "samples/s", "n_gpus": world,
            "rccl_world": torch.distributed.get_world_size() if world > 1 else 1, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"stage-1 training step (train_stage1.py:291-357), current_step > gan_start: Encoder -> Codebook -> Decoder, L1 + codebook "
                                   f"loss + 0.05 * mean(-D(decoded)), backward, Adam (G); 0.5 * (mean(-D(imgs)) + mean(D(decoded))) backward, R1 penalty "
                                   f"(gradient of a gradient) on 1 step in 16, Adam (D); batch {B}/GPU, {size}x{size} (option/train_stage1.yml geometry)",
                       "batch_per_gpu": B, "size": size, "trainable_parameters": trainer.n_params,
                       "discriminator_parameters": sum(p.numel() for p in disc.parameters()),
                       "parallelism": f"data parallel over {world} GPU(s), RCCL all-reduce of the two flat gradient buffers" if world > 1 else "single GPU"},
            "losses_last_step": {k: float(v.item()) for k, v in o.items() if hasattr(v, "item")},
            "roofline": {"bound": "mfma", "kernel": "f32 MFMA convolution family (forward incl. attention and discriminator GEMMs, data gradients, wgrad_kernel)",
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MATRIX_TFLOPS, 4), "traffic": None, "launches_per_step": launches // max(steps, 1),
                         "forward_and_dgrad_tflops": tf(fam["conv_mfma"]), "wgrad_tflops": tf(fam["conv_wgrad"]),
                         "kernel_time_share_of_step": round(ms * 1e-3 / dt_prof, 3),
                         "profiled_pass_ms_per_step": round(1e3 * dt_prof / max(steps, 1), 2),
                         "note": "value / ms_per_step: no per-launch events in the timed region; this block: a second pass with HIP events around every convolution launch"},
            "extras": {"generator_phase": {"value": round(world * B * gsteps / gdt, 3), "unit": "samples/s", "ms_per_step": round(1e3 * gdt / gsteps, 2),
                                           "steps": gsteps, "what": "current_step <= gan_start (train_stage1.py:313-326): no discriminator"}},
            "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
