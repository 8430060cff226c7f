"""CPU side of the stage-3 training step: the oracle's autograd against the reference golden, the scalar LR schedulers
against the reference's own scheduler classes (values emitted by oracle/gen_golden_train.py), and the 2-rank gloo
gradient averaging that Stage3Trainer.step uses for world > 1."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_schedulers_match_reference(golden_dir):
    from gpemsr_amd.train import CosineAnnealingLRRestart, MultiStepLRRestart
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    sc = CosineAnnealingLRRestart(4e-4, [6, 10, 8], restarts=[6, 16], weights=[1, 0.5], eta_min=1e-7)
    got = np.array([sc.step() for _ in range(len(d["sched_cosine"]))])
    assert np.allclose(got, d["sched_cosine"], rtol=1e-12, atol=0), (got, d["sched_cosine"])
    ms = MultiStepLRRestart(2e-4, [3, 7, 7, 12], restarts=[9], weights=[0.5], gamma=0.5)
    got = np.array([ms.step() for _ in range(len(d["sched_multistep"]))])
    assert np.allclose(got, d["sched_multistep"], rtol=1e-12, atol=0), (got, d["sched_multistep"])


def test_oracle_autograd_matches_reference_golden(golden_dir):
    """The CPU restatement (forward + stage3_losses) under torch autograd against the reference's training step: losses and
    the gradient statistics of all trainable tensors.  Tolerances as in tests/test_train_gpu.py (the gradient is piecewise:
    kinks move single elements when rounding differs, DESIGN_HISTORY.md §3.6)."""
    import yaml
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.arch import param_specs
    from gpemsr_amd.synth import synth_state_dict
    from oracle import gpemsr_oracle as orc
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    opt = yaml.safe_load(open(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml")))
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=8, **kw), seed=0)
    names = [str(n) for n in d["grad_names"]]
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    out, ref = orc.gpemsr_forward(sd, torch.from_numpy(d["LR"]), scale=8, forced_idx=torch.from_numpy(d["code_idx"]).long(),
                                  forced_flow=torch.from_numpy(d["flow"]))
    rec, refl, _ = orc.stage3_losses(sd, out, ref.detach(), torch.from_numpy(d["GT"]))
    (rec * TRAIN_OPT["rec_loss_factor"] + TRAIN_OPT["ref_loss_factor"] * refl).backward()
    assert abs(rec.item() - float(d["rec_loss_1"])) <= 1e-6 * float(d["rec_loss_1"])
    assert abs(refl.item() - float(d["ref_loss_1"])) <= 1e-5 * float(d["ref_loss_1"])
    downstream = ("recon_trunk.", "upconv", "HRconv", "conv_last")
    errs = {}
    for i, k in enumerate(names):
        want = d["grad_stats"][i]
        if want[0] == 0.0:
            assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0
            continue
        g = sd[k].grad.reshape(-1).double()
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    for k, e in errs.items():
        assert e <= (1e-3 if k.startswith(downstream) else 6e-2), f"{k}: {e:.2e}"
    assert np.median(list(errs.values())) <= 3e-3 and np.percentile(list(errs.values()), 90) <= 1.5e-2


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from gpemsr_amd import dist as gdist
    gdist.init_from_env(backend="gloo")
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)
    gdist.average_gradients(g, world)
    ok = torch.allclose(g, torch.arange(10, dtype=torch.float32) * 1.5)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def test_gradient_averaging_gloo_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_oracle_stage2_step_matches_reference_golden(golden_dir):
    """Indexer training step (train_stage2.py:351-366): the oracle's encoder / nearest-code / indexer / cross-entropy under
    autograd against the reference golden (oracle/gen_golden_stage2.py)."""
    import yaml
    from train_constants import projection
    from gpemsr_amd.arch import param_specs
    from gpemsr_amd.synth import synth_state_dict
    from oracle import gpemsr_oracle as orc
    d = np.load(os.path.join(golden_dir, "stage2_x8.npz"))
    opt = yaml.safe_load(open(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml")))
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    sd = synth_state_dict(param_specs(scale=8, **kw), seed=0)
    names = [str(n) for n in d["grad_names"]]
    for k in names:
        sd["refmodel." + k] = sd["refmodel." + k].clone().requires_grad_(True)
    loss, logits, target = orc.stage2_loss(sd, torch.from_numpy(d["LR"]), torch.from_numpy(d["GT"]))
    assert torch.equal(target.to(torch.int32), torch.from_numpy(d["target_idx"]))          # free-running arg-min agrees
    assert abs(loss.item() - float(d["loss_1"])) <= 1e-5 * float(d["loss_1"])
    assert np.abs(logits.detach().numpy()[::4] - d["logits_1_every4"]).max() <= 1e-4 * np.abs(d["logits_1_every4"]).max()
    loss.backward()
    errs = []
    for i, k in enumerate(names):
        g = sd["refmodel." + k].grad.reshape(-1).double()
        want = d["grad_stats"][i]
        errs.append(max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0])
    assert max(errs) <= 2e-2 and np.median(errs) <= 1e-3, (max(errs), np.median(errs))


# ---------------------------------------------------------------------------------------------------------------------
# The boundary of BASELINE configs[4]: option/train_stage3_x{8,16}.yml and the validation loop of R:train_stage3.py:199-317
# ---------------------------------------------------------------------------------------------------------------------
def _flat_keys(d, prefix=""):
    out = []
    for k, v in d.items():
        p = f"{prefix}.{k}" if prefix else str(k)
        out.append(p)
        if isinstance(v, dict):
            out += _flat_keys(v, p)
    return out


@pytest.mark.parametrize("s", [8, 16])
def test_training_option_files_follow_the_reference_contract(golden_dir, s):
    """Every key of the reference's option/train_stage3_x{s}.yml (key set and `train:` block emitted by oracle/gen_golden_options.py
    from the reference's own files) is present in ours with the same value where the value is arithmetic; our extra keys are additive."""
    import json
    import yaml
    g = json.load(open(os.path.join(golden_dir, "train_options.json")))[f"x{s}"]
    opt = yaml.safe_load(open(os.path.join(ROOT, "option", f"train_stage3_x{s}.yml")))
    mine = set(_flat_keys(opt))
    missing = [k for k in g["keys"] if k not in mine]
    assert not missing, missing
    assert mine - set(g["keys"]) <= {"precision", "synthetic_data_if_missing"}
    assert opt["train"] == g["train"], (opt["train"], g["train"])
    assert opt["scale"] == g["scale"] and opt["val"]["val_freq"] == g["val"]["val_freq"]
    tr = opt["datasets"]["train"]
    assert (tr["batch_size"], tr["GT_size"], tr["LQ_size"]) == (g["batch_size"], g["GT_size"], g["LQ_size"])
    ref = f"/root/reference/GPEMSR-CREMI/GPEMSR/option/train_stage3_x{s}.yml"
    if os.path.exists(ref):                              # build container only: the reference's file itself loads through our loader
        from gpemsr_amd.config import load_options
        assert load_options(ref)["train"] == g["train"]


@pytest.mark.parametrize("s", [8, 16])
def test_train_block_drives_the_scheduler_like_the_reference(golden_dir, s):
    """`train:` block -> gpemsr_amd.train.make_scheduler (what Stage3Trainer builds) against the learning rates of the reference's own
    CosineAnnealingLR_Restart stepped 480,000 times with the same block (sampled at the first steps, around every restart, every
    5000th step and the end; oracle/gen_golden_options.py)."""
    import json
    import yaml
    from gpemsr_amd.train import make_scheduler
    g = json.load(open(os.path.join(golden_dir, "train_options.json")))[f"x{s}"]
    tr = yaml.safe_load(open(os.path.join(ROOT, "option", f"train_stage3_x{s}.yml")))["train"]
    sc = make_scheduler(tr)
    want = {int(k): v for k, v in g["lr_trace"].items()}
    worst = 0.0
    for step in range(1, int(tr["niter"]) + 1):
        lr = sc.step()
        if step in want:
            worst = max(worst, abs(lr - want[step]) / want[step])
    assert worst < 1e-9, worst
    with pytest.raises(NotImplementedError):
        make_scheduler(dict(tr, lr_scheme="StepLR"))


class _BilinearSR:
    """Stand-in SR model for the validation loop's host logic: bilinear x scale of the centre slice (per crop)."""

    def __init__(self, scale):
        self.scale, self.calls = scale, 0

    def __call__(self, x):
        self.calls += 1
        return torch.nn.functional.interpolate(x[:, x.shape[1] // 2], scale_factor=self.scale, mode="bilinear", align_corners=False), None


def _val_worker(rank, world, port, q):
    import torch.distributed as dist
    from gpemsr_amd.data import SyntheticCrops
    from gpemsr_amd.validate import validate_psnr
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    ds = SyntheticCrops(5, 8, 16, seed=3)
    m = _BilinearSR(8)
    p = validate_psnr(m, ds, 8, torch.device("cpu"), rank, world)
    q.put((rank, p, m.calls))
    dist.destroy_process_group()


def test_validation_loop_matches_the_reference_recipe_and_shards_over_ranks():
    """R:train_stage3.py:199-317: quadrant crops -> tensor2img -> PSNR on uint8, sample idx on rank idx % world, PSNR vector reduced onto
    rank 0, mean.  One process and two gloo ranks must agree with a direct restatement of the recipe."""
    from gpemsr_amd.data import SyntheticCrops
    from gpemsr_amd.imgutil import calculate_psnr, tensor2img
    from gpemsr_amd.validate import validate_psnr
    ds = SyntheticCrops(5, 8, 16, seed=3)
    m = _BilinearSR(8)
    got = validate_psnr(m, ds, 8, torch.device("cpu"))
    assert m.calls == 4 * len(ds)
    want = []
    for i in range(len(ds)):
        it = ds[i]
        LQ, H, W = it["LQ"].unsqueeze(0), it["LQ"].shape[2], it["LQ"].shape[3]
        SR = np.zeros((H * 8, W * 8), np.uint8)
        for ys in (slice(0, H // 2), slice(H // 2, H)):
            for xs in (slice(0, W // 2), slice(W // 2, W)):
                sr, _ = _BilinearSR(8)(LQ[:, :, :, ys, xs])
                SR[ys.start * 8:ys.stop * 8, xs.start * 8:xs.stop * 8] = tensor2img(sr)
        want.append(calculate_psnr(tensor2img(it["GT"]), SR))
    assert abs(got - float(np.mean(np.array(want, dtype=np.float32)))) < 1e-4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_val_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
    assert res[0][1] is not None and abs(res[0][1] - got) < 1e-4 and res[1][1] is None
    assert res[0][2] == 4 * 3 and res[1][2] == 4 * 2          # samples 0, 2, 4 | 1, 3
