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
    kinks move single elements when rounding differs)."""
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
