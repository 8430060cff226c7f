"""Stage-1 (VQGAN) training, generator phase (R:train_stage1.py:313-326), through the C ABI against vectors emitted by the UNMODIFIED
reference ``model/vqgan.py::Generator`` (oracle/gen_golden_stage1.py -> tests/golden/stage1_gen.npz): losses, the decoded image, the
gradient of every generator tensor (encoder, codebook, decoder) and two Adam steps.  Code indices are teacher-forced (the arg-min over
1024 codes is discontinuous; the smallest top-2 margin of the reference run is 5e-4)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(got, want, tol, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-12)
    assert err <= tol * ref, f"{what}: max err {err:.3e} vs {ref:.3e} (tol {tol})"


def _trainer(d, dev):
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train_stage1 import Stage1Trainer
    lr_g, b1, b2, rec_f, cb_f, beta = [float(v) for v in d["train_opt"]]
    topt = dict(lr_G=lr_g, beta1=b1, beta2=b2, T_period=[40000, 80000, 120000, 120000, 120000], restarts=[40000, 120000, 240000, 360000],
                restart_weights=[1, 1, 1, 1], eta_min=1e-7, rec_loss_factor=rec_f, codebook_loss_factor=cb_f, gan_start=40000)
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)           # the synthetic prior == the generator the golden run loaded
    return Stage1Trainer(model, topt, dev, beta=beta), rec_f, cb_f


def test_generator_phase_matches_the_reference_golden(golden_dir):
    from train_constants import projection
    d = np.load(os.path.join(golden_dir, "stage1_gen.npz"))
    dev = torch.device("cuda", 0)
    tr, rec_f, cb_f = _trainer(d, dev)
    imgs = torch.from_numpy(d["imgs"]).to(dev)
    rec, q, idx = tr.forward_backward(imgs, forced_idx=torch.from_numpy(d["code_idx_1"]).to(dev))
    torch.cuda.synchronize()
    agree = float((idx.cpu().numpy() == d["code_idx_1"]).mean())
    safe = d["code_margin"] > 1e-2
    assert (idx.cpu().numpy()[safe] == d["code_idx_1"][safe]).all(), "a code with a comfortable distance margin flipped"
    assert abs(rec.item() - float(d["rec_loss_1"])) <= 2e-5 * float(d["rec_loss_1"])
    assert abs(q.item() - float(d["q_loss_1"])) <= 2e-5 * float(d["q_loss_1"])
    _close(tr.last_decoded.nchw(), torch.from_numpy(d["decoded"]), 2e-5, "decoded image")
    names = [str(n) for n in d["grad_names"]]
    errs = {}
    for i, k in enumerate(names):
        full = "refmodel." + k
        base, leaf = full.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d["grad_stats"][i]
        assert want[0] > 0, k
        if k.endswith(".k.bias"):          # exactly zero in theory (softmax rows are shift-invariant): rounding noise on both sides
            assert float(g.abs().max()) <= 1e-5 and want[0] <= 1e-5
            continue
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(f"stage-1 generator phase: code agreement {agree:.4f}; gradient parity worst", [(k, f"{e:.1e}") for k, e in worst],
          "median %.1e" % np.median(list(errs.values())))
    assert max(errs.values()) <= 2e-3 and np.median(list(errs.values())) <= 1e-4
    for k in [f[len("grad__"):] for f in d.files if f.startswith("grad__")]:
        base, leaf = ("refmodel." + k).rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().cpu().reshape(d["grad__" + k].shape)
        _close(g, torch.from_numpy(d["grad__" + k]), 2e-3, "grad " + k)


def test_two_generator_steps_match_the_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, "stage1_gen.npz"))
    dev = torch.device("cuda", 0)
    tr, _, _ = _trainer(d, dev)
    imgs = torch.from_numpy(d["imgs"]).to(dev)
    sd_keys = {k for k in tr.generator_state_dict()}
    assert {str(n) for n in d["grad_names"]} <= sd_keys                  # the reference generator's parameter names
    for step in (1, 2):
        r = tr.step(imgs, forced_idx=torch.from_numpy(d[f"code_idx_{step}"]).to(dev))
        torch.cuda.synchronize()
        assert abs(r["rec_loss"].item() - float(d[f"rec_loss_{step}"])) <= (2e-5 if step == 1 else 2e-3) * float(d[f"rec_loss_{step}"])
        assert abs(r["q_loss"].item() - float(d[f"q_loss_{step}"])) <= (2e-5 if step == 1 else 2e-3) * float(d[f"q_loss_{step}"])
        assert abs(r["lr"] - float(d[f"lr_after_{step}"])) <= 1e-12
        gsd = tr.generator_state_dict()
        for k in [f[len(f"param{step}__"):] for f in d.files if f.startswith(f"param{step}__")]:
            _close(gsd[k].reshape(d[f"param{step}__" + k].shape), torch.from_numpy(d[f"param{step}__" + k]), 1e-3 if step == 1 else 5e-3, f"step {step} {k}")
    tr.current_step = 40000                                             # the adversarial phase needs a discriminator (test_stage1_adv_gpu.py)
    with pytest.raises(RuntimeError, match="discriminator"):
        tr.forward_backward(imgs)


def _fast_refresh_equals_layerwise(tr):
    """TrainEngine.enable_fast_refresh (two index planes + multiplier, one gather) leaves exactly what the layer-by-layer repack leaves."""
    eng = tr.eng
    assert eng._ridx is not None and eng._ridx.dtype == torch.int32
    assert int((eng._rmask != 0).sum()) >= tr.n_params
    g = torch.Generator(device="cpu").manual_seed(5)
    tr.flat_p.mul_(1.0 + 0.1 * torch.rand(tr.flat_p.numel(), generator=g).to(tr.flat_p.device))
    eng.refresh_weights()
    fast = {}
    for name in sorted(eng.trainable):
        pc = eng.pc.get(name)
        if pc is not None:
            fast[name + "@w"] = pc.w.clone()
            if pc.b is not None:
                fast[name + "@b"] = pc.b.clone()
        for leaf in ("weight", "bias"):
            if f"{name}.{leaf}" in eng.par:
                fast[f"{name}.{leaf}@par"] = eng.par[f"{name}.{leaf}"].clone()
    eng._ridx = None                                   # the layer-by-layer path (fresh tensors)
    eng.refresh_weights()
    n = 0
    for name in sorted(eng.trainable):
        pc = eng.pc.get(name)
        if pc is not None:
            assert torch.equal(fast[name + "@w"], pc.w), name
            n += 1
            if pc.b is not None:
                assert torch.equal(fast[name + "@b"], pc.b), name
        for leaf in ("weight", "bias"):
            if f"{name}.{leaf}" in eng.par:
                assert torch.equal(fast[f"{name}.{leaf}@par"], eng.par[f"{name}.{leaf}"]), name
    return n


def test_one_gather_repack_of_the_42M_parameter_generator(golden_dir):
    """the stage-1 generator (42.6 M parameters, beyond the 2^24 exact integers of one fp32 index plane) repacks with ONE
    gather per step; attention q projections carry their folded C^-1/2 through the multiplier plane."""
    d = np.load(os.path.join(golden_dir, "stage1_gen.npz"))
    tr, _, _ = _trainer(d, torch.device("cuda", 0))
    assert tr.n_params > (1 << 24)
    assert any(v != 1.0 for v in tr.eng.wscale.values())
    assert _fast_refresh_equals_layerwise(tr) > 40
