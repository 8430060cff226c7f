"""Stage-2 (indexer) training step (train_stage2.py:351-366) on the HIP kernels, through the C ABI: the three kernels it adds
(GroupNorm+ReLU backward, softmax backward, cross-entropy) against torch autograd, the recorded attention block and VQGAN
ResidualBlock against autograd of the CPU oracle, and two whole steps against vectors from the UNMODIFIED reference
(oracle/gen_golden_stage2.py -> tests/golden/stage2_x8.npz)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_train_gpu import _close, _dev, _rand, _to_act, _zeros_like_act      # noqa: E402


@pytest.mark.parametrize("cfg", [(2, 64, 8, 8, True, None), (3, 128, 6, 10, False, None), (1, 512, 16, 16, True, 520), (2, 256, 33, 7, True, None)])
def test_groupnorm_bwd(cfg):
    from gpemsr_amd import ops
    n, c, h, w, relu, ld = cfg
    dev = _dev()
    x = _rand(n, c, h, w, seed=1, scale=2.0).requires_grad_(True)
    ga = (_rand(c, seed=2) + 1.5).requires_grad_(True)
    be = _rand(c, seed=3, scale=0.5).requires_grad_(True)
    y = F.group_norm(x, 32, ga, be, 1e-6)
    y = F.relu(y) if relu else y
    gy = _rand(n, c, h, w, seed=4)
    y.backward(gy)
    xa = _to_act(x.detach(), dev, ld, 0)
    lib = ops._abi.load()
    hw = h * w
    parts = max(1, min(64, hw // 64))
    ws = torch.empty(n * parts * c * 2 + n * c * 2 + n * 64, device=dev)
    mr = torch.empty(n * 32 * 2, device=dev)
    gad, bed = ga.detach().to(dev), be.detach().to(dev)
    ops._abi.check(lib.gpemsr_groupnorm_stats(xa.ptr, n, hw, c, xa.ld, 32, 1e-6, ws.data_ptr(), parts, mr.data_ptr(), ops._stream()), "stats")
    dx = _zeros_like_act(xa)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    gya = _to_act(gy, dev)
    ops._abi.check(lib.gpemsr_groupnorm_bwd(xa.ptr, xa.ld, gya.ptr, gya.ld, n, hw, c, 32, mr.data_ptr(), gad.data_ptr(), bed.data_ptr(), int(relu),
                                            ws.data_ptr(), ws.numel(), dx.ptr, dx.ld, dg.data_ptr(), db.data_ptr(), ops._stream()), "gn_bwd")
    _close(dx.nchw(), x.grad, 5e-5, "gn dx"); _close(dg, ga.grad, 5e-5, "gn dgamma"); _close(db, be.grad, 5e-5, "gn dbeta")


def test_softmax_bwd_and_cross_entropy():
    from gpemsr_amd import ops
    dev = _dev()
    lib = ops._abi.load()
    for rows, cols in ((64, 256), (20, 4096), (7, 5000)):
        s = _rand(rows, cols, seed=5, scale=3.0).requires_grad_(True)
        p = F.softmax(s, dim=1)
        dp = _rand(rows, cols, seed=6)
        p.backward(dp)
        pd, dd = p.detach().to(dev).contiguous(), dp.to(dev).contiguous()
        ops._abi.check(lib.gpemsr_softmax_bwd_rows(pd.data_ptr(), dd.data_ptr(), rows, cols, ops._stream()), "softmax_bwd")
        _close(dd, s.grad, 2e-5, "softmax bwd")
    rows, cols = 300, 1024
    x = _rand(rows, cols, seed=7, scale=4.0).requires_grad_(True)
    t = torch.randint(0, cols, (rows,), generator=torch.Generator().manual_seed(8))
    loss = F.cross_entropy(x, t)
    (loss * 0.5).backward()
    xd = x.detach().to(dev).contiguous()
    rl, lo, dl = torch.empty(rows, device=dev), torch.empty(1, device=dev), torch.empty(rows, cols, device=dev)
    ops._abi.check(lib.gpemsr_cross_entropy(xd.data_ptr(), t.to(torch.int32).to(dev).data_ptr(), rows, cols, 0.5, rl.data_ptr(), lo.data_ptr(),
                                            dl.data_ptr(), ops._stream()), "cross_entropy")
    _close(lo[0], loss.detach(), 1e-6, "CE value"); _close(dl, x.grad, 2e-5, "CE grad")


_TR = {}


def _trainer():
    if "t" not in _TR:
        from gen_golden_stage2 import TRAIN_OPT
        from gpemsr_amd.config import build_model, load_options
        from gpemsr_amd.train_stage2 import Stage2Trainer
        opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
        _TR["t"] = Stage2Trainer(build_model(opt, load_prior_files=False).to(_dev()), TRAIN_OPT, _dev())
    return _TR["t"]


@pytest.mark.parametrize("layer", ["refmodel.indexer.feat_extract.8", "refmodel.indexer.feat_extract.3", "refmodel.indexer.output_layer.0"])
def test_recorded_vq_layer_backward(layer):
    """NonLocalBlock (feat_extract.8), ResidualBlock with channel_up (feat_extract.3: 64 -> 128) and plain ResidualBlock: forward,
    dX and every parameter gradient against torch autograd of the oracle's block."""
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    tr = _trainer()
    eng, dev = tr.eng, _dev()
    keys = [k for k in eng.sd if k.startswith(layer + ".")]
    sd = {k: eng.sd[k].detach().cpu().clone().requires_grad_(True) for k in keys}
    cin = sd[layer + ".gn.weight"].shape[0] if (layer + ".gn.weight") in sd else sd[layer + ".block.0.weight"].shape[1]
    x = _rand(2, cin, 8, 8, seed=11).requires_grad_(True)
    y = orc._vq_layer(sd, layer, x)
    gy = _rand(*y.shape, seed=12)
    y.backward(gy)
    xa = _to_act(x.detach(), dev).mark_grad()
    tr.flat_g.zero_()
    eng.tape = []
    out = eng.vq_layer_train(xa, layer)
    _close(out.nchw(), y.detach(), 3e-5, layer + " forward")
    ops.axpy(_to_act(gy, dev), out.grad())
    for fn in reversed(eng.tape):
        fn()
    eng.tape = None
    torch.cuda.synchronize()
    _close(xa.grad().nchw(), x.grad, 1e-4, layer + " dX")
    for k in keys:
        base, leaf = k.rsplit(".", 1)
        got = (tr.gw if leaf == "weight" else tr.gb)[base]
        if k.endswith(".k.bias"):          # exactly zero in theory (softmax rows are shift-invariant): rounding noise on both sides
            assert float(got.abs().max()) <= 1e-5 and float(sd[k].grad.abs().max()) <= 1e-5
            continue
        _close(got.reshape(sd[k].shape), sd[k].grad, 1e-4, k)
    tr.flat_g.zero_()


def test_two_stage2_steps_match_reference_golden(golden_dir):
    from gen_golden_stage2 import FULL, TRAIN_OPT
    from train_constants import projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train_stage2 import Stage2Trainer
    d = np.load(os.path.join(golden_dir, "stage2_x8.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    tr = Stage2Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    want_idx = torch.from_numpy(d["target_idx"]).to(dev)
    loss, target = tr.forward_backward(LR, GT)                      # free-running targets: Encoder + nearest code on the GPU
    torch.cuda.synchronize()
    agree = float((tr.last_target == want_idx).float().mean())
    print("encoder code-index agreement:", agree, "(reference top-2 distance margin %.2e)" % float(d["min_distance_margin"]))
    assert agree == 1.0
    assert abs(loss.item() - float(d["loss_1"])) <= 1e-5 * float(d["loss_1"])
    lg = tr.last_logits.torch().reshape(-1, 1024)[::4].cpu().numpy()
    assert np.abs(lg - d["logits_1_every4"]).max() <= 1e-4 * np.abs(d["logits_1_every4"]).max()
    names = [str(n) for n in d["grad_names"]]
    errs = {}
    for i, k in enumerate(names):
        base, leaf = ("refmodel." + k).rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d["grad_stats"][i]
        if k.endswith(".k.bias"):      # zero in exact arithmetic (softmax is shift-invariant along its rows): noise on both sides
            assert g.norm().item() <= 1e-5 and want[0] <= 1e-5
            continue
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    print("stage-2 gradient parity, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:4], "median %.1e" % np.median(list(errs.values())))
    assert max(errs.values()) <= 2e-2 and np.median(list(errs.values())) <= 1e-3       # ReLU kinks, as in stage 3 (DESIGN_HISTORY.md §3.6)
    for k in FULL:
        base, leaf = ("refmodel." + k).rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().cpu().reshape(d["grad__" + k].shape)
        _close(g, torch.from_numpy(d["grad__" + k]), 2e-2, "grad " + k)
    # two optimizer steps from a fresh model
    tr2 = Stage2Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    o1 = tr2.step(LR, GT, want_idx)
    assert abs(o1["lr"] - float(d["lr_after_1"])) <= 1e-12
    sdm = tr2.model.state_dict()
    for k in FULL:
        if k.endswith(".k.bias"):          # Adam turns the zero-mean rounding noise of this gradient into +-lr steps in every
            continue                       # implementation (the reference included); the bias cannot change the output
        want = torch.from_numpy(d["param1__" + k])
        got = sdm["refmodel." + k].detach().cpu().reshape(want.shape)
        frac_bad = ((got - want).abs() > 1e-6 + 1e-4 * want.abs()).float().mean().item()
        assert frac_bad <= 0.02, f"param after step 1 {k}: {frac_bad:.3f} of the elements differ"
    o2 = tr2.step(LR, GT, want_idx)
    torch.cuda.synchronize()
    print("step-2 loss", o2["loss"].item(), float(d["loss_2"]))
    assert abs(o2["loss"].item() - float(d["loss_2"])) <= 5e-3 * float(d["loss_2"])


def test_x16_stage2_step_matches_reference_golden(golden_dir):
    """Indexer16 (no down-sampling stage, model/indexer.py:6-55) through lrGenerator16: targets, loss and the gradients of all
    110 trainable tensors against the reference's step."""
    from gen_golden_stage2 import TRAIN_OPT
    from train_constants import projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train_stage2 import Stage2Trainer
    d = np.load(os.path.join(golden_dir, "stage2_x16.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x16.yml"))
    tr = Stage2Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    loss, _ = tr.forward_backward(LR, GT)
    torch.cuda.synchronize()
    assert float((tr.last_target == torch.from_numpy(d["target_idx"]).to(dev)).float().mean()) == 1.0
    assert abs(loss.item() - float(d["loss_1"])) <= 1e-5 * float(d["loss_1"])
    errs = {}
    for i, k in enumerate([str(n) for n in d["grad_names"]]):
        base, leaf = ("refmodel." + k).rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d["grad_stats"][i]
        if k.endswith(".k.bias"):
            assert g.norm().item() <= 1e-5 and want[0] <= 1e-5
            continue
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    print("x16 stage-2 gradient parity, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:3], "median %.1e" % np.median(list(errs.values())))
    assert max(errs.values()) <= 2e-2 and np.median(list(errs.values())) <= 1e-3


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


def test_one_gather_repack_of_the_indexer():
    from gen_golden_stage2 import TRAIN_OPT
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train_stage2 import Stage2Trainer
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    tr = Stage2Trainer(build_model(opt, load_prior_files=False).to(_dev()), TRAIN_OPT, _dev())
    assert _fast_refresh_equals_layerwise(tr) >= 30
