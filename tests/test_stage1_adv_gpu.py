"""Adversarial phase of stage-1 (VQGAN) training (R:train_stage1.py:300-312,330-372; R:model/discriminator.py:9-32) through the C ABI:
  * the kernels of csrc/stage1_adv.hip against torch (im2col / col2im of the 4x4 convolutions, LeakyReLU(0.2), InstanceNorm2d and its
    first- and second-order backward against torch.autograd's double backward in float64);
  * the discriminator engine (forward, backward to image and weights, R1 penalty's gradient of a gradient) against the float64 autograd
    oracle of the reference module (oracle/disc_oracle.py);
  * two adversarial steps (one with the R1 penalty) against vectors emitted by the UNMODIFIED reference ``train_vqgan_onestep``
    (oracle/gen_golden_stage1_adv.py -> tests/golden/stage1_adv.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda", 0)
DARGS = dict(im_channel=1, num_filters_last=64, n_layers=3)          # option/train_stage1.yml network.Discriminator


def _rel(got, want):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def _act(t_nhwc):
    from gpemsr_amd.ops import Act
    n, h, w, c = t_nhwc.shape
    return Act(t_nhwc.to(DEV, torch.float32).contiguous(), n, h, w, c, c, 0)


def _cols(x_nhwc, stride, kp):
    """[n][oh][ow][(ky*4+kx)*c + ci], zero padded to kp."""
    n, h, w, c = x_nhwc.shape
    u = x_nhwc.unfold(1, 4, stride).unfold(2, 4, stride)             # [n][oh][ow][c][ky][kx]
    col = u.permute(0, 1, 2, 4, 5, 3).reshape(n, u.shape[1], u.shape[2], 16 * c)
    return F.pad(col, (0, kp - 16 * c))


@pytest.mark.parametrize("c,stride,h,w", [(1, 2, 38, 30), (8, 2, 21, 18), (8, 1, 11, 9), (64, 2, 14, 14)])
def test_im2col4_and_its_adjoint(c, stride, h, w):
    from gpemsr_amd import ops
    g = torch.Generator().manual_seed(c * 100 + h)
    x = torch.randn(2, h, w, c, generator=g, dtype=torch.float64)
    kp = (16 * c + 31) // 32 * 32
    want = _cols(x, stride, kp)
    got = ops.im2col4(_act(x), stride, kp)
    assert torch.equal(got.torch().view(want.shape).cpu(), want.float())
    d = torch.randn(want.shape, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    (_cols(xr, stride, kp) * d).sum().backward()
    dx = ops.new_act(2, h, w, c, device=DEV)
    ops.col2im4(_act(d), h, w, c, stride, dx, accumulate=False)
    assert _rel(dx.torch().view(2, h, w, c), xr.grad) <= 1e-6
    ops.col2im4(_act(d), h, w, c, stride, dx, accumulate=True)
    assert _rel(dx.torch().view(2, h, w, c), 2 * xr.grad) <= 1e-6


def test_lrelu_slope_and_backward():
    from gpemsr_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 7, 12, generator=g)
    x[0, 0, 0, :4] = 0.0
    dy = torch.randn(2, 9, 7, 12, generator=g)
    y = ops.lrelu_slope(_act(x), 0.2)
    assert torch.equal(y.torch().view(x.shape).cpu(), F.leaky_relu(x, 0.2))
    xr = x.clone().requires_grad_(True)
    F.leaky_relu(xr, 0.2).backward(dy)
    dx = ops.lrelu_slope_bwd(_act(dy), y, 0.2)
    assert torch.equal(dx.torch().view(x.shape).cpu(), xr.grad)


@pytest.mark.parametrize("n,h,w,c", [(2, 14, 14, 256), (3, 11, 11, 64), (1, 30, 30, 128)])
def test_instnorm_first_and_second_order_against_autograd(n, h, w, c):
    """dx = IN'(x)[dy] and, for L = <g, dx>:  dL/d(dy) and dL/dx (what gpemsr_instnorm_bwd_bwd returns) against float64 autograd."""
    from gpemsr_amd import ops
    gen = torch.Generator().manual_seed(n * 1000 + c)
    x = (torch.randn(n, h, w, c, generator=gen, dtype=torch.float64) * 1.7 + 0.3)
    dy = torch.randn(n, h, w, c, generator=gen, dtype=torch.float64)
    g = torch.randn(n, h, w, c, generator=gen, dtype=torch.float64)
    nchw, nhwc = (lambda t: t.permute(0, 3, 1, 2).contiguous()), (lambda t: t.permute(0, 2, 3, 1))
    xr, dyr = nchw(x).requires_grad_(True), nchw(dy).requires_grad_(True)    # contiguous NCHW leaves: torch's CPU instance_norm backward
    y = F.instance_norm(xr, eps=1e-5)                                         # is wrong for a PERMUTED input with n == 1
    (dx,) = torch.autograd.grad(y, xr, dyr, create_graph=True)
    gx_want, gdy_want = torch.autograd.grad((dx * nchw(g)).sum(), (xr, dyr))
    y, dx, gx_want, gdy_want = nhwc(y), nhwc(dx), nhwc(gx_want), nhwc(gdy_want)
    xa = _act(x)
    ya, mr = ops.instnorm(xa)
    assert _rel(ya.torch().view(x.shape), y) <= 2e-5
    dxa = ops.instnorm_bwd(xa, mr, _act(dy))
    assert _rel(dxa.torch().view(x.shape), dx) <= 2e-5
    gx = ops.new_act(n, h, w, c, device=DEV)
    gdy = ops.instnorm_bwd_bwd(xa, mr, _act(dy), _act(g), gx, accumulate_gx=False)
    assert _rel(gdy.torch().view(x.shape), gdy_want) <= 2e-5
    assert _rel(gx.torch().view(x.shape), gx_want) <= 5e-5
    ops.instnorm_bwd_bwd(xa, mr, _act(dy), _act(g), gx, accumulate_gx=True)
    assert _rel(gx.torch().view(x.shape), 2 * gx_want) <= 5e-5


def _disc_pair(seed=3):
    from gpemsr_amd.discriminator import Discriminator
    d = Discriminator(DARGS, init_seed=seed).to(DEV)
    from oracle.disc_oracle import DiscOracle            # float64 autograd restatement of R:model/discriminator.py (pinned on CPU)
    return d, DiscOracle(d.state_dict(), DARGS["im_channel"], DARGS["num_filters_last"], DARGS["n_layers"])


def test_discriminator_state_dict_keys_and_forward():
    d, td = _disc_pair()
    assert sorted(d.state_dict()) == sorted(["model.0.weight", "model.0.bias", "model.2.weight", "model.5.weight", "model.8.weight",
                                             "model.11.weight", "model.11.bias"])
    x = torch.rand(2, 1, 128, 128, generator=torch.Generator().manual_seed(1))
    want = td(x.double())
    got = d(x.to(DEV))
    assert got.shape == want.shape == (2, 1, 8, 8)
    assert _rel(got, want) <= 5e-5
    with pytest.raises(RuntimeError, match="no CPU path"):
        d(x)


def test_discriminator_backward_to_image_and_weights():
    from gpemsr_amd import ops
    d, td = _disc_pair()
    gen = torch.Generator().manual_seed(2)
    x = torch.rand(2, 1, 96, 128, generator=gen)
    xr = x.double().requires_grad_(True)
    out = td(xr)
    seed = torch.randn(out.shape, generator=gen, dtype=torch.float64)
    (out * seed).sum().backward()
    eng = d.engine(DEV)
    o, saved = eng.forward(ops.from_nchw(x.to(DEV)), save=True)
    da = ops.new_act(o.n, o.h, o.w, o.c, device=DEV, zero=True)
    da.torch().view(-1, o.c)[:, 0] = seed.reshape(-1).float().to(DEV)
    gw = {k: torch.zeros_like(v) for k, v in d.state_dict().items()}
    dx = eng.backward(saved, da, True, gw)
    assert _rel(dx.nchw(), xr.grad) <= 1e-4
    for k, g in gw.items():
        assert _rel(g, td.p[k.replace(".", "_")].grad) <= 2e-4, k


def test_r1_penalty_gradient_of_a_gradient():
    """R:train_stage1.py:360-372 with create_graph=True, float64 autograd as the reference."""
    from gpemsr_amd import ops
    d, td = _disc_pair(seed=7)
    x = torch.rand(2, 1, 128, 128, generator=torch.Generator().manual_seed(9))
    xr = x.double().requires_grad_(True)
    pred = td(xr)
    (grad_real,) = torch.autograd.grad(pred.sum(), xr, create_graph=True)
    pen = grad_real.pow(2).view(2, -1).sum(1).mean()
    scale = 3.0
    (scale * pen).backward()
    eng = d.engine(DEV)
    gw = {k: torch.zeros_like(v) for k, v in d.state_dict().items()}
    got = eng.r1_penalty(ops.from_nchw(x.to(DEV)), scale, gw)
    assert abs(got.item() - pen.item()) <= 1e-4 * pen.item()
    for k, g in gw.items():
        want = td.p[k.replace(".", "_")].grad
        if k == "model.11.bias":                                      # the output bias never reaches the input gradient
            assert float(g.abs().max()) == 0.0 and (want is None or float(want.abs().max()) == 0.0)
            continue
        assert _rel(g, want) <= 5e-4, k


def _adv_trainer(d, dev):
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.discriminator import Discriminator
    from gpemsr_amd.train_stage1 import Stage1Trainer
    lr_g, lr_d, b1, b2, rec_f, cb_f, gan_f, r1_w, reg_every, gen_rate, beta = [float(v) for v in d["train_opt"]]
    topt = dict(lr_G=lr_g, lr_D=lr_d, beta1=b1, beta2=b2, T_period=[40000, 80000, 120000, 120000, 120000], restarts=[40000, 120000, 240000, 360000],
                restart_weights=[1, 1, 1, 1], eta_min=1e-7, rec_loss_factor=rec_f, codebook_loss_factor=cb_f, gan_start=0, gan_loss_factor=gan_f,
                r1_reg_weight=r1_w, net_d_reg_every=int(reg_every), generator_update_rate=int(gen_rate))
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)
    ic, nf, nl = [int(v) for v in d["disc_args"]]
    disc = Discriminator(dict(im_channel=ic, num_filters_last=nf, n_layers=nl), init_seed=0).to(dev)
    return Stage1Trainer(model, topt, dev, beta=beta, discriminator=disc), disc, topt


def _check_step(tr, disc, d, r, step, tag, tol, sharp_params):
    from train_constants import projection
    gnames, dnames = [str(n) for n in d["g_names"]], [str(n) for n in d["d_names"]]
    for key in ("rec_loss", "q_loss", "g_loss", "d_loss_real", "d_loss_fake"):
        want = float(d[f"{key}_{tag}"])
        assert abs(r[key].item() - want) <= tol * 1e-4 * max(abs(want), 0.05), (tag, key, r[key].item(), want)
    if step % 16 == 0:
        want = float(d[f"r1_{tag}"])
        assert abs(r["r1_loss"].item() - want) <= tol * 2e-4 * want, (r["r1_loss"].item(), want)
    else:
        assert "r1_loss" not in r
    assert abs(r["lr"] - float(d[f"lr_g_after_{tag}"])) <= 1e-12 and abs(r["lr_d"] - float(d[f"lr_d_after_{tag}"])) <= 1e-12
    errs = {}
    for i, k in enumerate(gnames):
        base, leaf = ("refmodel." + k).rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d[f"g_grad_stats_{tag}"][i]
        if k.endswith(".k.bias"):                # zero in theory (softmax rows are shift-invariant): rounding noise on both sides
            continue
        errs["G " + k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    for i, k in enumerate(dnames):
        g = tr.d_gw[k].detach().reshape(-1).double().cpu()
        want = d[f"d_grad_stats_{tag}"][i]
        if want[0] <= 1e-9:                      # the output bias: 0.5 * (-1 + 1) in d_loss, never reached by the R1 penalty
            assert float(g.abs().max()) <= 1e-6, k
            continue
        errs["D " + k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    dworst = max(v for k, v in errs.items() if k.startswith("D "))
    print(f"stage-1 adversarial step {tag}: gradient parity worst", [(k, f"{e:.1e}") for k, e in worst], "median %.1e" % np.median(list(errs.values())),
          f"discriminator worst {dworst:.1e}")
    assert max(errs.values()) <= tol * 2e-3 and np.median(list(errs.values())) <= tol * 2e-4
    gsd, dsd = tr.generator_state_dict(), disc.state_dict()
    lr_sum = {"g": 2 * float(d["train_opt"][0]), "d": 2 * float(d["train_opt"][1])}
    for f in d.files:
        for kind in ("g", "d"):
            if f.startswith(f"{kind}_grad_{tag}__"):
                k = f[len(f"{kind}_grad_{tag}__"):]
                if kind == "g":
                    base, leaf = ("refmodel." + k).rsplit(".", 1)
                    got = (tr.gw if leaf == "weight" else tr.gb)[base]
                else:
                    got = tr.d_gw[k]
                if float(np.abs(d[f]).max()) <= 1e-9:
                    assert float(got.abs().max()) <= 1e-6, f
                else:
                    assert _rel(got.reshape(d[f].shape), torch.from_numpy(d[f])) <= tol * 2e-3, f
            elif f.startswith(f"{kind}_param_{tag}__"):
                k = f[len(f"{kind}_param_{tag}__"):]
                got, want = (gsd if kind == "g" else dsd)[k].reshape(d[f].shape).cpu().double(), torch.from_numpy(d[f]).double()
                if sharp_params:
                    # first Adam step = lr * g / (|g| + eps): sign-like, so only elements whose gradient is not rounding noise compare
                    gr = torch.from_numpy(d[f"{kind}_grad_{tag}__{k}"]).double().abs()
                    solid = gr > 1e-3 * gr.max()
                    assert bool(solid.any()) or float(gr.max()) <= 1e-9, f
                    assert ((got - want).abs() * solid).max().item() <= 1e-3 * want.abs().max().item(), f
                    assert (got - want).abs().max().item() <= 1.01 * lr_sum[kind], f
                else:
                    # a second Adam step after a first, sign-like one: elements whose gradient is rounding noise may move the other way
                    # (|update| <= lr per step), everything else agrees
                    e = (got - want).abs()
                    assert e.max().item() <= 2 * lr_sum[kind] and e.mean().item() <= 0.1 * lr_sum[kind], (f, e.max().item(), e.mean().item())


def test_two_adversarial_steps_match_the_reference_golden(golden_dir):
    """Steps 15 and 16 of the reference run, consecutively (generator step with the GAN term, discriminator step; step 16 adds R1)."""
    d = np.load(os.path.join(golden_dir, "stage1_adv.npz"))
    tr, disc, topt = _adv_trainer(d, DEV)
    imgs = torch.from_numpy(d["imgs"]).to(DEV)
    assert [str(n) for n in d["d_names"]] == [k for k, _ in disc.named_parameters()]
    for step, tol, sharp in ((15, 1.0, True), (16, 25.0, False)):      # step 16 starts from step 15's Adam update: looser bars
        r = tr.step(imgs, forced_idx=torch.from_numpy(d[f"code_idx_{step}"]).to(DEV), current_step=step)
        torch.cuda.synchronize()
        _check_step(tr, disc, d, r, step, str(step), tol, sharp)
    assert tr.step_count == 2 and tr.d_steps == 2 and tr.current_step == 16
    st = tr.state_dict()
    assert st["current_step"] == 16 and st["disc"]["steps"] == 2
    tr2, disc2, _ = _adv_trainer(d, DEV)                                  # resume: the state travels
    tr2.model.load_state_dict(tr.model.state_dict())
    disc2.load_state_dict(disc.state_dict())
    tr2.load_state_dict(st)
    assert tr2.current_step == 16 and tr2.d_steps == 2 and torch.equal(tr2.d_flat_v, tr.d_flat_v) and tr2.lr_d == tr.lr_d
    ra, rb = tr.step(imgs), tr2.step(imgs)
    assert torch.equal(tr.d_flat_p, tr2.d_flat_p) and ra["d_loss_fake"].item() == rb["d_loss_fake"].item() and ra["rec_loss"].item() == rb["rec_loss"].item()
    e = (tr.flat_p - tr2.flat_p).abs()                                  # the generator's weight-gradient kernels accumulate with float atomics
    assert e.mean().item() <= 1e-7 and e.max().item() <= 2 * tr.lr


def test_r1_step_from_fresh_weights_matches_the_reference_golden(golden_dir):
    """Step 16 (16 % net_d_reg_every == 0: the R1 penalty, R:train_stage1.py:339-345) from the initial weights: sharp bars."""
    d = np.load(os.path.join(golden_dir, "stage1_adv.npz"))
    tr, disc, topt = _adv_trainer(d, DEV)
    imgs = torch.from_numpy(d["imgs"]).to(DEV)
    r = tr.step(imgs, forced_idx=torch.from_numpy(d["code_idx_16f"]).to(DEV), current_step=16)
    torch.cuda.synchronize()
    _check_step(tr, disc, d, r, 16, "16f", 1.0, True)


def test_generator_update_rate_skips_the_generator_step(golden_dir):
    d = np.load(os.path.join(golden_dir, "stage1_adv.npz"))
    tr, disc, topt = _adv_trainer(d, DEV)
    tr.opt["generator_update_rate"] = 2
    imgs = torch.from_numpy(d["imgs"]).to(DEV)
    before_g, before_d = tr.flat_p.clone(), tr.d_flat_p.clone()
    r = tr.step(imgs, current_step=15)                                  # 15 % 2 != 0: forward only for G (R:train_stage1.py:327-328), D steps
    assert torch.equal(tr.flat_p, before_g) and not torch.equal(tr.d_flat_p, before_d)
    assert tr.step_count == 0 and tr.d_steps == 1 and r["g_loss"] is None
    tr.step(imgs, current_step=16)
    assert tr.step_count == 1 and not torch.equal(tr.flat_p, before_g)
