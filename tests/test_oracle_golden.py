"""CPU: the oracle (oracle/gpemsr_oracle.py) against the golden vectors that
oracle/gen_golden.py captured from the UNMODIFIED reference modules
(/root/reference/GPEMSR-CREMI/GPEMSR/model/*.py imported behind shims).  This is the
pin of the oracle; the GPU parity tests then compare the HIP path with both."""
import json
import os

import numpy as np
import pytest
import torch


def _load(golden_dir, tag):
    return np.load(os.path.join(golden_dir, tag + ".npz"))


def _weights(scale):
    from gpemsr_amd.arch import param_specs
    from gpemsr_amd.config import load_options
    from gpemsr_amd.synth import synth_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt = load_options(os.path.join(root, "option", f"output_GPEMSR_x{scale}.yml"))
    kw = {k: v for k, v in opt["network"].items() if k not in ("ref_path_G", "ref_path_Indexer")}
    return synth_state_dict(param_specs(scale=scale, **kw), seed=0)


def _check(got, d, name, tol):
    if name in d.files:
        want, g = d[name], got.detach().numpy()
    else:
        stride = int(d[name + "__stride"][0])
        want, g = d[name + "__sub"], got.detach().numpy().reshape(-1)[::stride]
    err = np.abs(g.reshape(want.shape).astype(np.float64) - want).max()
    assert err <= tol * np.abs(want).max(), f"{name}: {err:.3e}"


@pytest.mark.parametrize("tag", ["x8_lr16_b1_uniform", "x16_lr16_b1_smooth"])
def test_oracle_reproduces_reference(tag, golden_dir):
    from oracle import gpemsr_oracle as orc
    d = _load(golden_dir, tag)
    scale = int(d["scale"])
    sd = _weights(scale)
    tr = {}
    torch.set_num_threads(8)
    with torch.no_grad():
        out, ref_img = orc.gpemsr_forward(sd, torch.from_numpy(d["x"]), scale=scale, trace=tr)
    assert np.array_equal(tr["code_idx"].numpy(), d["code_idx"]), "codebook indices differ from the reference"
    for name, t in (("L1_fea", tr["L1_fea"]), ("logits", tr["logits"]), ("mask_cos", tr["mask_cos"]),
                    ("L1_fused", tr["L1_fused"]), ("aligned", tr["aligned"]), ("fused", tr["fused"]),
                    ("ref_img", ref_img), ("out", out)):
        _check(t, d, name, 1e-5)
    # image space (util/util.py:139-163, 253-260)
    u8 = orc.tensor2img_u8(out[0:1])
    diff = np.abs(u8.astype(np.int32) - d["out_u8"].astype(np.int32))      # 1e-7 float noise may flip a .5 rounding
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    base = torch.nn.functional.interpolate(torch.from_numpy(d["x"])[0:1, 2], scale_factor=scale, mode="bilinear", align_corners=False)
    assert abs(orc.psnr_u8(u8, orc.tensor2img_u8(base)) - float(d["psnr_vs_base"])) < 1e-3


def test_oracle_as_written_equals_deduplicated(golden_dir):
    """Evaluating SpyNet twice (model/GPEMSR.py:99-100) changes nothing."""
    from oracle import gpemsr_oracle as orc
    d = _load(golden_dir, "x8_lr16_b1_uniform")
    sd = _weights(8)
    x = torch.from_numpy(d["x"])
    with torch.no_grad():
        a, _ = orc.gpemsr_forward(sd, x, scale=8, as_written=True)
        b, _ = orc.gpemsr_forward(sd, x, scale=8, as_written=False)
    assert torch.equal(a, b)


def test_oracle_teacher_forcing_and_fp64(golden_dir):
    """forced indices reproduce the free-running result; fp64 evaluation bounds the fp32 noise."""
    from oracle import gpemsr_oracle as orc
    d = _load(golden_dir, "x8_lr16_b1_uniform")
    sd = _weights(8)
    x = torch.from_numpy(d["x"])
    idx = torch.from_numpy(d["code_idx"]).long()
    with torch.no_grad():
        a, _ = orc.gpemsr_forward(sd, x, scale=8, forced_idx=idx)
        b, _ = orc.gpemsr_forward({k: v.double() for k, v in sd.items()}, x.double(), scale=8, forced_idx=idx)
    _check(a, d, "out", 1e-5)
    assert float((a.double() - b).abs().max() / b.abs().max()) < 1e-4      # fp32 vs fp64 evaluation of the same graph


def test_deform_conv_restatements_agree():
    """Two independent restatements of torchvision.ops.deform_conv2d (gather form in the oracle,
    grid_sample form in the import shim) agree, including far out-of-image offsets."""
    import sys
    from oracle import gpemsr_oracle as orc
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shims")
    sys.path.insert(0, shim)
    try:
        import importlib
        tv_ops = importlib.import_module("torchvision.ops")
    finally:
        sys.path.remove(shim)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 9, 11, generator=g)
    off = torch.randn(2, 144, 9, 11, generator=g) * 4
    mask = torch.rand(2, 72, 9, 11, generator=g)
    w, b = torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g)
    a = orc.deform_conv2d_v2(x, off, mask, w, b)
    c = tv_ops.deform_conv2d(x, off, w, b, 1, 1, 1, mask)
    assert float((a - c).abs().max()) < 1e-4
    for m in [m for m in list(sys.modules) if m.split(".")[0] == "torchvision"]:
        del sys.modules[m]


def test_gen_report_is_clean(golden_dir):
    rep = json.load(open(os.path.join(golden_dir, "gen_report.json")))
    assert len(rep) == 3
    for r in rep:
        assert r["idx_agree"] == 1.0
        for k in ("out", "ref_img", "L1_fea", "logits", "mask_cos", "L1_fused", "aligned", "fused"):
            assert r[k] < 1e-5, (r["case"], k, r[k])


def test_oracle_contextual_loss_matches_reference_golden(golden_dir):
    """oracle.contextual_loss / vgg19_taps / stage3_losses vs the vectors emitted by the reference's own
    model/contextual.py + model/VGG.py (oracle/gen_golden_cx.py)."""
    import numpy as np
    import torch
    from oracle import gpemsr_oracle as orc
    d = np.load(os.path.join(golden_dir, "cx_x8.npz"))
    T = lambda k: torch.from_numpy(d[k])
    loss, c = orc.contextual_loss(T("f_x"), T("f_y"), 0.5)
    assert abs(float(loss) - float(d["f_loss"])) <= 1e-6 * abs(float(d["f_loss"]))
    assert np.allclose(c.numpy(), d["f_c"], rtol=1e-6, atol=1e-7)
    loss, c = orc.contextual_loss(T("f_x"), T("f_x").flip(0) * 0.5 + 0.2, 0.1)
    assert abs(float(loss) - float(d["f_loss_bw01"])) <= 1e-6 and np.allclose(c.numpy(), d["f_c_bw01"], rtol=1e-5, atol=1e-7)
    sd = {k: v for k, v in _weights(8).items() if k.startswith("vgg.")}
    with torch.no_grad():
        loss, c, fx, _ = orc.contextual_loss_vgg(sd, "vgg", T("i_x"), T("i_y"))
        assert np.allclose(fx.numpy(), d["i_x_relu3_4"], rtol=1e-5, atol=1e-6)
        assert abs(float(loss) - float(d["i_loss"])) <= 1e-5 and np.allclose(c.numpy(), d["i_c"], rtol=1e-4, atol=1e-6)
        m = torch.tensor(orc.VGG_MEAN).view(1, 3, 1, 1); s = torch.tensor(orc.VGG_STD).view(1, 3, 1, 1)
        taps = orc.vgg19_taps(sd, "vgg", (T("i_x") - m) / s)
        for name in orc.VGG_TAP_NAMES:
            ref = d["i_x_" + name]
            assert np.abs(taps[name].numpy() - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30), name
        rec, ref_loss, u = orc.stage3_losses(sd, T("t_sr"), T("t_ref_img"), T("t_gt"))
        assert abs(float(rec) - float(d["t_rec_loss"])) <= 1e-7
        assert abs(float(ref_loss) - float(d["t_ref_loss"])) <= 1e-5 and np.allclose(u.numpy(), d["t_u"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag,step", [("15", 15), ("16f", 16)])
def test_discriminator_oracle_reproduces_the_reference_adversarial_step(tag, step, golden_dir):
    """oracle/disc_oracle.py against the UNMODIFIED reference's ``train_vqgan_onestep`` (tests/golden/stage1_adv.npz): the losses it logged
    and the gradient it left in every discriminator tensor (step 16: with the R1 penalty), from the same initial weights."""
    from oracle import disc_oracle as do
    from gpemsr_amd.discriminator import Discriminator
    from train_constants import projection
    d = _load(golden_dir, "stage1_adv")
    ic, nf, nl = [int(v) for v in d["disc_args"]]
    init = Discriminator(dict(im_channel=ic, num_filters_last=nf, n_layers=nl), init_seed=0).state_dict()      # seeded initial weights only
    disc = do.DiscOracle(init, ic, nf, nl)
    opt = [float(v) for v in d["train_opt"]]
    imgs, decoded = torch.from_numpy(d["imgs"]), torch.from_numpy(d[f"decoded_{tag}"])
    g_loss, _ = do.generator_gan_term(disc, decoded)
    assert abs(g_loss.item() - float(d[f"g_loss_{tag}"])) <= 2e-5 * abs(float(d[f"g_loss_{tag}"]))
    out = do.discriminator_losses(disc, imgs, decoded, step, r1_reg_weight=opt[7], net_d_reg_every=int(opt[8]))
    for k in ("d_loss_real", "d_loss_fake"):
        assert abs(out[k] - float(d[f"{k}_{tag}"])) <= 2e-5 * abs(float(d[f"{k}_{tag}"])), k
    if step % int(opt[8]) == 0:
        assert abs(out["r1_loss"] - float(d[f"r1_{tag}"])) <= 1e-4 * float(d[f"r1_{tag}"])
    for i, k in enumerate(str(n) for n in d["d_names"]):
        want, g = d[f"d_grad_stats_{tag}"][i], disc.grad(k).reshape(-1)
        if want[0] <= 1e-9:
            assert float(g.abs().max()) <= 1e-9, k
            continue
        err = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
        assert err <= 2e-4, (k, err)
