"""Stage-3 loss forward on the HIP kernels (forward part): model.vgg taps, contextual loss and
the loss half of train_EMSR_onestep, against the vectors emitted by the reference's own model/contextual.py +
model/VGG.py (tests/golden/cx_x8.npz, oracle/gen_golden_cx.py) and against the CPU oracle at the training size.
Tolerance: 1e-3 relative (north_star), the measured errors are ~1e-6."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_M = {}


def _model():
    if "m" not in _M:
        from gpemsr_amd.config import build_model, load_options
        opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
        _M["m"] = build_model(opt, load_prior_files=False).eval().to(torch.device("cuda", 0))
    return _M["m"]


def _rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_contextual_loss_on_features_matches_reference(golden_dir):
    from gpemsr_amd.contextual import contextual_loss
    d = np.load(os.path.join(golden_dir, "cx_x8.npz"))
    fx, fy = torch.from_numpy(d["f_x"]).cuda(), torch.from_numpy(d["f_y"]).cuda()
    loss, c = contextual_loss(fx, fy, band_width=0.5)
    assert abs(float(loss) - float(d["f_loss"])) <= 1e-5 * abs(float(d["f_loss"]))
    assert c.shape == d["f_c"].shape and _rel(c, d["f_c"]) <= 1e-5
    loss, c = contextual_loss(fx, fx.flip(0) * 0.5 + 0.2, band_width=0.1)
    assert abs(float(loss) - float(d["f_loss_bw01"])) <= 1e-5 and _rel(c, d["f_c_bw01"]) <= 1e-4
    with pytest.raises(NotImplementedError):
        contextual_loss(fx, fy, loss_type='L2')
    with pytest.raises(RuntimeError):
        contextual_loss(fx.cpu(), fy.cpu())


def test_vgg_taps_and_contextual_loss_module_match_reference(golden_dir):
    from gpemsr_amd.contextual import ContextualLoss, VGG_MEAN, VGG_STD
    d = np.load(os.path.join(golden_dir, "cx_x8.npz"))
    model = _model()
    x3, y3 = torch.from_numpy(d["i_x"]).cuda(), torch.from_numpy(d["i_y"]).cuda()
    m = torch.tensor(VGG_MEAN, device="cuda").view(1, 3, 1, 1); s = torch.tensor(VGG_STD, device="cuda").view(1, 3, 1, 1)
    taps = model.vgg((x3 - m) / s)
    assert taps._fields == ('relu1_2', 'relu2_2', 'relu3_4', 'relu4_4', 'relu5_4')
    for name in taps._fields:
        assert _rel(getattr(taps, name), d["i_x_" + name]) <= 1e-4, name
    crit = ContextualLoss(model.vgg).to("cuda")
    loss, c = crit(x3, y3)
    assert abs(float(loss) - float(d["i_loss"])) <= 1e-4 * abs(float(d["i_loss"]))
    assert _rel(c, d["i_c"]) <= 1e-3
    # the loss half of train_EMSR_onestep (train_stage3.py:349-359)
    sr, ref_img, gt = (torch.from_numpy(d[k]).cuda() for k in ("t_sr", "t_ref_img", "t_gt"))
    b, _, h, w = sr.shape
    t = ref_img.shape[1]
    sr_b = sr[:, None].expand(-1, -1, 3, -1, -1).expand(-1, t, -1, -1, -1).reshape(b * t, 3, h, w)
    ref_b = ref_img.expand(-1, -1, 3, -1, -1).reshape(b * t, 3, h, w)
    ref_loss, u = crit(sr_b, ref_b)
    assert abs(float(ref_loss) - float(d["t_ref_loss"])) <= 1e-4 * abs(float(d["t_ref_loss"])) and _rel(u, d["t_u"]) <= 1e-3


def test_contextual_loss_at_training_size_against_oracle():
    """config 5 geometry: SR 256^2 -> relu3_4 64x64 = 4096 positions, dist [n, 4096, 4096]."""
    from oracle import gpemsr_oracle as orc
    from gpemsr_amd.contextual import ContextualLoss
    model = _model()
    g = torch.Generator().manual_seed(11)
    x3 = torch.rand(2, 3, 256, 256, generator=g)
    y3 = (x3 * 0.7 + 0.3 * torch.rand(2, 3, 256, 256, generator=g)).clamp(0, 1)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("vgg.")}
    with torch.no_grad():
        want, cw, fx, _ = orc.contextual_loss_vgg(sd, "vgg", x3, y3)
    loss, c = ContextualLoss(model.vgg).to("cuda")(x3.cuda(), y3.cuda())
    assert c.shape == (2, 1, 64, 64)
    assert abs(float(loss) - float(want)) <= 1e-3 * abs(float(want)), (float(loss), float(want))
    # c gathers dist at an argmax position: a near-tie may pick another row, so compare in aggregate
    assert abs(float(c.mean()) - float(cw.mean())) <= 1e-3 * float(cw.mean())
    assert float((c.cpu() - cw).abs().max() <= 1e-3 * cw.abs().max()) or float(((c.cpu() - cw).abs() > 1e-3 * cw.abs().max()).float().mean()) < 1e-3
