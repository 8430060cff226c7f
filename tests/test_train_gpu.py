"""Stage-3 training step (train_stage3.py:343-366) on the HIP kernels, through the C ABI.

Three layers of evidence:
  1. known-answer tests of every backward entry point against torch CPU autograd of the matching functional op
     (tight tolerance: these pin the kernels);
  2. recorded-convolution tests: the data/weight/bias gradients of representative layers of the engine (stride 1 / 2,
     transposed, PixelShuffle, multi-source with a padded source, 1-channel ends, un-fused residual and mask multiply)
     against torch autograd with the same weights;
  3. the end-to-end golden: losses, gradients of all 212 trainable tensors and the parameters after two Adam steps
     against vectors emitted by the UNMODIFIED reference (oracle/gen_golden_train.py -> tests/golden/train_x8.npz).

Tolerance of (3): the code indices and SpyNet flows of the frozen sub-networks are teacher-forced.  What remains is the
conditioning of the gradient itself: LeakyReLU/ReLU kinks, max-pool arg-maxes and the floor() of the deformable sampling
make it piecewise -- perturbing the weights by 1e-7 (relative) moves some POD-side gradient tensors by 2e-3 in the CPU
oracle itself (DESIGN_HISTORY.md §3.6), and ONE flipped sign in a 4x4x64 map of the ThreeDA attention pyramid moves that
branch's tensors by 1/sqrt(2048) = 2e-2.  So the reconstruction trunk and upsampler (large maps, no amplification) are held
to 1e-3, every other tensor to 6e-2 with the median below 3e-3 and the 90th percentile below 1.5e-2, and the kernels are
pinned by (1) and (2) at 1e-5 .. 5e-5."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _to_act(x_nchw, dev, ld=None, off=0):
    from gpemsr_amd import ops
    n, c, h, w = x_nchw.shape
    ld = c if ld is None else ld
    buf = torch.full((n, h, w, ld), 7.0)
    buf[..., off:off + c] = x_nchw.permute(0, 2, 3, 1)
    return ops.Act(buf.to(dev).contiguous(), n, h, w, c, ld, off)


def _zeros_like_act(a, ld=None, off=0):
    from gpemsr_amd import ops
    ld = a.c if ld is None else ld
    return ops.Act(torch.zeros(a.n * a.h * a.w * ld, device=a.buf.device), a.n, a.h, a.w, a.c, ld, off)


def _close(got, want, tol=2e-5, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-6)
    assert err <= tol * ref + 1e-7, f"{what}: max err {err:.3e} vs ref max {ref:.3e}"


# ------------------------------------------------------------------------------------------------ (1) kernels
WGRAD_CASES = [
    # n, h, w, cin, cout, k, stride, x_ld, x_off, dz_ld
    (2, 16, 16, 64, 64, 3, 1, None, 0, None),
    (1, 20, 36, 34, 64, 3, 1, 48, 0, None),          # logical 34 channels of a 48-wide buffer (POD offset_conv1)
    (2, 32, 32, 2, 16, 3, 4, None, 0, 48),           # flowdsconv0: stride 4, output is a slice of the 48-wide buffer
    (2, 16, 16, 16, 16, 3, 2, 48, 16, 48),
    (1, 16, 16, 64, 216, 3, 1, None, 0, None),
    (2, 8, 8, 320, 64, 1, 1, None, 0, None),
    (1, 64, 64, 64, 1, 3, 1, None, 0, None),         # conv_last
    (1, 16, 16, 1, 64, 3, 1, None, 0, None),         # conv_first
    (1, 18, 18, 128, 128, 3, 2, None, 0, None),
    (3, 7, 45, 64, 64, 3, 1, None, 0, None),         # ragged tiles
    (1, 16, 16, 576, 64, 1, 1, None, 0, None),       # DCN contraction
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_wgrad(case):
    from gpemsr_amd import ops
    n, h, w, cin, cout, k, stride, x_ld, x_off, dz_ld = case
    dev = _dev()
    p = k // 2
    oh, ow = (h + 2 * p - k) // stride + 1, (w + 2 * p - k) // stride + 1
    x = _rand(n, cin, h, w, seed=1)
    dz = _rand(n, cout, oh, ow, seed=2)
    wt = torch.zeros(cout, cin, k, k, requires_grad=True)
    F.conv2d(x, wt, None, stride, p).backward(dz)
    xa = _to_act(x, dev, x_ld, x_off)
    dza = _to_act(dz, dev, dz_ld, 0)
    cin_total, cin_off = cin + 5, 3
    prior = _rand(cout, cin_total, k, k, seed=3)
    dw = prior.clone().to(dev)
    ops.conv2d_wgrad(xa, dza, k, stride, dw, cin_total, cin_off)
    want = prior.clone()
    want[:, cin_off:cin_off + cin] += wt.grad
    _close(dw, want, 3e-5, "wgrad")                 # accumulates, leaves the other channels alone


def test_conv2d_wgrad_transposed_roles():
    """ConvTranspose2d(k3,s2,p1,op1) weight gradient = the stride-2 wgrad with x := dOut, dz := input."""
    from gpemsr_amd import ops
    dev = _dev()
    x = _rand(2, 64, 8, 12, seed=4)
    wt = (_rand(64, 32, 3, 3, seed=5) * 0.1).requires_grad_(True)
    go = _rand(2, 32, 16, 24, seed=6)
    F.conv_transpose2d(x, wt, None, stride=2, padding=1, output_padding=1).backward(go)
    dw = torch.zeros(64, 32, 3, 3, device=dev)
    ops.conv2d_wgrad(_to_act(go, dev), _to_act(x, dev), 3, 2, dw, 32, 0)
    _close(dw, wt.grad, 3e-5, "convT wgrad")


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("ps", [False, True])
def test_act_bwd(act, ps):
    from gpemsr_amd import ops
    dev = _dev()
    z = _rand(2, 16, 6, 10, seed=7, scale=2.0).requires_grad_(True)
    f = {0: lambda t: t, 1: F.relu, 2: lambda t: F.leaky_relu(t, 0.1), 3: torch.sigmoid,
         4: lambda t: torch.sigmoid(F.leaky_relu(t, 0.1))}[act]
    y = f(F.pixel_shuffle(z, 2) if ps else z)
    gy = _rand(*y.shape, seed=8)
    y.backward(gy)
    dz = ops.act_bwd(_to_act(gy, dev), _to_act(y.detach(), dev), 2, 6, 10, 16, act, ps)
    _close(dz.nchw(), z.grad, 1e-5, "act_bwd")


def test_bias_grad_axpy_mulpix():
    from gpemsr_amd import ops
    dev = _dev()
    dz = _rand(3, 216, 9, 11, seed=9)
    db = torch.ones(216, device=dev)
    ops.bias_grad(_to_act(dz, dev, 224, 4), db)
    _close(db, 1.0 + dz.sum(dim=(0, 2, 3)), 2e-5, "bias_grad")
    a, b = _rand(2, 20, 5, 7, seed=10), _rand(2, 20, 5, 7, seed=11)
    bb = _to_act(b, dev, 24, 2)
    ops.axpy(_to_act(a, dev), bb, 0.5)
    _close(bb.nchw(), b + 0.5 * a, 1e-6, "axpy")
    x = _rand(2, 64, 6, 8, seed=12).requires_grad_(True)
    m = torch.rand(2, 1, 6, 8, generator=torch.Generator().manual_seed(13)).requires_grad_(True)
    gy = _rand(2, 64, 6, 8, seed=14)
    (x * m).backward(gy)
    xa, ma = _to_act(x.detach(), dev), _to_act(m.detach(), dev)
    _close(ops.mul_pix(xa, ma).nchw(), (x * m).detach(), 1e-6, "mul_pix")
    dx, dm = _zeros_like_act(xa), _zeros_like_act(ma)
    ops.mul_pix_bwd(_to_act(gy, dev), xa, ma, dx, dm)
    _close(dx.nchw(), x.grad, 1e-6, "mul_pix dx")
    _close(dm.nchw(), m.grad, 1e-5, "mul_pix dm")


@pytest.mark.parametrize("cfg", [(8, 8, 16, 16, False), (6, 10, 24, 40, False), (8, 8, 64, 64, False), (16, 16, 8, 8, False),
                                 (5, 7, 10, 14, True), (9, 6, 20, 17, False)])
def test_bilinear_bwd(cfg):
    from gpemsr_amd import ops
    h, w, oh, ow, align = cfg
    dev = _dev()
    x = _rand(2, 3, h, w, seed=15).requires_grad_(True)
    y = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=align) * 2.0
    gy = _rand(*y.shape, seed=16)
    y.backward(gy)
    dx = _zeros_like_act(_to_act(x.detach(), dev), 4, 1)
    ops.bilinear_bwd(_to_act(gy, dev), dx, align, 2.0)
    _close(dx.nchw(), x.grad, 2e-5, "bilinear_bwd")


def test_dcn_columns_bwd():
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    dev = _dev()
    B, C, H, W = 2, 64, 12, 10
    x = _rand(B, C, H, W, seed=17).requires_grad_(True)
    om = (_rand(B, 216, H, W, seed=18) * 2.5).requires_grad_(True)       # offsets reach outside the image
    wt = (_rand(64, 64, 3, 3, seed=19) * 0.05)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    y = orc.deform_conv2d_v2(x, torch.cat((o1, o2), 1), torch.sigmoid(m), wt, torch.zeros(64))
    gy = _rand(*y.shape, seed=20)
    y.backward(gy)
    # HIP: col = dcn_columns(x, om); y = col . W1  ->  dcol = gy . W1^T (torch, plumbing of the test), then the kernel
    w1 = wt.permute(0, 2, 3, 1).reshape(64, 9 * 64)                      # [cout][tap][cin]
    dcol = torch.einsum("bohw,ok->bhwk", gy, w1).contiguous()
    xa, oma = _to_act(x.detach(), dev), _to_act(om.detach(), dev)
    col = ops.dcn_columns(xa, oma, 8)
    ref_col_y = torch.einsum("bhwk,ok->bohw", col.torch().cpu(), w1)
    _close(ref_col_y, y.detach(), 2e-5, "dcn forward through columns")
    dx, dom = _zeros_like_act(xa), _zeros_like_act(oma)
    ops.dcn_columns_bwd(xa, oma, 8, ops.from_nhwc(dcol.to(dev)), dx, dom)
    _close(dx.nchw(), x.grad, 3e-5, "dcn dx")
    _close(dom.nchw(), om.grad, 3e-5, "dcn d(offset, mask)")
    # the default scatter is the fixed-point one (gpemsr_dcn_columns_bwd_det): bit-stable run to run, also where most contributions leave
    # the tile's LDS window (offsets up to +-15 pixels) and into a strided gradient buffer; the float-atomic form agrees to rounding
    om_far = _to_act((om.detach() * 6.0), dev)
    firsts = None
    for rep in range(4):
        dxs = ops.Act(torch.zeros(B, H, W, 96, device=dev), B, H, W, C, 96, 16)
        doms = _zeros_like_act(om_far)
        ops.dcn_columns_bwd(xa, om_far, 8, ops.from_nhwc(dcol.to(dev)), dxs, doms)
        torch.cuda.synchronize()
        assert float(dxs.buf[..., :16].abs().max()) == 0.0 and float(dxs.buf[..., 80:].abs().max()) == 0.0       # nothing outside the channel slice
        cur = (dxs.nchw().clone(), doms.nchw().clone())
        if firsts is None:
            firsts = cur
        assert torch.equal(cur[0], firsts[0]) and torch.equal(cur[1], firsts[1]), "fixed-point scatter is not bit-stable"
    dxf, domf = _zeros_like_act(xa), _zeros_like_act(om_far)
    ops.dcn_columns_bwd(xa, om_far, 8, ops.from_nhwc(dcol.to(dev)), dxf, domf, deterministic=False)
    _close(firsts[0].cpu(), dxf.nchw().cpu(), 2e-6, "fixed-point vs float-atomic dx")
    _close(firsts[1].cpu(), domf.nchw().cpu(), 1e-6, "d(offset, mask) of the two forms")


def test_threeda_pieces_bwd():
    from gpemsr_amd import ops
    dev = _dev()
    b, t, c, h, w = 2, 5, 64, 6, 7
    al = _rand(b * t, c, h, w, seed=21).requires_grad_(True)
    emb = _rand(b * t, c, h, w, seed=22, scale=0.3).requires_grad_(True)
    er = _rand(b, c, h, w, seed=23, scale=0.3).requires_grad_(True)
    corr = torch.sigmoid((emb.view(b, t, c, h, w) * er[:, None]).sum(2, keepdim=True))
    af = (al.view(b, t, c, h, w) * corr).reshape(b, t * c, h, w)
    g = _rand(*af.shape, seed=24)
    af.backward(g)
    ala, ea, ra = _to_act(al.detach(), dev), _to_act(emb.detach(), dev), _to_act(er.detach(), dev)
    afa = ops.temporal_gate(ala, ea, ra, b, t)
    _close(afa.nchw(), af.detach(), 1e-5, "temporal_gate")
    d1, d2, d3 = _zeros_like_act(ala), _zeros_like_act(ea), _zeros_like_act(ra)
    ops.temporal_gate_bwd(ala, ea, ra, _to_act(g, dev), b, t, d1, d2, d3)
    _close(d1.nchw(), al.grad, 1e-5, "d aligned"); _close(d2.nchw(), emb.grad, 2e-5, "d emb"); _close(d3.nchw(), er.grad, 2e-5, "d emb_ref")

    # Conv3d(t,t,1) over the frame axis + LeakyReLU
    x = _rand(b, t * c, h, w, seed=25).requires_grad_(True)
    M = _rand(t, t, seed=26).requires_grad_(True)
    bias = _rand(t, seed=27).requires_grad_(True)
    y = F.leaky_relu(torch.einsum("ik,bkchw->bichw", M, x.view(b, t, c, h, w)) + bias.view(1, t, 1, 1, 1), 0.1).reshape(b, t * c, h, w)
    gy = _rand(*y.shape, seed=28)
    y.backward(gy)
    xa = _to_act(x.detach(), dev)
    Md, bd = M.detach().to(dev).contiguous(), bias.detach().to(dev).contiguous()
    ya = ops.frame_mix_lrelu(xa, t, Md, bd)
    _close(ya.nchw(), y.detach(), 1e-5, "frame_mix")
    dxa, dM, dB = _zeros_like_act(xa), torch.zeros(t, t, device=dev), torch.zeros(t, device=dev)
    ops.frame_mix_lrelu_bwd(xa, ya, _to_act(gy, dev), t, Md, dxa, dM, dB)
    _close(dxa.nchw(), x.grad, 1e-5, "frame_mix dx"); _close(dM, M.grad, 3e-5, "frame_mix dM"); _close(dB, bias.grad, 3e-5, "frame_mix db")

    # MaxPool2d(3,2,1) | AvgPool2d(3,2,1)
    for hh, ww in ((8, 8), (7, 9)):
        x = _rand(2, 16, hh, ww, seed=29).requires_grad_(True)
        y = torch.cat([F.max_pool2d(x, 3, 2, 1), F.avg_pool2d(x, 3, 2, 1)], 1)
        gy = _rand(*y.shape, seed=30)
        y.backward(gy)
        xa = _to_act(x.detach(), dev)
        dxa = _zeros_like_act(xa)
        ops.pool3s2_maxavg_bwd(xa, _to_act(gy, dev), dxa)
        _close(dxa.nchw(), x.grad, 1e-5, "pool3s2 bwd")

    # out = feat * sigmoid(attn) * 2 + add + f2 + f3
    ts = [_rand(2, 64, 5, 6, seed=31 + i).requires_grad_(True) for i in range(5)]
    out = ts[0] * torch.sigmoid(ts[1]) * 2 + ts[2] + ts[3] + ts[4]
    go = _rand(*out.shape, seed=40)
    out.backward(go)
    acts = [_to_act(v.detach(), dev) for v in ts]
    gs = [_zeros_like_act(a) for a in acts]
    ops.threeda_combine_bwd(acts[0], acts[1], _to_act(go, dev), *gs)
    for gi, v in zip(gs, ts):
        _close(gi.nchw(), v.grad, 1e-5, "threeda_combine bwd")


def test_maxpool2_scatter_l1_gray_bwd():
    from gpemsr_amd import ops
    dev = _dev()
    x = F.relu(_rand(2, 8, 10, 12, seed=41)).requires_grad_(True)       # ties at 0 as after a ReLU
    y = F.max_pool2d(x, 2, 2)
    gy = _rand(*y.shape, seed=42)
    y.backward(gy)
    xa = _to_act(x.detach(), dev)
    dxa = _zeros_like_act(xa)
    ops.maxpool2_bwd(xa, _to_act(gy, dev), dxa)
    keep = (x.detach() > 0)
    _close(dxa.nchw().cpu() * keep, x.grad * keep, 1e-6, "maxpool2 bwd (positions a ReLU keeps)")
    _close(dxa.nchw().cpu().sum(), x.grad.sum(), 1e-5, "maxpool2 bwd mass")

    src = _rand(4, 8, 3, 4, seed=43)
    idx = torch.tensor([2, 0, 2, 3, 2, 1, 0], dtype=torch.int32)
    gd = _rand(7, 8, 3, 4, seed=44)
    want = torch.zeros_like(src).index_add_(0, idx.long(), gd)
    tgt = _zeros_like_act(_to_act(src, dev))
    ops.scatter_add_images(_to_act(gd, dev), idx.to(dev), tgt)
    _close(tgt.nchw(), want, 1e-6, "scatter_add_images")

    sr = _rand(2, 1, 16, 16, seed=45).requires_grad_(True)
    gt = _rand(2, 1, 16, 16, seed=46)
    loss = torch.nn.L1Loss()(gt, sr) * 0.7
    loss.backward()
    dsr = torch.zeros(2, 1, 16, 16, device=dev)
    got = ops.l1_loss(sr.detach().to(dev).contiguous(), gt.to(dev).contiguous(), 0.7, dsr)
    _close(got[0] * 0.7, loss.detach(), 1e-6, "l1 value"); _close(dsr, sr.grad, 1e-6, "l1 grad")

    from gpemsr_amd.train import VGG_MEAN, VGG_STD
    x1 = torch.rand(2, 1, 6, 6, generator=torch.Generator().manual_seed(47)).requires_grad_(True)
    m, s = torch.tensor(VGG_MEAN).view(1, 3, 1, 1), torch.tensor(VGG_STD).view(1, 3, 1, 1)
    y3 = (x1.expand(-1, 3, -1, -1) - m) / s
    g3 = _rand(2, 3, 6, 6, seed=48)
    y3.backward(g3)
    xa = _to_act(x1.detach(), dev)
    _close(ops.gray_normalize3(xa, VGG_MEAN, VGG_STD).nchw(), y3.detach(), 1e-6, "gray_normalize3")
    dxa = _zeros_like_act(xa)
    ops.gray_normalize3_bwd(_to_act(g3, dev), VGG_STD, dxa)
    _close(dxa.nchw(), x1.grad, 1e-6, "gray_normalize3 bwd")


def test_adam_matches_torch():
    from gpemsr_amd import ops
    dev = _dev()
    p0 = _rand(1000, seed=49)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=4e-4, betas=(0.9, 0.99))
    p, m, v = p0.clone().to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for step in range(1, 4):
        g = _rand(1000, seed=50 + step, scale=1e-3)
        g[::7] = 0.0
        ref.grad = g.clone()
        opt.step()
        ops.adam_step(p, g.to(dev), m, v, 4e-4, 0.9, 0.99, 1e-8, 0.0, step)
        _close(p, ref.detach(), 1e-6, f"adam step {step}")


# ------------------------------------------------------------------------------------------------ trainer fixtures
_TR = {}


def _trainer():
    if "t" not in _TR:
        from train_constants import TRAIN_OPT
        from gpemsr_amd.config import build_model, load_options
        from gpemsr_amd.train import Stage3Trainer
        opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
        model = build_model(opt, load_prior_files=False).to(_dev())
        _TR["t"] = Stage3Trainer(model, TRAIN_OPT, _dev())
    return _TR["t"]


def test_contextual_backward_matches_autograd():
    """d CX / d features against torch autograd of the oracle's contextual_loss (x repeated t times)."""
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    tr = _trainer()
    dev = _dev()
    b, t, c, h, w = 2, 3, 64, 8, 8
    fx = F.relu(_rand(b, c, h, w, seed=60)).requires_grad_(True)
    fy = F.relu(_rand(b * t, c, h, w, seed=61) + 0.2)
    loss, _ = orc.contextual_loss(fx.repeat_interleave(t, dim=0), fy, 0.5)
    (loss * 0.37).backward()
    fxa = _to_act(fx.detach(), dev).mark_grad()
    tr.eng.tape = []
    got = tr.contextual_features(fxa, _to_act(fy, dev), t, 0.37)
    for fn in reversed(tr.eng.tape):
        fn()
    tr.eng.tape = None
    _close(got[0], loss.detach(), 1e-5, "cx loss")
    _close(fxa.grad().nchw(), fx.grad, 2e-4, "d cx / d features")


LAYER_CASES = [
    # name, source channel widths (as the engine passes them), h, w, act, stride, with_residual, with_pixmul
    ("recon_trunk.0.conv1", (64,), 16, 16, 1, 1, False, False),
    ("recon_trunk.0.conv2", (64,), 12, 20, 0, 1, True, False),
    ("fusion_fea_block1.0.conv2", (64,), 16, 16, 0, 1, True, True),
    ("down_fea_conv1", (64,), 16, 16, 0, 2, False, False),
    ("down_fea_conv2", (64, 64), 16, 32, 0, 2, False, False),
    ("reffusionconv2", (64, 128, 64), 8, 8, 0, 1, False, False),
    ("reffea_L2_conv1", (64,), 8, 12, 2, 1, False, False),
    ("upconv1", (64,), 8, 8, 2, 1, False, False),
    ("align_module.flowdsconv1_1", (16,), 16, 16, 0, 2, False, False),
    ("align_module.L3_offset_conv1", (64, 64, 48), 8, 8, 2, 1, False, False),
    ("align_module.L1_dcnpack.conv_offset", (64,), 8, 8, 0, 1, False, False),
    ("refmaskconv3", (64,), 8, 8, 4, 1, False, False),
    ("conv_last", (64,), 16, 32, 0, 1, True, False),
    ("ThreeDA.feat_fusion", (320,), 8, 8, 2, 1, True, False),
    ("reduce_dim_conv", (64, 128, 64), 8, 8, 0, 1, False, False),
]


@pytest.mark.parametrize("case", LAYER_CASES)
def test_recorded_convolution_backward(case):
    """Engine.conv with the tape on: dX per source, dW, db (and d residual, d mask) against torch autograd."""
    from gpemsr_amd import ops
    name, widths, h, w, act, stride, with_res, with_mul = case
    tr = _trainer()
    eng, dev = tr.eng, _dev()
    W = eng.sd[name + ".weight"].detach().cpu()
    Bv = eng.sd[name + ".bias"].detach().cpu()
    transposed = name.startswith("reffea_L")
    ps = name.startswith("upconv")
    cin_total = W.shape[0] if transposed else W.shape[1]
    n = 2
    xs, acts, c0 = [], [], 0
    for i, cw in enumerate(widths):
        ci = min(cw, cin_total - c0)
        x = _rand(n, cw, h, w, seed=70 + i)
        if ci < cw:
            x[:, ci:] = 0.0                                             # padded channels of the 48-wide POD buffer
        x.requires_grad_(True)
        xs.append(x)
        acts.append(_to_act(x.detach(), dev).mark_grad())
        c0 += ci
    xin = xs[0]
    if len(xs) > 1:
        pieces, c0 = [], 0
        for x in xs:
            ci = min(x.shape[1], cin_total - c0)
            pieces.append(x[:, :ci]); c0 += ci
        xin = torch.cat(pieces, 1)
    Wt = W.clone().requires_grad_(True)
    Bt = Bv.clone().requires_grad_(True)
    if transposed:
        z = F.conv_transpose2d(xin, Wt, Bt, stride=2, padding=1, output_padding=1)
    else:
        z = F.conv2d(xin, Wt, Bt, stride, W.shape[2] // 2)
    if ps:
        z = F.pixel_shuffle(z, 2)
    y = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.1), 4: lambda v: torch.sigmoid(F.leaky_relu(v, 0.1))}[act](z)
    res = mul = None
    kw = {}
    if with_res:
        res = _rand(*y.shape, seed=80).requires_grad_(True)
        y = y + res
        kw["residual"] = _to_act(res.detach(), dev).mark_grad()
    if with_mul:
        mul = torch.rand(n, 1, y.shape[2], y.shape[3], generator=torch.Generator().manual_seed(81)).requires_grad_(True)
        y = y * mul
        kw["pixmul"] = _to_act(mul.detach(), dev).mark_grad()
    gy = _rand(*y.shape, seed=82)
    y.backward(gy)

    base = name
    tr.flat_g.zero_()
    eng.tape = []
    if stride != 1:
        kw["stride"] = stride
    out = eng.conv(acts if len(acts) > 1 else acts[0], name, act, **kw)
    _close(out.nchw(), y.detach(), 2e-5, name + " forward")
    ops.axpy(_to_act(gy, dev), out.grad())
    for fn in reversed(eng.tape):
        fn()
    eng.tape = None
    torch.cuda.synchronize()
    _close(tr.gw[base], Wt.grad, 5e-5, name + " dW")
    _close(tr.gb[base], Bt.grad, 5e-5, name + " db")
    for x, a in zip(xs, acts):
        ci = min(x.shape[1], cin_total)
        _close(a.grad().nchw()[:, :ci], x.grad[:, :ci], 5e-5, name + " dX")
    if with_res:
        _close(kw["residual"].grad().nchw(), res.grad, 2e-5, name + " d residual")
    if with_mul:
        _close(kw["pixmul"].grad().nchw(), mul.grad, 5e-5, name + " d mask")
    tr.flat_g.zero_()


def test_dcn_layer_backward():
    """Engine.dcn (conv_offset -> deformable columns -> contraction) against autograd of the oracle's DCNv2Pack."""
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    tr = _trainer()
    eng, dev = tr.eng, _dev()
    p = "align_module.L1_dcnpack"
    sd = {k: eng.sd[k].detach().cpu().clone().requires_grad_(True) for k in (p + ".weight", p + ".bias", p + ".conv_offset.weight", p + ".conv_offset.bias")}
    x = _rand(2, 64, 10, 12, seed=90).requires_grad_(True)
    feat = _rand(2, 64, 10, 12, seed=91, scale=3.0).requires_grad_(True)
    y = orc.dcn_v2_pack(sd, p, x, feat)
    gy = _rand(*y.shape, seed=92)
    y.backward(gy)
    xa, fa = _to_act(x.detach(), dev).mark_grad(), _to_act(feat.detach(), dev).mark_grad()
    tr.flat_g.zero_()
    eng.tape = []
    out = eng.dcn(xa, fa, p, 0)
    _close(out.nchw(), y.detach(), 3e-5, "dcn forward")
    ops.axpy(_to_act(gy, dev), out.grad())
    for fn in reversed(eng.tape):
        fn()
    eng.tape = None
    _close(tr.gw[p], sd[p + ".weight"].grad, 5e-5, "dcn dW"); _close(tr.gb[p], sd[p + ".bias"].grad, 5e-5, "dcn db")
    _close(tr.gw[p + ".conv_offset"], sd[p + ".conv_offset.weight"].grad, 1e-4, "conv_offset dW")
    _close(xa.grad().nchw(), x.grad, 5e-5, "dcn dx"); _close(fa.grad().nchw(), feat.grad, 1e-4, "dcn dfeat")
    tr.flat_g.zero_()


# ------------------------------------------------------------------------------------------------ (3) the reference golden
def _grad_stat_errors(tr, names, stats, projection):
    errs = {}
    for i, k in enumerate(names):
        base, leaf = k.rsplit(".", 1)
        want = stats[i]
        if want[0] == 0.0:
            continue
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    return errs


@pytest.mark.parametrize("scale", [8, 16])
def test_gradient_distance_to_fp64_against_the_references_own(scale, golden_dir):
    """VERDICT r01 item 6c.  Both fp32 evaluations of step 1 -- the unmodified reference's (tests/golden/train_x8.npz) and the HIP
    one -- are measured against the SAME network differentiated in float64 (tests/golden/train_x8_fp64.npz, emitted by the
    unmodified reference model cast to double, oracle/gen_golden_train64.py; code indices and SpyNet flows forced alike).
    Per tensor: e = max(|norm - norm64|, |projection - projection64|) / norm64."""
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d, d64 = np.load(os.path.join(golden_dir, f"train_x{scale}.npz")), np.load(os.path.join(golden_dir, f"train_x{scale}_fp64.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{scale}.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    names = [str(n) for n in d["grad_names"]]
    assert names == [str(n) for n in d64["grad_names"]]
    sr64 = torch.from_numpy(d64["SR64"])
    print("SR distance to fp64: HIP %.2e | reference %.2e" % (
        float((tr.last_sr.view(sr64.shape).double().cpu() - sr64).abs().max() / sr64.abs().max()),
        float((torch.from_numpy(d["SR"]).double() - sr64).abs().max() / sr64.abs().max())))
    s64, s32 = d64["grad_stats64"], d["grad_stats"]
    e_hip = _grad_stat_errors(tr, names, s64, projection)
    e_ref = {k: max(abs(s32[i, 0] - s64[i, 0]), abs(s32[i, 2] - s64[i, 2])) / s64[i, 0] for i, k in enumerate(names) if s64[i, 0] > 0}
    keys = sorted(e_hip, key=lambda k: -e_hip[k])
    print("loss vs fp64: rec HIP %.2e / reference %.2e; ref HIP %.2e / reference %.2e" % (
        abs(rec.item() - float(d64["rec_loss_1"])) / float(d64["rec_loss_1"]), abs(float(d["rec_loss_1"]) - float(d64["rec_loss_1"])) / float(d64["rec_loss_1"]),
        abs(ref.item() - float(d64["ref_loss_1"])) / float(d64["ref_loss_1"]), abs(float(d["ref_loss_1"]) - float(d64["ref_loss_1"])) / float(d64["ref_loss_1"])))
    print("gradient distance to fp64 (HIP | reference), worst HIP tensors:")
    for k in keys[:12]:
        print(f"   {k:55s} {e_hip[k]:.2e} | {e_ref[k]:.2e}")
    hv, rv = np.array([e_hip[k] for k in keys]), np.array([e_ref[k] for k in keys])
    print(f"   median {np.median(hv):.2e} | {np.median(rv):.2e};  90th pct {np.percentile(hv, 90):.2e} | {np.percentile(rv, 90):.2e};  max {hv.max():.2e} | {rv.max():.2e};"
          f"  tensors with HIP <= reference: {int((hv <= rv).sum())}/{len(hv)}")
    downstream = ("recon_trunk.", "upconv", "HRconv", "conv_last")
    worse = [k for k in keys if e_hip[k] > 2.0 * e_ref[k] + 2e-5]
    if scale == 16:
        # measured: HIP is CLOSER to float64 than the reference on 198 of 214 tensors (median 1.5e-5 vs 1.4e-4)
        assert np.median(hv) <= np.median(rv) and np.percentile(hv, 90) <= np.percentile(rv, 90) and hv.max() <= rv.max()
        assert len(worse) <= len(keys) // 20, worse
    else:
        # On THIS input one pre-activation of the 4x4 ThreeDA spatial-attention pyramid lies within fp32 rounding of a LeakyReLU /
        # max-pool kink: the reference's fp32 and float64 runs fall on one side, the HIP forward (same SR to 1.7e-7, different
        # summation order) on the other, and everything the pyramid's backward feeds moves with it (worst: spatial_attn3.weight,
        # 4.6e-2).  The x16 input above has no such element.  So here: every tensor DOWNSTREAM of the fusion (untouched by that
        # kink) must be as close to float64 as the reference, the rest stays inside the old allowance and is reported.
        for k in keys:
            if k.startswith(downstream):
                assert e_hip[k] <= 2.0 * e_ref[k] + 2e-5, (k, e_hip[k], e_ref[k])
        assert hv.max() <= 6e-2 and np.median(hv) <= 1e-3
        assert all(not k.startswith(downstream) for k in worse)


def test_winograd_frozen_option_stays_fp32_grade(golden_dir):
    """Train option `winograd_frozen` (off by default): the frozen 3x3 layers' forward and data gradients in the Winograd F(2x2,3x3) form.
    Stated bars, x16 golden input: losses as close to float64 as the reference's own fp32 step, SR within 3e-7, per-tensor gradient
    distance to float64 median <= 4e-4 / max <= 3e-3 (measured 2.2e-4 / 1.5e-3; the reference's fp32 step: 1.4e-4 / 7.2e-4; the direct
    kernels: 1.5e-5 / 3.1e-4) -- fp32-grade, about 1.5x the reference's own distance, which is why the option is not the default."""
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d, d64 = np.load(os.path.join(golden_dir, "train_x16.npz")), np.load(os.path.join(golden_dir, "train_x16_fp64.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x16.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), dict(TRAIN_OPT, winograd_frozen=True), dev)
    assert tr.eng.wino_train == 7 and any(pc.wino is not None for pc in tr.eng.pc.values())
    assert all(tr.eng.pc[n].wino is None for n in tr.eng.trainable if n in tr.eng.pc), "a trainable layer must not keep a Winograd form"
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    names = [str(n) for n in d["grad_names"]]
    sr64 = torch.from_numpy(d64["SR64"])
    assert float((tr.last_sr.view(sr64.shape).double().cpu() - sr64).abs().max() / sr64.abs().max()) <= 3e-7
    assert abs(rec.item() - float(d64["rec_loss_1"])) / float(d64["rec_loss_1"]) <= 2e-7
    assert abs(ref.item() - float(d64["ref_loss_1"])) / float(d64["ref_loss_1"]) <= 1e-5
    e = np.array(list(_grad_stat_errors(tr, names, d64["grad_stats64"], projection).values()))
    print(f"winograd_frozen: gradient distance to fp64 median {np.median(e):.2e}, 90th pct {np.percentile(e, 90):.2e}, max {e.max():.2e}")
    assert np.median(e) <= 4e-4 and e.max() <= 3e-3


def test_two_training_steps_match_reference_golden(golden_dir):
    from train_constants import FULL, TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)           # a fresh model: this test updates the weights
    tr = Stage3Trainer(model, TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    idx, flow = torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev)
    names = [str(n) for n in d["grad_names"]]
    downstream = ("recon_trunk.", "upconv", "HRconv", "conv_last")

    # ---- step 1: losses, SR, every gradient tensor
    rec, ref = tr.forward_backward(LR, GT, idx, flow)
    torch.cuda.synchronize()
    assert abs(rec.item() - float(d["rec_loss_1"])) <= 1e-5 * float(d["rec_loss_1"])
    assert abs(ref.item() - float(d["ref_loss_1"])) <= 2e-5 * float(d["ref_loss_1"])
    _close(tr.last_sr.view(d["SR"].shape), torch.from_numpy(d["SR"]), 1e-5, "SR")
    errs = {}
    for i, k in enumerate(names):
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d["grad_stats"][i]
        if want[0] == 0.0:                       # tensors the x8 graph never touches: no gradient in the reference
            assert float(g.abs().max()) == 0.0, k
            continue
        got = (g.norm().item(), (g * projection(k, g.numel())).sum().item())
        errs[k] = max(abs(got[0] - want[0]), abs(got[1] - want[2])) / want[0]
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("gradient parity, worst:", [(k, f"{e:.1e}") for k, e in worst], "median %.1e" % np.median(list(errs.values())))
    for k, e in errs.items():
        assert e <= (1e-3 if k.startswith(downstream) else 6e-2), f"{k}: gradient statistic off by {e:.2e}"
    assert np.median(list(errs.values())) <= 3e-3 and np.percentile(list(errs.values()), 90) <= 1.5e-2
    for k in FULL:
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().cpu().reshape(d["grad__" + k].shape)
        _close(g, torch.from_numpy(d["grad__" + k]), 1e-3 if k.startswith(downstream) else 6e-2, "grad " + k)

    # ---- Adam step 1 (+ scheduler), then step 2 from the updated weights
    tr2 = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    o1 = tr2.step(LR, GT, idx, flow)
    assert abs(o1["lr"] - float(d["lr_after_1"])) <= 1e-12
    sdm = tr2.model.state_dict()
    for k in FULL:                                # first Adam step moves every element by ~lr * sign(g): elements whose
        want = torch.from_numpy(d["param1__" + k])          # gradient is within rounding of zero may go either way
        got = sdm[k].detach().cpu().reshape(want.shape)
        frac_bad = ((got - want).abs() > 1e-6 + 1e-4 * want.abs()).float().mean().item()
        assert frac_bad <= 0.10, f"param after step 1 {k}: {frac_bad:.3f} of the elements differ"
    o2 = tr2.step(LR, GT, idx, flow)
    torch.cuda.synchronize()
    assert abs(o2["rec_loss"].item() - float(d["rec_loss_2"])) <= 2e-3 * float(d["rec_loss_2"])
    assert abs(o2["ref_loss"].item() - float(d["ref_loss_2"])) <= 2e-3 * float(d["ref_loss_2"])
    assert abs(o2["lr"] - float(d["lr_after_2"])) <= 1e-12
    print("step-2 losses", o2["rec_loss"].item(), float(d["rec_loss_2"]), o2["ref_loss"].item(), float(d["ref_loss_2"]))


def test_training_reduces_the_loss_and_is_repeatable():
    """Size-independent properties: a few steps on one batch lower the total loss; two trainers from the same state give
    the same losses BIT FOR BIT (deterministic reductions -- the deformable scatter, once the only float-atomic kernel of
    the step, accumulates 64-bit fixed point with integer atomics: gpemsr_dcn_columns_bwd_det)."""
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    topt = dict(lr_G=1e-4, beta1=0.9, beta2=0.99, T_period=[1000, 1000], restarts=[1000], restart_weights=[1], eta_min=1e-7,
                rec_loss_factor=1, ref_loss_factor=0.001)
    LR = synth_lr_tiles(1, 5, 32, 32, seed=5, kind="smooth").to(dev)           # the reference's training crop: 32 -> 256
    GT = torch.rand(1, 1, 256, 256, generator=torch.Generator().manual_seed(6)).to(dev)
    runs = []
    for _ in range(2):
        tr = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), topt, dev)
        losses = []
        for _ in range(4):
            o = tr.step(LR, GT)
            losses.append(o["rec_loss"].item() + 0.001 * o["ref_loss"].item())
        runs.append(losses)
    print("losses", runs[0])
    assert all(np.isfinite(runs[0])) and runs[0][-1] < runs[0][0]
    assert runs[0] == runs[1], (runs[0], runs[1])


def test_bf16x3_frozen_forward_meets_the_same_bar(golden_dir):
    """precision='bf16x3' in training: the frozen sub-networks' forward convolutions on the split-bf16 kernel (fp32-grade);
    losses and the gradients downstream of the alignment must stay within the fp32 path's tolerances of the reference."""
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision="bf16x3").to(dev), TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    assert abs(rec.item() - float(d["rec_loss_1"])) <= 1e-4 * float(d["rec_loss_1"])
    assert abs(ref.item() - float(d["ref_loss_1"])) <= 1e-3 * float(d["ref_loss_1"])
    _close(tr.last_sr.view(d["SR"].shape), torch.from_numpy(d["SR"]), 1e-3, "SR")
    names = [str(n) for n in d["grad_names"]]
    errs = {}
    for i, k in enumerate(names):
        want = d["grad_stats"][i]
        if want[0] == 0.0:
            continue
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    print("bf16x3 training: worst", sorted(errs.items(), key=lambda kv: -kv[1])[:3], "median %.1e" % np.median(list(errs.values())))
    assert max(errs.values()) <= 6e-2 and np.median(list(errs.values())) <= 5e-3


def test_fast_refresh_equals_the_layer_by_layer_repack():
    """The per-step repack of the trainable layers as ONE gather from the flat parameter buffer (TrainEngine.enable_fast_refresh; the
    index map comes from packing index-valued tensors through the same routines) must leave exactly what the layer-by-layer repack
    leaves: every packed weight, bias and parameter table, for x8 and x16."""
    from train_constants import TRAIN_OPT
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    for scale in (8, 16):
        opt = load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{scale}.yml"))
        tr = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
        eng = tr.eng
        assert eng._ridx is not None and int(eng._rmask.sum()) >= tr.n_params        # every trainable element is mapped at least once
        g = torch.Generator(device="cpu").manual_seed(5)
        tr.flat_p.mul_(1.0 + 0.1 * torch.rand(tr.flat_p.numel(), generator=g).to(dev))
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
        assert n > 80
        del tr


def test_bf16_frozen_subnetworks_training_step(golden_dir):
    """precision='bf16' in training -- the frozen sub-networks that carry no gradient (VQGAN prior, VGG relu1_2 mask,
    SpyNet) on the bf16 DATA PATH of the inference engine, everything trained or differentiated through as in 'bf16x3'.  Against the
    reference's own step (code indices and flows teacher-forced like the fp32 test): the SR output at the bf16 inference bar (1e-3),
    the L1 loss within 1e-3, the contextual loss within 2e-2 (its TARGET features come from the bf16 prior image, 7e-3 relative), and
    the gradients within the bars measured for a 7e-3 perturbation of the prior features (printed)."""
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d = np.load(os.path.join(golden_dir, "train_x8.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision="bf16").to(dev), TRAIN_OPT, dev)
    assert tr.eng._frozen16 is not None and tr.eng._frozen16.bf16
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    e_rec = abs(rec.item() - float(d["rec_loss_1"])) / float(d["rec_loss_1"])
    e_ref = abs(ref.item() - float(d["ref_loss_1"])) / float(d["ref_loss_1"])
    _close(tr.last_sr.view(d["SR"].shape), torch.from_numpy(d["SR"]), 1e-3, "SR")
    names = [str(n) for n in d["grad_names"]]
    errs = {}
    for i, k in enumerate(names):
        want = d["grad_stats"][i]
        if want[0] == 0.0:
            continue
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    print(f"bf16 frozen sub-networks: rec loss err {e_rec:.1e}, ref loss err {e_ref:.1e}; gradients worst",
          sorted(errs.items(), key=lambda kv: -kv[1])[:3], "median %.1e" % np.median(list(errs.values())))
    assert e_rec <= 1e-3 and e_ref <= 2e-2
    # per-tensor statistics against the reference: small kink-sensitive tensors (offset convolutions: floor() of the deformable sampling,
    # LeakyReLU kinks) move by up to ~0.2 when the prior features move by 2^-8 (measured: worst 0.21, median 3e-2) ...
    assert max(errs.values()) <= 0.35 and np.median(list(errs.values())) <= 6e-2
    # ... while the gradient as a whole stays within 1e-2 of the exact-fp32 HIP step (measured 3.7e-3; bf16x3: 2e-4)
    g16 = tr.flat_g.clone()
    tr32 = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    tr32.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    whole = float((g16 - tr32.flat_g).norm() / tr32.flat_g.norm())
    print(f"whole-gradient distance to the fp32 step: {whole:.2e}; SR distance {float((tr.last_sr - tr32.last_sr).abs().max() / tr32.last_sr.abs().max()):.2e}")
    assert whole <= 1e-2
    del tr32
    # a full step runs (Adam, scheduler, repack) and the free-running SpyNet path of the bf16 engine is exercised
    o = tr.step(LR, GT)
    torch.cuda.synchronize()
    assert np.isfinite(float(o["rec_loss"].item())) and np.isfinite(float(o["ref_loss"].item()))


def test_x16_training_step_matches_reference_golden(golden_dir):
    """x16 (option/output_GPEMSR_x16.yml): every x16-only layer (reffea_L4_conv1, reffusionconv4, fusion_fea_block4,
    down_fea_conv3, upconv4) receives a gradient; losses and gradient statistics against the reference's step."""
    from train_constants import TRAIN_OPT, projection
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.train import Stage3Trainer
    d = np.load(os.path.join(golden_dir, "train_x16.npz"))
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x16.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), TRAIN_OPT, dev)
    LR, GT = torch.from_numpy(d["LR"]).to(dev), torch.from_numpy(d["GT"]).to(dev)
    rec, ref = tr.forward_backward(LR, GT, torch.from_numpy(d["code_idx"]).to(dev), torch.from_numpy(d["flow"]).to(dev))
    torch.cuda.synchronize()
    assert abs(rec.item() - float(d["rec_loss_1"])) <= 1e-5 * float(d["rec_loss_1"])
    assert abs(ref.item() - float(d["ref_loss_1"])) <= 2e-5 * float(d["ref_loss_1"])
    _close(tr.last_sr.view(d["SR"].shape), torch.from_numpy(d["SR"]), 1e-5, "SR")
    names = [str(n) for n in d["grad_names"]]
    downstream = ("recon_trunk.", "upconv", "HRconv", "conv_last")
    errs = {}
    for i, k in enumerate(names):
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base].detach().reshape(-1).double().cpu()
        want = d["grad_stats"][i]
        assert want[0] > 0.0, k
        errs[k] = max(abs(g.norm().item() - want[0]), abs((g * projection(k, g.numel())).sum().item() - want[2])) / want[0]
    print("x16 gradient parity, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:4], "median %.1e" % np.median(list(errs.values())))
    for k, e in errs.items():
        assert e <= (1e-3 if k.startswith(downstream) else 6e-2), f"{k}: gradient statistic off by {e:.2e}"
    assert np.median(list(errs.values())) <= 3e-3 and np.percentile(list(errs.values()), 90) <= 1.5e-2


def test_training_step_non_square_ragged_crop_matches_oracle():
    """LR 24x32 (SR 192x256; the latent token count must stay a multiple of 32): non-square maps whose rows/cols are ragged
    against the 32-pixel conv / wgrad tiles and the 8x8 DCN tiles, 48x64 contextual-loss positions.  Losses and
    upsampler-side gradients against the CPU oracle under torch autograd (code indices teacher-forced)."""
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    from oracle import gpemsr_oracle as orc
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    model = build_model(opt, load_prior_files=False).to(dev)
    topt = dict(lr_G=1e-4, beta1=0.9, beta2=0.99, T_period=[1000, 1000], restarts=[1000], restart_weights=[1], eta_min=1e-7,
                rec_loss_factor=1, ref_loss_factor=0.001)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tr = Stage3Trainer(model, topt, dev)
    LR = synth_lr_tiles(1, 5, 24, 32, seed=21, kind="smooth")
    GT = torch.rand(1, 1, 192, 256, generator=torch.Generator().manual_seed(22))
    probe = ("conv_last.weight", "HRconv.bias", "upconv1.weight", "recon_trunk.0.conv1.weight")
    for k in probe:
        sd[k].requires_grad_(True)
    tro = {}
    out, ref = orc.gpemsr_forward(sd, LR, scale=8, trace=tro)
    rec_o, ref_o, _ = orc.stage3_losses(sd, out, ref.detach(), GT)
    (rec_o + 0.001 * ref_o).backward()
    rec, refl = tr.forward_backward(LR.to(dev), GT.to(dev), tro["code_idx"].to(dev))
    torch.cuda.synchronize()
    assert abs(rec.item() - rec_o.item()) <= 1e-5 * rec_o.item()
    assert abs(refl.item() - ref_o.item()) <= 1e-4 * ref_o.item()
    for k in probe:
        base, leaf = k.rsplit(".", 1)
        g = (tr.gw if leaf == "weight" else tr.gb)[base]
        _close(g, sd[k].grad, 1e-3, "grad " + k)


def test_trainer_resume_continues_bit_for_bit_on_the_optimizer_side():
    """Save after one step (model.state_dict() + trainer.state_dict()), rebuild, load, take the second step: parameters equal
    those of the uninterrupted run, bit for bit (no float atomics anywhere in the step)."""
    import io
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    topt = dict(lr_G=2e-4, beta1=0.9, beta2=0.99, T_period=[4, 6], restarts=[4], restart_weights=[0.5], eta_min=1e-7,
                rec_loss_factor=1, ref_loss_factor=0.001)
    LR = synth_lr_tiles(1, 5, 16, 16, seed=31, kind="smooth").to(dev)
    GT = torch.rand(1, 1, 128, 128, generator=torch.Generator().manual_seed(32)).to(dev)
    a = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), topt, dev)
    a.step(LR, GT)
    buf = io.BytesIO()
    torch.save({"model": a.model.state_dict(), "trainer": a.state_dict()}, buf)
    a.step(LR, GT)
    ck = torch.load(io.BytesIO(buf.getvalue()), map_location="cpu", weights_only=False)
    m2 = build_model(opt, load_prior_files=False)
    m2.load_state_dict(ck["model"], strict=True)
    b = Stage3Trainer(m2.to(dev), topt, dev)
    b.load_state_dict(ck["trainer"])
    assert b.step_count == 1 and b.lr == ck["trainer"]["lr"]
    b.step(LR, GT)
    assert a.lr == b.lr and a.step_count == b.step_count == 2
    assert torch.equal(b.flat_p, a.flat_p), float((b.flat_p - a.flat_p).abs().max())
    assert torch.equal(b.flat_m, a.flat_m), float((b.flat_m - a.flat_m).abs().max())


def test_optimizer_state_round_trips_through_torch_adam_format():
    """the flat Adam moments export as / import from ``torch.optim.Adam.state_dict()`` (what train_stage3.py:183
    saves), so a run can move between the reference loop and this trainer.  Cross-check with a real torch Adam over the same
    parameter list: it must accept the exported dict, and a trainer loaded from torch's own state dict continues identically."""
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    topt = dict(lr_G=2e-4, beta1=0.9, beta2=0.99, T_period=[4, 6], restarts=[4], restart_weights=[0.5], eta_min=1e-7,
                rec_loss_factor=1, ref_loss_factor=0.001)
    LR = synth_lr_tiles(1, 5, 16, 16, seed=41, kind="smooth").to(dev)
    GT = torch.rand(1, 1, 128, 128, generator=torch.Generator().manual_seed(42)).to(dev)
    a = Stage3Trainer(build_model(opt, load_prior_files=False).to(dev), topt, dev)
    a.step(LR, GT)
    sd = a.torch_optimizer_state_dict()
    params = [p for _, p in a.model.named_parameters() if p.requires_grad]
    tadam = torch.optim.Adam(params, lr=topt["lr_G"], betas=(0.9, 0.99))
    tadam.load_state_dict(sd)                                             # torch accepts the layout
    back = tadam.state_dict()
    assert len(back["state"]) == len(params) and float(back["state"][0]["step"]) == 1.0
    weights = {k: v.clone() for k, v in a.model.state_dict().items()}
    a.step(LR, GT)
    m2 = build_model(opt, load_prior_files=False)
    m2.load_state_dict(weights, strict=True)
    b = Stage3Trainer(m2.to(dev), topt, dev)
    b.load_torch_optimizer_state_dict(back, scheduler_state={"last_epoch": 1})
    assert b.step_count == 1
    b.step(LR, GT)
    assert b.step_count == a.step_count == 2 and abs(a.lr - b.lr) < 1e-12
    _close(b.flat_p, a.flat_p, 1e-6, "parameters after a resume through the torch-Adam layout")


def test_validation_between_training_steps_uses_the_current_weights():
    """step, validate, step, validate -- the inference engine's packed weights must follow the optimizer
    (R:train_stage3.py:197-312 validates every val_freq steps).  The second validation equals a freshly built model that loaded
    the current state dict, and differs from the first."""
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    topt = dict(lr_G=4e-4, beta1=0.9, beta2=0.99, T_period=[1 << 20], restarts=None, restart_weights=None, eta_min=1e-7,
                rec_loss_factor=1, ref_loss_factor=0.001)
    LR = synth_lr_tiles(1, 5, 16, 16, seed=51, kind="smooth").to(dev)
    GT = torch.rand(1, 1, 128, 128, generator=torch.Generator().manual_seed(52)).to(dev)
    model = build_model(opt, load_prior_files=False).to(dev)
    tr = Stage3Trainer(model, topt, dev)
    tr.step(LR, GT)
    model.eval()
    with torch.no_grad():
        v1, _ = model(LR)
    model.train()
    tr.step(LR, GT)
    model.eval()
    with torch.no_grad():
        v2, _ = model(LR)
    fresh = build_model(opt, load_prior_files=False)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    fresh = fresh.eval().to(dev)
    with torch.no_grad():
        vf, _ = fresh(LR)
    torch.cuda.synchronize()
    assert torch.equal(v2, vf), float((v2 - vf).abs().max())
    assert float((v2 - v1).abs().max()) > 0.0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config5_geometry_batch8_of_32_to_256(precision):
    """BASELINE configs[4] at its REAL geometry (R:option/train_stage3_x8.yml: batch_size 8, LQ_size 32, GT_size 256; the step of
    R:train_stage3.py:343-366), driven by the `train:` block of option/train_stage3_x8.yml: finite losses and gradient; repeatable BIT FOR
    BIT (the deformable scatter accumulates in fixed point: no float atomics in the step); and -- both losses are batch means -- the gradient of the batch equals
    the mean of the eight single-sample gradients to 1e-4 of its largest entry.  fp32 and the bf16 frozen-sub-network mode."""
    from gpemsr_amd.config import build_model, load_options
    from gpemsr_amd.synth import synth_lr_tiles
    from gpemsr_amd.train import Stage3Trainer
    dev = _dev()
    topt_all = load_options(os.path.join(ROOT, "option", "train_stage3_x8.yml"))
    ds = topt_all["datasets"]["train"]
    B, lq, gt = int(ds["batch_size"]), int(ds["LQ_size"]), int(ds["GT_size"])
    assert (B, lq, gt, int(topt_all["scale"])) == (8, 32, 256, 8)
    opt = load_options(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml"))
    tr = Stage3Trainer(build_model(opt, load_prior_files=False, precision=precision).to(dev), dict(topt_all["train"]), dev)
    assert tr.lr == 4e-4 and tr.adam_hparams()[:2] == (0.9, 0.99)
    LR = synth_lr_tiles(B, 5, lq, lq, seed=55, kind="smooth").to(dev)
    GT = torch.rand(B, 1, gt, gt, generator=torch.Generator().manual_seed(56)).to(dev)

    def grad_of(lr_, gt_):
        rec, ref = tr.forward_backward(lr_, gt_)
        torch.cuda.synchronize()
        return float(rec.item()), float(ref.item()), tr.flat_g.detach().clone()
    rec, ref, g = grad_of(LR, GT)
    assert np.isfinite(rec) and np.isfinite(ref) and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    rec2, ref2, g2 = grad_of(LR, GT)
    gmax = float(g.abs().max())
    assert rec == rec2 and ref == ref2, (rec, rec2, ref, ref2)
    assert torch.equal(g, g2), float((g - g2).abs().max()) / gmax
    acc, recs, refs = torch.zeros_like(g), [], []
    for i in range(B):
        r1, f1, gi = grad_of(LR[i:i + 1], GT[i:i + 1])
        acc += gi; recs.append(r1); refs.append(f1)
    acc /= B
    err = float((g - acc).abs().max()) / gmax
    print(f"config 5 geometry, {precision}: rec {rec:.5f} ref {ref:.5f}; |grad(batch) - mean grad(sample)| / max|grad| = {err:.2e}")
    assert abs(rec - float(np.mean(recs))) <= 1e-5 * abs(rec) and abs(ref - float(np.mean(refs))) <= 1e-4 * abs(ref)
    assert err <= 1e-4, err
    before = tr.flat_p.detach().clone()
    o = tr.step(LR, GT)
    torch.cuda.synchronize()
    assert np.isfinite(o["rec_loss"].item()) and float((tr.flat_p - before).abs().max()) > 0 and bool(torch.isfinite(tr.flat_p).all())
