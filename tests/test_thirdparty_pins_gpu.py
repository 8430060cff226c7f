"""Opportunistic pins of the third-party arithmetic the reference imports (torchvision, basicsr) against the REAL packages when the
machine running the GPU tests has them (otherwise both sides of those comparisons are this repo's own
restatements under oracle/ref_shims).  Each test skips cleanly when its package is missing; the log line says which ran."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_dcn_against_torchvision_deform_conv2d():
    """gpemsr_dcn_columns + the 1x1 contraction == torchvision.ops.deform_conv2d with basicsr's DCNv2Pack conventions (offset
    channels = cat(o1, o2) of the offset conv's three chunks, mask = sigmoid(third chunk), 8 deformable groups)."""
    tvo = pytest.importorskip("torchvision.ops")
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_dcn
    dev = torch.device("cuda", 0)
    n, c, h, w, dg = 2, 64, 12, 16, 8
    x, feat = _rand(n, c, h, w, seed=201), _rand(n, c, h, w, seed=202)
    wt, b = _rand(64, 64, 3, 3, seed=203, scale=0.05), _rand(64, seed=204, scale=0.1)
    wo, bo = _rand(3 * dg * 9, 64, 3, 3, seed=205, scale=0.08), _rand(3 * dg * 9, seed=206)
    out = torch.nn.functional.conv2d(feat, wo, bo, 1, 1)
    o1, o2, mask = torch.chunk(out, 3, dim=1)                         # basicsr/archs/arch_util.py DCNv2Pack.forward
    want = tvo.deform_conv2d(x, torch.cat((o1, o2), dim=1), wt, b, 1, 1, 1, torch.sigmoid(mask))
    A = lambda t: ops.from_nhwc(t.permute(0, 2, 3, 1).contiguous().to(dev))      # noqa: E731
    om = ops.conv2d([A(feat)], pack_conv(wo, bo, dev), 0)
    got = ops.conv2d([ops.dcn_columns(A(x), om, dg)], pack_dcn(wt, b, dev), 0).nchw().cpu()
    err = float((got - want).abs().max() / want.abs().max())
    print(f"torchvision.ops.deform_conv2d pin ran: max rel err {err:.2e}")
    assert err <= 3e-5


def test_vgg19_layer_table_against_torchvision():
    """model._VGG_LAYERS (slice, index, kind) == torchvision.models.vgg19().features as model/VGG.py:17-29 slices it."""
    tvm = pytest.importorskip("torchvision.models")
    from gpemsr_amd.model import _VGG_LAYERS
    feats = tvm.vgg19().features
    bounds = {1: (0, 4), 2: (4, 9), 3: (9, 18), 4: (18, 27), 5: (27, 36)}       # model/VGG.py:17-29
    table = {idx: (sl, kind) for sl, idx, kind in _VGG_LAYERS}
    for i, layer in enumerate(feats):
        if i >= 36:
            break
        sl = next(s for s, (a, b) in bounds.items() if a <= i < b)
        if isinstance(layer, torch.nn.Conv2d):
            assert table.get(i) == (sl, "conv"), (i, table.get(i))
            assert layer.kernel_size == (3, 3) and layer.padding == (1, 1)
        elif isinstance(layer, torch.nn.MaxPool2d):
            assert table.get(i) == (sl, "pool"), (i, table.get(i))
        else:
            assert isinstance(layer, torch.nn.ReLU) and i not in table
    print("torchvision vgg19 layer-table pin ran")


def test_spynet_and_flow_warp_against_basicsr():
    """The SpyNet forward (6 levels, ceil-32 resize, x2 flow up-scaling, border-mode warp) of the HIP engine == basicsr's own
    SpyNet with the same weights."""
    arch = pytest.importorskip("basicsr.archs.spynet_arch")
    from gpemsr_amd.config import build_model, load_options
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = build_model(load_options(os.path.join(root, "option", "output_GPEMSR_x8.yml")), load_prior_files=False).eval().cuda()
    sd = {k[len("align_module.spynet."):]: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("align_module.spynet.")}
    net = arch.SpyNet(load_path=None) if "load_path" in arch.SpyNet.__init__.__code__.co_varnames else arch.SpyNet()
    net.load_state_dict(sd, strict=False)
    net.eval()
    a, b = torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(5)), torch.rand(2, 1, 64, 96, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        want = net(a.expand(-1, 3, -1, -1), b.expand(-1, 3, -1, -1))
    from gpemsr_amd import ops
    eng = model._get_engine(torch.device("cuda", 0))
    A = lambda t: ops.Act(t.reshape(-1).cuda(), t.shape[0], t.shape[2], t.shape[3], 1, 1, 0)      # noqa: E731  1-channel NHWC == NCHW
    got = eng.spynet(A(a), A(b)).nchw().cpu()
    err = float((got - want).abs().max() / want.abs().max())
    print(f"basicsr SpyNet pin ran: max rel err {err:.2e}")
    assert err <= 2e-3
