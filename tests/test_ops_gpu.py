"""Known-answer tests of every C-ABI op against torch CPU functional ops
(none exist upstream, so the build authors them).  All calls go
through ctypes -> libgpemsr_hip.so; tolerances are fp32-accumulation level."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _to_act(x_nchw, dev, ld=None, off=0):
    """NCHW cpu tensor -> Act on device, optionally embedded in a wider buffer."""
    from gpemsr_amd import ops
    n, c, h, w = x_nchw.shape
    ld = c if ld is None else ld
    buf = torch.full((n, h, w, ld), 7.0)
    buf[..., off:off + c] = x_nchw.permute(0, 2, 3, 1)
    return ops.Act(buf.to(dev).contiguous(), n, h, w, c, ld, off)


def _close(got, want, tol=2e-5, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-6)
    assert err <= tol * ref + 1e-6, f"{what}: max err {err:.3e} vs ref max {ref:.3e}"


CONV_CASES = [
    # (n, cins, cout, k, stride, h, w, act, residual, pixmul)
    (2, (64,), 64, 3, 1, 16, 16, 1, True, False),
    (1, (1,), 64, 3, 1, 24, 40, 2, False, False),
    (2, (64, 64), 64, 3, 1, 17, 19, 0, True, True),
    (1, (64, 128, 64), 64, 3, 1, 16, 32, 0, False, False),
    (1, (64, 64, 32, 2), 64, 3, 1, 8, 8, 2, False, False),
    (1, (64,), 216, 3, 1, 16, 16, 0, False, False),
    (2, (64,), 64, 3, 2, 32, 32, 2, False, False),
    (1, (256,), 512, 3, 2, 16, 16, 0, False, False),
    (1, (128,), 256, 3, 1, 16, 16, 0, False, False),
    (1, (8,), 32, 7, 1, 32, 32, 1, False, False),
    (1, (32,), 64, 7, 1, 16, 48, 1, False, False),
    (1, (64,), 32, 7, 1, 9, 11, 1, False, False),
    (2, (512,), 512, 1, 1, 8, 8, 0, True, False),
    (1, (64, 128, 64), 64, 1, 1, 16, 16, 0, False, False),
    (1, (320,), 64, 1, 1, 12, 20, 2, False, False),
    (1, (512,), 1024, 1, 1, 8, 8, 0, False, False),
    (1, (64,), 20, 3, 1, 16, 16, 3, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_mfma(case):
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    n, cins, cout, k, stride, h, w, act, use_res, use_mul = case
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=1)
    wt = _rand(cout, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k))
    b = _rand(cout, seed=3, scale=0.1)
    want = F.conv2d(x, wt, b, stride, k // 2)
    if act == 1:
        want = F.relu(want)
    elif act == 2:
        want = F.leaky_relu(want, 0.1)
    elif act == 3:
        want = torch.sigmoid(want)
    oh, ow = want.shape[2:]
    res = _rand(n, cout, oh, ow, seed=4) if use_res else None
    mul = torch.rand(n, 1, oh, ow, generator=torch.Generator().manual_seed(5)) if use_mul else None
    if res is not None:
        want = want + res
    if mul is not None:
        want = want * mul
    srcs, off = [], 0
    for i, c in enumerate(cins):      # each source lives in its own (sometimes wider) buffer
        pad = 8 if (i % 2 == 1 and c % 4 == 0) else 0
        srcs.append(_to_act(x[:, off:off + c], dev, ld=c + pad, off=pad // 2 if pad else 0))
        off += c
    pc = pack_conv(wt, b, dev, cins)
    out = ops.conv2d(srcs, pc, act, stride=stride, residual=_to_act(res, dev) if use_res else None,
                     pixmul=_to_act(mul, dev) if use_mul else None, force_mfma=True)
    torch.cuda.synchronize()
    _close(out.nchw(), want, what=f"conv {case}")


def test_conv2d_output_slice_and_images():
    """out may be a channel slice of a wider buffer (free torch.cat of producers)."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    dev = _dev()
    x = _rand(3, 64, 16, 16, seed=11)
    wt = _rand(64, 64, 3, 3, seed=12, scale=0.05)
    b = _rand(64, seed=13)
    want = F.conv2d(x, wt, b, 1, 1)
    big = ops.new_act(3, 16, 16, 160, device=dev)
    big.buf.fill_(-3.0)
    ops.conv2d([_to_act(x, dev)], pack_conv(wt, b, dev), 0, out=big.slice(32, 64))
    torch.cuda.synchronize()
    t = big.torch().cpu()
    _close(t[..., 32:96].permute(0, 3, 1, 2), want, what="slice out")
    assert float(t[..., :32].min()) == -3.0 and float(t[..., 96:].max()) == -3.0
    sub = big.slice(32, 64).images(1, 2)
    _close(sub.nchw(), want[1:3], what="images view")


def test_conv_transpose():
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_convT
    dev = _dev()
    for (n, cin, cout, h, w, act) in [(2, 64, 64, 16, 16, 2), (1, 512, 256, 8, 8, 0), (1, 64, 64, 9, 21, 0), (1, 128, 64, 16, 16, 0)]:
        x = _rand(n, cin, h, w, seed=21)
        wt = _rand(cin, cout, 3, 3, seed=22, scale=1.0 / np.sqrt(cin * 2.25))
        b = _rand(cout, seed=23, scale=0.1)
        want = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
        if act == 2:
            want = F.leaky_relu(want, 0.1)
        out = ops.conv2d([_to_act(x, dev)], pack_convT(wt, b, dev), act)
        torch.cuda.synchronize()
        _close(out.nchw(), want, what=f"convT {(n, cin, cout, h, w)}")


def test_pixel_shuffle_fused():
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    dev = _dev()
    x = _rand(2, 64, 12, 20, seed=31)
    wt = _rand(256, 64, 3, 3, seed=32, scale=0.05)
    b = _rand(256, seed=33, scale=0.1)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, wt, b, 1, 1), 2), 0.1)
    out = ops.conv2d([_to_act(x, dev)], pack_conv(wt, b, dev, pixel_shuffle=True), ops.ACT_LRELU)
    torch.cuda.synchronize()
    _close(out.nchw(), want, what="pixel shuffle")


def test_batched_weight_gemm():
    """torch.bmm as a 1x1 conv with per-image weights (attention products)."""
    from gpemsr_amd import ops
    dev = _dev()
    n, T, Cc = 3, 64, 96
    q = _rand(n, T, Cc, seed=41)
    k = _rand(n, T, Cc, seed=42)
    want = torch.bmm(q, k.transpose(1, 2))                       # [n,T,T]
    qa = ops.Act(q.to(dev).contiguous(), n, T // 16, 16, Cc, Cc, 0)
    kd = k.to(dev).contiguous()
    S = ops.conv2d([qa], ops.PackedConv(kd, None, 1, T, (Cc,), 32), 0, weight_image_stride=T * Cc)
    torch.cuda.synchronize()
    _close(S.torch().reshape(n, T, T), want, what="bmm")
    # shared A (image stride 0) with per-image weights
    a = _rand(32, Cc, seed=43)
    want2 = torch.einsum("ic,njc->nij", a, k)
    aa = ops.Act(a.to(dev).contiguous(), n, 2, 16, Cc, Cc, 0)
    o2 = ops.new_act(n, 2, 16, T, device=dev)
    ops.conv2d([aa], ops.PackedConv(kd, None, 1, T, (Cc,), 32), 0, weight_image_stride=T * Cc, src_image_stride=[0], out=o2)
    torch.cuda.synchronize()
    _close(o2.torch().reshape(n, 32, T), want2, what="shared-A bmm")


DIRECT_CASES = [(2, 64, 1, 3, 1, 32, 32), (1, 16, 2, 7, 1, 16, 24), (1, 2, 16, 3, 4, 64, 64), (2, 16, 16, 3, 2, 16, 16),
                (1, 64, 1, 3, 1, 7, 9), (3, 64, 1, 3, 1, 20, 37)]


@pytest.mark.parametrize("case", DIRECT_CASES)
def test_conv2d_direct(case):
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    n, cin, cout, k, stride, h, w = case
    dev = _dev()
    x = _rand(n, cin, h, w, seed=51)
    wt = _rand(cout, cin, k, k, seed=52, scale=1.0 / np.sqrt(cin * k * k))
    b = _rand(cout, seed=53, scale=0.1)
    want = F.leaky_relu(F.conv2d(x, wt, b, stride, k // 2), 0.1)
    res = _rand(*want.shape, seed=54)
    want = want + res
    out = ops.conv2d([_to_act(x, dev, ld=cin + (16 if cin % 4 == 0 else 0))], pack_conv(wt, b, dev), ops.ACT_LRELU, stride=stride,
                     residual=_to_act(res, dev))
    torch.cuda.synchronize()
    _close(out.nchw(), want, what=f"direct {case}")


def test_groupnorm_relu_residual():
    from gpemsr_amd import ops
    dev = _dev()
    for (n, c, h, w) in [(2, 64, 16, 16), (1, 512, 8, 8), (3, 128, 32, 32), (1, 256, 16, 16)]:
        x = _rand(n, c, h, w, seed=61) * 3 + 0.7
        g, b = _rand(c, seed=62) + 1.5, _rand(c, seed=63)
        res = _rand(n, c, h, w, seed=64)
        want = F.relu(F.group_norm(x.double(), 32, g.double(), b.double(), eps=1e-6)).float() + res
        out = ops.groupnorm_relu(_to_act(x, dev), g.to(dev), b.to(dev), True, residual=_to_act(res, dev))
        torch.cuda.synchronize()
        _close(out.nchw(), want, tol=1e-5, what=f"groupnorm {(n, c, h, w)}")
        want2 = F.group_norm(x, 32, g, b, eps=1e-6)
        a = _to_act(x, dev)
        ops.groupnorm_relu(a, g.to(dev), b.to(dev), False, out=a)
        torch.cuda.synchronize()
        _close(a.nchw(), want2, tol=1e-5, what="groupnorm in place")


def test_softmax_argmax_gather():
    from gpemsr_amd import ops, _abi
    dev = _dev()
    for cols in (64, 256, 4096):
        x = _rand(37, cols, seed=71) * 6
        d = x.to(dev).contiguous()
        ops.softmax_rows_(d, 37, cols)
        torch.cuda.synchronize()
        _close(d, F.softmax(x, dim=1), tol=1e-5, what="softmax")
    x = _rand(1001, 1024, seed=72)
    x[5, 100] = x[5, 900] = 9.0          # tie -> lowest index
    x[6, 1023] = 10.0
    a = ops.Act(x.to(dev).contiguous(), 1, 1001, 1, 1024, 1024, 0)
    idx = ops.argmax_rows(a)
    torch.cuda.synchronize()
    assert torch.equal(idx.cpu().long(), torch.argmax(x, dim=1))
    assert int(idx[5]) == 100 and int(idx[6]) == 1023
    table = _rand(1024, 512, seed=73)
    g = ops.gather_rows(table.to(dev), idx, 1, 1001, 1)
    torch.cuda.synchronize()
    assert torch.equal(g.torch().reshape(1001, 512).cpu(), table[idx.cpu().long()])


def test_bilinear_and_pools():
    from gpemsr_amd import ops
    dev = _dev()
    x = _rand(2, 5, 12, 20, seed=81)
    for s in (2, 4, 8, 0.5):
        want = F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False)
        out = ops.bilinear(_to_act(x, dev, ld=8, off=2), want.shape[2], want.shape[3])
        torch.cuda.synchronize()
        _close(out.nchw(), want, tol=1e-5, what=f"bilinear x{s}")
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
    out = ops.bilinear(_to_act(x, dev), 24, 40, align_corners=True, mul=2.0)
    torch.cuda.synchronize()
    _close(out.nchw(), want, tol=1e-5, what="bilinear align_corners")
    want = F.interpolate(x, size=(32, 64), mode="bilinear", align_corners=False)
    out = ops.bilinear(_to_act(x, dev), 32, 64)
    torch.cuda.synchronize()
    _close(out.nchw(), want, tol=1e-5, what="bilinear to size")
    _close(ops.avgpool2(_to_act(x, dev)).nchw(), F.avg_pool2d(x, 2, 2), tol=1e-6, what="avgpool2")
    x = _rand(2, 64, 13, 16, seed=82)
    want = torch.cat([F.max_pool2d(x, 3, 2, 1), F.avg_pool2d(x, 3, 2, 1)], 1)
    _close(ops.pool3s2_maxavg(_to_act(x, dev)).nchw(), want, tol=1e-6, what="pool3s2")


def test_spynet_prep_matches_basicsr_level():
    """One SpyNet level input vs the restated basicsr math (oracle.flow_warp)."""
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    dev = _dev()
    n, h, w = 2, 16, 32
    ref, supp = torch.rand(n, 1, h, w, generator=torch.Generator().manual_seed(91)), torch.rand(n, 1, h, w, generator=torch.Generator().manual_seed(92))
    flow = _rand(n, 2, h // 2, w // 2, seed=93) * 3
    mean = torch.tensor(orc._SPY_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(orc._SPY_STD).view(1, 3, 1, 1)
    up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
    want = torch.cat([(ref - mean) / std, orc.flow_warp((supp - mean) / std, up.permute(0, 2, 3, 1)), up], 1)
    u, inp = ops.spynet_prep(_to_act(ref, dev), _to_act(supp, dev), _to_act(flow, dev), orc._SPY_MEAN, orc._SPY_STD)
    torch.cuda.synchronize()
    _close(u.nchw(), up, tol=1e-5, what="up flow")
    _close(inp.nchw(), want, tol=2e-5, what="spynet level input")
    u0, inp0 = ops.spynet_prep(_to_act(ref, dev), _to_act(supp, dev), None, orc._SPY_MEAN, orc._SPY_STD)
    torch.cuda.synchronize()
    want0 = torch.cat([(ref - mean) / std, (supp - mean) / std, torch.zeros(n, 2, h, w)], 1)
    _close(inp0.nchw(), want0, tol=1e-5, what="spynet level 0")


def test_dcn_matches_torchvision_semantics():
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_dcn
    from oracle import gpemsr_oracle as orc
    dev = _dev()
    n, c, h, w = 2, 64, 12, 16
    x = _rand(n, c, h, w, seed=101)
    feat = _rand(n, c, h, w, seed=102)
    sd = {"d.weight": _rand(64, 64, 3, 3, seed=103, scale=0.05), "d.bias": _rand(64, seed=104, scale=0.1),
          "d.conv_offset.weight": _rand(216, 64, 3, 3, seed=105, scale=0.08), "d.conv_offset.bias": _rand(216, seed=106)}
    want = orc.dcn_v2_pack(sd, "d", x, feat)
    om = ops.conv2d([_to_act(feat, dev)], pack_conv(sd["d.conv_offset.weight"], sd["d.conv_offset.bias"], dev), 0)
    col = ops.dcn_columns(_to_act(x, dev), om, 8)
    out = ops.conv2d([col], pack_dcn(sd["d.weight"], sd["d.bias"], dev), 0)
    torch.cuda.synchronize()
    _close(out.nchw(), want, tol=3e-5, what="dcn")
    # large offsets exercise the out-of-image rule
    sd["d.conv_offset.bias"] = _rand(216, seed=107) * 12
    want = orc.dcn_v2_pack(sd, "d", x, feat)
    om = ops.conv2d([_to_act(feat, dev)], pack_conv(sd["d.conv_offset.weight"], sd["d.conv_offset.bias"], dev), 0)
    out = ops.conv2d([ops.dcn_columns(_to_act(x, dev), om, 8)], pack_dcn(sd["d.weight"], sd["d.bias"], dev), 0)
    torch.cuda.synchronize()
    _close(out.nchw(), want, tol=3e-5, what="dcn big offsets")


def test_patch_cosine_and_threeda_pieces():
    from gpemsr_amd import ops
    from oracle import gpemsr_oracle as orc
    dev = _dev()
    a, b = torch.rand(2, 64, 32, 48, generator=torch.Generator().manual_seed(111)), torch.rand(2, 64, 32, 48, generator=torch.Generator().manual_seed(112))
    out = ops.patch_cosine(_to_act(a, dev), _to_act(b, dev))
    torch.cuda.synchronize()
    _close(out.nchw(), orc.patch_cosine(a, b), tol=1e-5, what="patch cosine")
    B, T, c, h, w = 2, 5, 64, 8, 12
    al, emb, er = _rand(B * T, c, h, w, seed=113), _rand(B * T, c, h, w, seed=114) * 0.3, _rand(B, c, h, w, seed=115)
    corr = torch.sigmoid((emb.view(B, T, c, h, w) * er.unsqueeze(1)).sum(2))          # B,T,h,w
    want = (al.view(B, T, c, h, w) * corr.unsqueeze(2)).reshape(B, T * c, h, w)
    af = ops.temporal_gate(_to_act(al, dev), _to_act(emb, dev), _to_act(er, dev), B, T)
    torch.cuda.synchronize()
    _close(af.nchw(), want, tol=1e-5, what="temporal gate")
    m, bias = _rand(T, T, seed=116), _rand(T, seed=117)
    want2 = F.leaky_relu(torch.einsum("ij,bjchw->bichw", m, want.view(B, T, c, h, w)) + bias.view(1, T, 1, 1, 1), 0.1).reshape(B, T * c, h, w)
    fm = ops.frame_mix_lrelu(af, T, m.to(dev), bias.to(dev))
    torch.cuda.synchronize()
    _close(fm.nchw(), want2, tol=1e-5, what="frame mix")
    f = [_rand(B, c, h, w, seed=120 + i) for i in range(5)]
    want3 = f[0] * torch.sigmoid(f[1]) * 2 + f[2] + f[3] + f[4]
    out = ops.threeda_combine(*[_to_act(t, dev) for t in f])
    torch.cuda.synchronize()
    _close(out.nchw(), want3, tol=1e-5, what="threeda combine")


def test_tensor2img_and_copies():
    from gpemsr_amd import ops
    dev = _dev()
    x = torch.cat([torch.linspace(-0.2, 1.2, 4001), torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 126.5 / 255])])
    want = (x.clamp(0, 1).numpy() * 255.0).round().astype(np.uint8)
    got = ops.tensor2img_u8(x.to(dev)).cpu().numpy()
    assert np.array_equal(got, want)
    src = _rand(10, 3, 4, 8, seed=131)
    a = _to_act(src, dev)
    d = ops.copy_images(a, 4, 2, 5, 3)          # dst j <- src (j//2)*5+3
    torch.cuda.synchronize()
    assert torch.equal(d.nchw().cpu(), src[[3, 3, 8, 8]])


STABILITY_CASES = [(20, 64, 64, 3, 1, 128, 128), (8, 256, 256, 3, 1, 128, 128), (4, 32, 64, 7, 1, 256, 256),
                   (8, 512, 512, 1, 1, 64, 64), (8, 64, 64, 3, 2, 256, 256), (3, 64, 216, 3, 1, 100, 76),
                   (2, 64, 32, 3, 1, 130, 94), (4, 64, 64, 0, 1, 128, 128)]


@pytest.mark.parametrize("case", STABILITY_CASES)
def test_conv_large_grids_are_bit_stable(case):
    """Race screen for the LDS-DMA pipeline (counted vmcnt + raw barriers): production-sized grids (every CU holds
    several co-resident workgroups, ragged tiles, masked DMA lanes) must be correct AND bit-identical run to run."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_convT
    n, cin, cout, k, stride, h, w = case
    dev = _dev()
    x = _rand(n, cin, h, w, seed=201)
    if k == 0:
        wt, b = _rand(cin, cout, 3, 3, seed=202, scale=0.05), _rand(cout, seed=203)
        pc, want = pack_convT(wt, b, dev), F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
    else:
        wt, b = _rand(cout, cin, k, k, seed=202, scale=1.0 / np.sqrt(cin * k * k)), _rand(cout, seed=203)
        pc, want = pack_conv(wt, b, dev), F.conv2d(x, wt, b, stride, k // 2)
    xa = ops.from_nchw(x.to(dev))
    ref = None
    for _ in range(5):
        out = ops.conv2d([xa], pc, 0, stride=stride if k else 1)
        torch.cuda.synchronize()
        o = out.torch().clone()
        if ref is None:
            ref = o
            _close(out.nchw(), want, tol=1e-5, what=f"large conv {case}")
        else:
            assert torch.equal(o, ref), f"non-deterministic result for {case}"


def test_conv_stem_one_channel():
    """1 -> 64 3x3 stem kernel (VGG conv1_1 on the expanded image, conv_first) vs torch."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_vgg_first
    dev = _dev()
    x = torch.rand(3, 1, 37, 52, generator=torch.Generator().manual_seed(301))
    wt, b = _rand(64, 1, 3, 3, seed=302), _rand(64, seed=303)
    out = ops.conv2d([_to_act(x, dev)], pack_conv(wt, b, dev), ops.ACT_LRELU)
    torch.cuda.synchronize()
    _close(out.nchw(), F.leaky_relu(F.conv2d(x, wt, b, 1, 1), 0.1), tol=1e-5, what="stem lrelu")
    w3, b3 = _rand(64, 3, 3, 3, seed=304), _rand(64, seed=305)
    out = ops.conv2d([_to_act(x, dev)], pack_vgg_first(w3, b3, dev), ops.ACT_RELU)
    torch.cuda.synchronize()
    _close(out.nchw(), F.relu(F.conv2d(x.expand(-1, 3, -1, -1), w3, b3, 1, 1)), tol=1e-5, what="vgg conv1_1")


def test_conv_split_transposed():
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_convT, pack_convT_split
    dev = _dev()
    for (n, cin, cout, h, w, act) in [(2, 64, 64, 16, 16, 2), (1, 512, 256, 8, 8, 0), (1, 64, 64, 9, 37, 0), (1, 128, 64, 40, 33, 0)]:
        x = _rand(n, cin, h, w, seed=421)
        wt, b = _rand(cin, cout, 3, 3, seed=422, scale=1.0 / np.sqrt(cin * 2.25)), _rand(cout, seed=423, scale=0.1)
        want = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1, output_padding=1).float()
        if act == 2:
            want = F.leaky_relu(want, 0.1)
        pc = pack_convT(wt, b, dev)
        pc.w16 = pack_convT_split(pc, dev)
        out = ops.conv2d([_to_act(x, dev)], pc, act, precision="bf16x3")
        torch.cuda.synchronize()
        _close(out.nchw(), want, tol=3e-5, what=f"bf16x3 convT {(n, cin, cout, h, w)}")


SPLIT7_CASES = [(2, 32, 64, 20, 44), (1, 64, 32, 33, 32), (1, 32, 16, 16, 70), (3, 16, 40, 9, 9)]


@pytest.mark.parametrize("case", SPLIT7_CASES)
def test_conv_split_7x7(case):
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_conv_split
    n, cin, cout, h, w = case
    dev = _dev()
    x = _rand(n, cin, h, w, seed=411)
    wt, b = _rand(cout, cin, 7, 7, seed=412, scale=1.0 / np.sqrt(cin * 49)), _rand(cout, seed=413, scale=0.1)
    want = F.relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 3)).float()
    pc = pack_conv(wt, b, dev)
    pc.w16 = pack_conv_split(pc, wt, dev)
    out = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_RELU, precision="bf16x3")
    torch.cuda.synchronize()
    _close(out.nchw(), want, tol=3e-5, what=f"bf16x3 7x7 {case}")


SPLIT_CASES = [(2, (64,), 64, 16, 16, 1, True, False, False), (1, (64, 128, 64), 64, 24, 40, 0, False, False, False),
               (2, (256,), 256, 16, 32, 0, False, False, False), (1, (64,), 256, 12, 20, 2, False, False, True),
               (1, (64, 64), 64, 33, 47, 0, True, True, False), (3, (128,), 32, 40, 36, 1, False, False, False),
               (1, (512,), 512, 8, 8, 0, False, False, False), (2, (64,), 216, 16, 16, 0, False, False, False)]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_split_bf16x3(case):
    """Split-bf16 (hi*hi + hi*lo + lo*hi) 3x3 convolution: fp32-grade result on the bf16 matrix pipe."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_conv_split
    n, cins, cout, h, w, act, use_res, use_mul, ps = case
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=401)
    wt = _rand(cout, cin, 3, 3, seed=402, scale=1.0 / np.sqrt(cin * 9))
    b = _rand(cout, seed=403, scale=0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    if act == 1:
        want = F.relu(want)
    elif act == 2:
        want = F.leaky_relu(want, 0.1)
    if ps:
        want = F.pixel_shuffle(want, 2)
    res = _rand(*want.shape, seed=404) if use_res else None
    mul = torch.rand(want.shape[0], 1, want.shape[2], want.shape[3], generator=torch.Generator().manual_seed(405)) if use_mul else None
    if res is not None:
        want = want + res
    if mul is not None:
        want = want * mul
    srcs, off = [], 0
    for c in cins:
        srcs.append(_to_act(x[:, off:off + c], dev)); off += c
    pc = pack_conv(wt, b, dev, cins, pixel_shuffle=ps)
    pc.w16 = pack_conv_split(pc, wt, dev, pixel_shuffle=ps)
    kw = dict(residual=_to_act(res, dev) if use_res else None, pixmul=_to_act(mul, dev) if use_mul else None)
    out3 = ops.conv2d(srcs, pc, act, precision="bf16x3", **kw)
    torch.cuda.synchronize()
    _close(out3.nchw(), want.float(), tol=3e-5, what=f"bf16x3 {case}")
    out1 = ops.conv2d(srcs, pc, act, precision="bf16op", **kw)
    torch.cuda.synchronize()
    _close(out1.nchw(), want.float(), tol=2e-2, what=f"bf16 {case}")
    again = ops.conv2d(srcs, pc, act, precision="bf16x3", **kw)
    torch.cuda.synchronize()
    assert torch.equal(again.torch(), out3.torch())


SPLIT_GEMM_CASES = [(2, (64,), 64, 16, 32, 1, True), (1, (512,), 512, 8, 32, 0, True), (3, (256,), 96, 12, 32, 2, False),
                    (1, (64, 32), 200, 5, 37, 0, False), (2, (32,), 1024, 4, 32, 0, False)]


@pytest.mark.parametrize("case", SPLIT_GEMM_CASES)
def test_conv_split_1x1(case):
    """1x1 convolution / Linear on the split-bf16 kernel (GEMM form: a stage = two 16-channel sub-chunks)."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_conv_split
    n, cins, cout, h, w, act, use_res = case
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=431)
    wt, b = _rand(cout, cin, 1, 1, seed=432, scale=1.0 / np.sqrt(cin)), _rand(cout, seed=433, scale=0.1)
    want = F.conv2d(x.double(), wt.double(), b.double())
    want = F.relu(want) if act == 1 else (F.leaky_relu(want, 0.1) if act == 2 else want)
    res = _rand(*want.shape, seed=434) if use_res else None
    if res is not None:
        want = want + res
    srcs, off = [], 0
    for c in cins:
        srcs.append(_to_act(x[:, off:off + c], dev)); off += c
    pc = pack_conv(wt, b, dev, cins)
    pc.w16 = pack_conv_split(pc, wt, dev)
    out = ops.conv2d(srcs, pc, act, precision="bf16x3", residual=_to_act(res, dev) if use_res else None)
    torch.cuda.synchronize()
    _close(out.nchw(), want.float(), tol=3e-5, what=f"bf16x3 1x1 {case}")
    out1 = ops.conv2d(srcs, pc, act, precision="bf16op", residual=_to_act(res, dev) if use_res else None)
    torch.cuda.synchronize()
    _close(out1.nchw(), want.float(), tol=2e-2, what=f"bf16 1x1 {case}")


def test_split_batched_matmul_with_device_split_operand():
    """Attention-shaped products: per-image B = an activation, split + re-ordered by gpemsr_split_pack_rows."""
    from gpemsr_amd import ops
    dev = _dev()
    n, T, c = 3, 256, 64
    q = _rand(n, T, c, seed=441).to(dev)
    k = _rand(n, T, c, seed=442).to(dev)
    qa = ops.Act(q.contiguous(), n, T // 32, 32, c, c, 0)
    ka = ops.Act(k.contiguous(), n, T // 32, 32, c, c, 0)
    k16 = ops.split_pack_rows(ka)
    assert k16.shape == (n, 2, c // 16, 2, T, 8) and k16.dtype == torch.bfloat16
    # the packed planes reproduce k: hi + lo == k to 2^-16
    rec = (k16[:, 0].float() + k16[:, 1].float()).permute(0, 3, 1, 2, 4).reshape(n, T, c)
    assert float((rec - k).abs().max()) <= 2.0 ** -15 * float(k.abs().max())
    S = ops.conv2d([qa], ops.PackedConv(ka.buf, None, 1, T, (c,), 32, w16=k16), ops.ACT_NONE, weight_image_stride=T * c,
                   precision="bf16x3")
    torch.cuda.synchronize()
    want = torch.bmm(q.double(), k.double().transpose(1, 2)).float()
    got = S.buf.view(n, T, T)
    assert float((got - want).abs().max()) <= 3e-5 * float(want.abs().max())
    Sf = ops.conv2d([qa], ops.PackedConv(ka.buf, None, 1, T, (c,), 32), ops.ACT_NONE, weight_image_stride=T * c)
    torch.cuda.synchronize()
    assert float((Sf.buf.view(n, T, T) - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("n,h,w,ld,off", [(2, 16, 24, 64, 0), (1, 37, 131, 96, 16), (1, 70, 63, 64, 0)])
def test_tap_sum_fp32_conv_64_to_1(n, h, w, ld, off):
    """gpemsr_conv_c64_cout1_f32: Conv2d(64 -> 1, 3x3) of the exact-fp32 path as tap partial products on v_mfma_f32_32x32x2_f32."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_cout1_taps_f32
    dev = _dev()
    x = _rand(n, 64, h, w, seed=500 + h)
    wt = _rand(1, 64, 3, 3, seed=501, scale=1.0 / 24); b = _rand(1, seed=502)
    base = _rand(n, 1, h, w, seed=503)
    pc = pack_conv(wt, b, dev)
    pc.wtap32 = pack_cout1_taps_f32(wt, dev)
    xa = _to_act(x, dev, ld, off)
    want = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    got = ops.conv2d([xa], pc, 0, residual=_to_act(base, dev))
    _close(got.nchw(), (want + base.double()).float(), tol=3e-6, what="fp32 tap-sum 64 -> 1 with residual")
    got = ops.conv2d([xa], pc, 4)
    _close(got.nchw(), torch.sigmoid(torch.nn.functional.leaky_relu(want, 0.1)).float(), tol=3e-6, what="fp32 tap-sum mask head")


@pytest.mark.parametrize("n,h,w", [(2, 8, 12), (1, 19, 67), (1, 33, 125), (1, 1, 1)])
def test_upconv_out_composed_operator_fp32(n, h, w):
    """gpemsr_upconv_out_c64_f32: the decoder tail as one composed operator on the fp32 matrix pipe == the layered fp64 evaluation."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_upconv_out_f32
    dev = _dev()
    x = _rand(n, 64, h, w, seed=510 + h)
    w1 = _rand(64, 64, 3, 3, seed=511, scale=1.0 / 12); b1 = _rand(64, seed=512, scale=0.5)
    w2 = _rand(1, 64, 3, 3, seed=513, scale=1.0 / 24); b2 = _rand(1, seed=514)
    want = torch.nn.functional.conv2d(torch.nn.functional.conv_transpose2d(x.double(), w1.double(), b1.double(), stride=2, padding=1, output_padding=1),
                                      w2.double(), b2.double(), padding=1)
    frag, consts = pack_upconv_out_f32(w1, b1, w2, b2, dev)
    got = ops.upconv_out_f32(_to_act(x, dev), frag, consts)
    _close(got.nchw(), want.float(), tol=5e-6, what="composed up-block + output layer, fp32")


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (1, 37, 70), (3, 5, 9)])
def test_spynet_flow_update_row_sums_fp32(n, h, w):
    """gpemsr_conv7_c16_cout2_f32: Conv2d(16 -> 2, 7x7) + residual of the exact-fp32 path as MFMA row sums + vertical 7-sum."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_rowsum7_f32
    dev = _dev()
    x = _rand(n, 16, h, w, seed=600 + h)
    wt = _rand(2, 16, 7, 7, seed=601, scale=1.0 / 28); b = _rand(2, seed=602)
    up = _rand(n, 2, h, w, seed=603, scale=3.0)
    pc = pack_conv(wt, b, dev)
    pc.wrow7_32 = pack_rowsum7_f32(wt, dev)
    want = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), 1, 3) + up.double()
    got = ops.conv2d([_to_act(x, dev)], pc, 0, residual=_to_act(up, dev))
    _close(got.nchw(), want.float(), tol=3e-6, what="fp32 flow update with residual")


@pytest.mark.parametrize("n,cin,h,w,res", [(2, 32, 37, 70, False), (1, 32, 16, 16, True), (1, 32, 5, 9, False), (3, 32, 64, 64, True), (1, 64, 33, 40, False)])
def test_conv7_cout16_row_pair_form(n, cin, h, w, res):
    """gpemsr_conv2d with descriptor.transposed = 2 (packing.pack_rowpair7): Conv2d(cin -> 16, 7x7, pad 3) + bias + ReLU (+ residual)
    with output rows 2i and 2i+1 sharing one 32-row matrix tile == the plain kernel's result to rounding and the fp64 convolution to
    3e-6; odd heights (the last pair has no second row), maps narrower than a tile, several images."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_rowpair7
    dev = _dev()
    x = _rand(n, cin, h, w, seed=700 + h)
    wt = _rand(16, cin, 7, 7, seed=701, scale=1.0 / (7 * cin ** 0.5)); b = _rand(16, seed=702)
    r = _rand(n, 16, h, w, seed=703) if res else None
    pc = pack_conv(wt, b, dev)
    plain = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_RELU, residual=_to_act(r, dev) if res else None).nchw().clone()
    pc.wpair7 = pack_rowpair7(wt, dev)
    got = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_RELU, residual=_to_act(r, dev) if res else None)
    want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), 1, 3))
    if res:
        want = want + r.double()
    _close(got.nchw(), want.float(), tol=3e-6, what="row-pair 7x7 vs fp64")
    _close(got.nchw(), plain, tol=2e-6, what="row-pair 7x7 vs the plain kernel")


@pytest.mark.parametrize("n,cin,cout,h,w,act", [(2, 32, 64, 16, 64, 1), (1, 64, 32, 37, 130, 1), (3, 32, 32, 8, 64, 0), (1, 8, 64, 5, 70, 2), (4, 8, 32, 64, 64, 1), (2, 64, 64, 64, 128, 1),
                                                (1, 32, 64, 3, 200, 1)])
def test_conv7_winograd_row_form(n, cin, cout, h, w, act):
    """gpemsr_conv2d with descriptor.transposed = 4 (csrc/conv7_wino.hip, packing.pack_winograd7): Conv2d(cin -> cout, 7x7, pad 3) + bias +
    activation in the 1-D Winograd F(2, 7) form (SpyNet's 32 -> 64 / 64 -> 32 layers, basicsr BasicModule via R:model/GPEMSR.py:67,98-100)
    == the float64 convolution to 2e-5 of the result and the direct fp32 kernel to the same; ragged heights (not a multiple of the 4-row
    tile) and widths (not a multiple of 64), both cout forms (32 and 64 per workgroup), one / eight chunks, run-to-run bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd7
    dev = _dev()
    x = _rand(n, cin, h, w, seed=900 + h)
    wt = _rand(cout, cin, 7, 7, seed=901, scale=1.0 / (7 * cin ** 0.5)); b = _rand(cout, seed=902)
    pc = pack_conv(wt, b, dev)
    pc.wino7 = pack_winograd7(wt, dev)
    xa = _to_act(x, dev)
    assert ops.winograd7_ok([xa], pc)
    direct = ops.conv2d([xa], pc, act, direct7=True).nchw().clone()
    got = ops.conv2d([xa], pc, act)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 3)
    want = {0: want, 1: torch.relu(want), 2: F.leaky_relu(want, 0.1)}[act]
    _close(got.nchw(), want.float(), tol=2e-5, what="F(2,7) row form vs fp64")
    _close(got.nchw(), direct, tol=2e-5, what="F(2,7) row form vs the direct kernel")
    again = ops.conv2d([xa], pc, act)
    assert torch.equal(got.buf, again.buf)


@pytest.mark.parametrize("n,cin,cout,h,w,act", [(2, 32, 64, 16, 64, 1), (1, 64, 32, 37, 130, 1), (3, 32, 32, 8, 16, 0), (1, 8, 64, 13, 70, 2), (4, 8, 32, 64, 64, 1), (2, 64, 64, 64, 128, 1),
                                                (1, 32, 64, 7, 200, 1), (2, 24, 64, 19, 45, 2), (3, 32, 64, 72, 200, 1), (5, 16, 32, 100, 90, 0), (2, 32, 16, 40, 64, 1), (3, 16, 48, 21, 50, 2), (4, 32, 16, 96, 160, 1)])
def test_conv7_winograd_2d_form(n, cin, cout, h, w, act):
    """gpemsr_conv2d with descriptor.transposed = 6 (csrc/conv7_wino2d.hip, packing.pack_winograd77): Conv2d(cin -> cout, 7x7, pad 3) + bias +
    activation in the 2-D Winograd F(2x2, 7x7) form (SpyNet's 32 -> 64 / 64 -> 32 layers, basicsr BasicModule via R:model/GPEMSR.py:67,98-100)
    == the float64 convolution to 4e-5 of the result and the direct fp32 kernel to the same; ragged heights / widths (not multiples of the
    8 x 16 tile), one to eight chunks (the raw-image ring wraps beyond three), one or two 32-cout blocks or one to three 16-cout blocks (the
    16x16x4 MFMA form of SpyNet's 32 -> 16 layers), up to four tiles per persistent workgroup, a strided output slice, run-to-run
    bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd77
    dev = _dev()
    x = _rand(n, cin, h, w, seed=950 + h)
    wt = _rand(cout, cin, 7, 7, seed=951, scale=1.0 / (7 * cin ** 0.5)); b = _rand(cout, seed=952)
    pc = pack_conv(wt, b, dev)
    pc.wino77 = pack_winograd77(wt, dev)
    assert tuple(pc.wino77.shape) == (cin // 8, 64, 2, cout, 4)
    xa = _to_act(x, dev)
    assert ops.winograd77_ok([xa], pc, act=act)
    direct = ops.conv2d([xa], pc, act, direct7=True).nchw().clone()
    nm = []
    got = ops.conv2d([xa], pc, act)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 3)
    want = {0: want, 1: torch.relu(want), 2: F.leaky_relu(want, 0.1)}[act]
    _close(got.nchw(), want.float(), tol=4e-5, what="F(2x2,7x7) form vs fp64")
    _close(got.nchw(), direct, tol=4e-5, what="F(2x2,7x7) form vs the direct kernel")
    again = ops.conv2d([xa], pc, act)
    assert torch.equal(got.buf, again.buf)
    # into a channel slice of a wider buffer (the neighbouring channels stay untouched)
    wide = ops.Act(torch.full((n, h, w, cout + 16), 7.0, device=dev), n, h, w, cout, cout + 16, 8)
    ops.conv2d([xa], pc, act, out=wide)
    assert torch.equal(wide.nchw(), got.nchw())
    assert float((wide.buf[..., :8] - 7.0).abs().max()) == 0.0 and float((wide.buf[..., cout + 8:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("n,cin,cout,k,h,w", [(2, 64, 64, 3, 37, 70), (1, 128, 256, 3, 20, 36), (3, 64, 512, 1, 16, 16), (1, 32, 128, 3, 64, 64), (2, 64, 32, 3, 9, 50)])
def test_groupnorm_statistics_from_the_fp32_conv_epilogue(n, cin, cout, k, h, w):
    """gpemsr_conv_desc.gn_partials: the conv output is unchanged (bit for bit) and groupnorm_relu on it -- now finish + apply, no
    statistics pass -- equals GroupNorm(32, eps 1e-6)(conv) + ReLU in fp64 to 1e-5 and the statistics-pass result to 2e-6."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    dev = _dev()
    x = _rand(n, cin, h, w, seed=800 + h)
    wt = _rand(cout, cin, k, k, seed=801, scale=1.0 / (k * cin ** 0.5)); b = _rand(cout, seed=802)
    g = (1.0 + 0.2 * _rand(cout, seed=803)).to(dev); be = (0.2 * _rand(cout, seed=804)).to(dev)
    pc = pack_conv(wt, b, dev)
    plain = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_NONE)
    assert plain.gn is None
    y0 = ops.groupnorm_relu(plain, g, be, True).nchw().clone()
    got = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_NONE, gn_stats=True)
    assert got.gn is not None, "this shape should take the lean epilogue"
    assert torch.equal(got.nchw(), plain.nchw())
    y1 = ops.groupnorm_relu(got, g, be, True)
    assert got.gn is None
    conv = F.conv2d(x.double(), wt.double(), b.double(), 1, k // 2)
    want = torch.relu(F.group_norm(conv, 32, g.double().cpu(), be.double().cpu(), 1e-6))
    _close(y1.nchw(), want.float(), tol=1e-5, what="GN from epilogue sums vs fp64")
    _close(y1.nchw(), y0, tol=2e-6, what="GN from epilogue sums vs the statistics pass")


@pytest.mark.parametrize("sym,group", [("1", "4"), ("0", "1"), ("0", "2"), ("1", "1")])
def test_winograd_launch_variants_are_bit_identical(sym, group, monkeypatch):
    """The A/B switches of the wide Winograd kernel -- GPEMSR_WINO_SYM (all eight waves issue the LDS-DMA vs one wave per SIMD) and
    GPEMSR_WINO_TN_GROUP (cout blocks of a pixel tile that are neighbours in the launch order) -- change who loads and in which order the
    workgroups run, never the arithmetic: results equal the default launch bit for bit (512 couts = 8 cout blocks, ragged tiles, 3 images)."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd
    dev = _dev()
    x = _rand(3, 128, 19, 45, seed=31)
    wt = _rand(512, 128, 3, 3, seed=32, scale=1.0 / np.sqrt(128 * 9))
    pc = pack_conv(wt, _rand(512, seed=33, scale=0.1), dev)
    pc.wino = pack_winograd(wt, dev)
    monkeypatch.delenv("GPEMSR_WINO_SYM", raising=False); monkeypatch.delenv("GPEMSR_WINO_TN_GROUP", raising=False)
    want = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_RELU, winograd=True).nchw().clone()
    monkeypatch.setenv("GPEMSR_WINO_SYM", sym); monkeypatch.setenv("GPEMSR_WINO_TN_GROUP", group)
    got = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_RELU, winograd=True).nchw()
    assert torch.equal(got, want)


@pytest.mark.parametrize("n,cins,cout,h,w,mode", [(3, (64,), 64, 37, 70, "res"), (40, (64,), 64, 64, 96, "plain"), (2, (64, 128, 64), 128, 19, 45, "plain"),
                                                  (6, (64,), 256, 40, 64, "ps"), (9, (64,), 64, 48, 64, "gn"), (5, (64,), 64, 32, 64, "cos")])
def test_persistent_winograd_kernel_equals_the_one_tile_kernel(n, cins, cout, h, w, mode, monkeypatch):
    """conv_wino2p_f32_kernel (csrc/conv_wino_p.hip: workgroups walk tiles, the next tile's first chunks land under the current tile's last two
    stages and its epilogue; chunk counts = 2 mod 3, i.e. the 64- and 256-channel layers) against conv_wino2_f32_kernel (GPEMSR_WINO_PERSIST=0):
    bit for bit -- same arithmetic, only who waits for what changes -- for the plain / residual + multiplier / PixelShuffle stores, the
    GroupNorm partial sums and the patch-cosine sums; more tiles than workgroups (40 x 8 x 3 = 960), ragged tiles, three sources."""
    import ctypes as C
    from gpemsr_amd import _abi, ops
    from gpemsr_amd.packing import pack_conv, pack_winograd
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=41)
    wt = _rand(cout, cin, 3, 3, seed=42, scale=1.0 / np.sqrt(cin * 9)); b = _rand(cout, seed=43, scale=0.1)
    ps = mode == "ps"
    pc = pack_conv(wt, b, dev, cins, pixel_shuffle=ps)
    pc.wino = pack_winograd(wt, dev, pixel_shuffle=ps)
    srcs, o = [], 0
    for c in cins:
        srcs.append(_to_act(x[:, o:o + c], dev)); o += c
    kw = {}
    if mode == "res":
        kw = dict(residual=_to_act(_rand(n, cout, h, w, seed=44), dev), pixmul=ops.Act((_rand(n, 1, h, w, seed=45).abs() + 0.5).reshape(-1).to(dev), n, h, w, 1, 1, 0))
    elif mode == "gn":
        kw = dict(gn_stats=True)
    elif mode == "cos":
        kw = dict(cos_with=_to_act(_rand(n, cout, h, w, seed=46).abs(), dev))

    def run():
        r = ops.conv2d(srcs, pc, ops.ACT_NONE if mode == "gn" else ops.ACT_RELU, winograd=True, **kw)
        extra = r.gn[0].clone() if mode == "gn" else None
        return r.nchw().clone(), extra
    monkeypatch.setenv("GPEMSR_WINO_PERSIST", "0")
    want, want_gn = run()
    monkeypatch.setenv("GPEMSR_WINO_PERSIST", "2")          # (2: every eligible chunk count; the default takes the 8-chunk layers only)
    got, got_gn = run()
    assert torch.equal(got, want), float((got - want).abs().max())
    if mode == "gn":
        assert torch.equal(got_gn, want_gn)
    # ... and the library really picked the persistent kernel for this descriptor
    d = _abi.ConvDesc()
    d.n, d.h, d.w, d.nsrc = n, h, w, len(srcs)
    for i, s_ in enumerate(srcs):
        d.src[i].ptr, d.src[i].ld, d.src[i].c = s_.ptr, s_.ld, s_.c
        d.src_image_stride[i] = -1
    d.cout, d.ksize, d.stride, d.transposed, d.weight, d.out, d.out_ld = cout, 3, 1, 3, pc.wino.data_ptr(), srcs[0].ptr, cout
    buf = C.create_string_buffer(160)
    assert _abi.load().gpemsr_conv2d_kernel_name(C.byref(d), buf, 160) == 0 and buf.value.decode() == "conv_wino2p_f32_kernel", buf.value


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 37, 70), (1, 128, 256, 20, 36), (1, 512, 512, 16, 16), (3, 128, 128, 8, 32), (1, 64, 64, 1, 1)])
def test_groupnorm_statistics_from_the_winograd_epilogue(n, cin, cout, h, w):
    """gpemsr_conv_desc.gn_partials with transposed = 3 (csrc/conv_wino.hip, wide kernel): the conv output equals the Winograd launch without
    partial sums bit for bit, one record per 8 x 32 tile, and groupnorm_relu on it (finish + apply, no statistics pass) equals
    GroupNorm(32, eps 1e-6)(conv) + ReLU in fp64 to 2e-5 and the statistics-pass result on the same conv output to 2e-6."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd
    dev = _dev()
    x = _rand(n, cin, h, w, seed=820 + h)
    wt = _rand(cout, cin, 3, 3, seed=821, scale=1.0 / (3 * cin ** 0.5)); b = _rand(cout, seed=822)
    g = (1.0 + 0.2 * _rand(cout, seed=823)).to(dev); be = (0.2 * _rand(cout, seed=824)).to(dev)
    pc = pack_conv(wt, b, dev)
    pc.wino = pack_winograd(wt, dev)
    plain = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_NONE, winograd=True)
    assert plain.gn is None
    y0 = ops.groupnorm_relu(plain, g, be, True).nchw().clone()
    got = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_NONE, gn_stats=True, winograd=True)
    assert got.gn is not None and got.gn[1] == -(-h // 8) * -(-w // 32)
    assert torch.equal(got.nchw(), plain.nchw())
    sums = got.gn[0].clone()
    y1 = ops.groupnorm_relu(got, g, be, True)
    assert got.gn is None
    conv = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    want = torch.relu(F.group_norm(conv, 32, g.double().cpu(), be.double().cpu(), 1e-6))
    _close(y1.nchw(), want.float(), tol=2e-5, what="GN from Winograd epilogue sums vs fp64")
    _close(y1.nchw(), y0, tol=2e-6, what="GN from Winograd epilogue sums vs the statistics pass")
    again = ops.conv2d([_to_act(x, dev)], pc, ops.ACT_NONE, gn_stats=True, winograd=True)
    assert torch.equal(again.gn[0], sums), "partial sums not bit-stable"


@pytest.mark.parametrize("n,h,w", [(2, 64, 64), (1, 128, 96), (3, 16, 32), (1, 48, 160)])
def test_patch_cosine_from_the_conv_epilogue(n, h, w):
    """gpemsr_conv_desc.cos_partials + gpemsr_patch_cosine_finish (R:model/GPEMSR.py:387-395): the cosine of relu(conv(t)) against a second
    64-channel map over 16x16 patches, without storing relu(conv(t)) -- equal to patch_cosine of the stored maps to 2e-6 and to the fp64
    value to 1e-5; the operand tensor is left untouched."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    dev = _dev()
    t = torch.relu(_rand(n, 64, h, w, seed=900 + h))
    a = torch.relu(_rand(n, 64, h, w, seed=901 + w) + 0.3)
    wt = _rand(64, 64, 3, 3, seed=902, scale=1.0 / 24); b = _rand(64, seed=903, scale=0.1)
    pc = pack_conv(wt, b, dev)
    ta, aa = _to_act(t, dev), _to_act(a, dev)
    assert ops.conv_cosine_ok(ta, pc)
    fb = ops.conv2d([ta], pc, ops.ACT_RELU)
    want32 = ops.patch_cosine(aa, fb).nchw().clone()
    a_before = aa.nchw().clone()
    got = ops.conv2d([ta], pc, ops.ACT_RELU, cos_with=aa)
    assert (got.n, got.h, got.w, got.c) == (n, h // 16, w // 16, 1)
    assert torch.equal(aa.nchw(), a_before)
    f64 = torch.relu(F.conv2d(t.double(), wt.double(), b.double(), 1, 1))
    def patches(z):
        return z.reshape(n, 64, h // 16, 16, w // 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(n, h // 16, w // 16, -1)
    pa, pb = patches(a.double()), patches(f64)
    want = ((pa * pb).sum(-1) / (pa.norm(dim=-1).clamp_min(1e-12) * pb.norm(dim=-1).clamp_min(1e-12))).unsqueeze(1)
    _close(got.nchw(), want.float(), tol=1e-5, what="epilogue patch cosine vs fp64")
    _close(got.nchw(), want32, tol=2e-6, what="epilogue patch cosine vs the stored-map kernel")
    # the same sums from the Winograd form's epilogue (csrc/conv_wino.hip; width % 32 == 0 as for the direct form)
    from gpemsr_amd.packing import pack_winograd
    pc.wino = pack_winograd(wt, dev)
    gw = ops.conv2d([ta], pc, ops.ACT_RELU, cos_with=aa, winograd=True)
    assert torch.equal(aa.nchw(), a_before)
    _close(gw.nchw(), want.float(), tol=1e-5, what="Winograd epilogue patch cosine vs fp64")
    assert torch.equal(gw.buf, ops.conv2d([ta], pc, ops.ACT_RELU, cos_with=aa, winograd=True).buf), "not bit-stable"
    # ... and from the F(4x4) form's (csrc/conv_wino4.hip, W4_COS)
    from gpemsr_amd.packing import pack_winograd4
    pc.wino4 = pack_winograd4(wt, dev)
    g4 = ops.conv2d([ta], pc, ops.ACT_RELU, cos_with=aa, winograd=True)
    assert torch.equal(aa.nchw(), a_before) and not torch.equal(g4.buf, gw.buf)          # (another summation order: the F(4x4) kernel ran)
    _close(g4.nchw(), want.float(), tol=2e-5, what="F(4x4) epilogue patch cosine vs fp64")
    assert torch.equal(g4.buf, ops.conv2d([ta], pc, ops.ACT_RELU, cos_with=aa, winograd=True).buf), "not bit-stable"


WINO_CASES = [
    # (n, cins, cout, h, w, act, residual, pixmul)
    (2, (64,), 64, 16, 32, 1, False, False),            # exactly one 16 x 32 tile per image
    (3, (64,), 64, 37, 70, 0, True, True),              # ragged tiles both ways, odd height (a 2x2 block across the bottom edge), residual + multiplier
    (1, (8,), 32, 5, 5, 2, False, False),               # one chunk, one block row
    (2, (64, 128, 64), 64, 17, 33, 0, False, False),    # concat of three sources, odd sizes
    (1, (512,), 512, 24, 40, 1, False, False),          # 64 chunks, 16 cout blocks (the VQGAN layers)
    (5, (128,), 256, 48, 64, 2, True, False),           # several images x tiles x cout blocks
    (1, (64,), 64, 1, 1, 0, False, False),              # a single pixel
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd_form(case):
    """gpemsr_conv_desc.transposed = 3: the Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolution (csrc/conv_wino.hip) against
    torch in float64 (tolerance 2e-5 of the result's scale: fp32 transforms of both operands; the direct kernel sits at ~1e-6), against
    the direct kernel, source / output / residual embedded in wider buffers, and run-to-run bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd
    n, cins, cout, h, w, act, use_res, use_mul = case
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=11)
    wt = _rand(cout, cin, 3, 3, seed=12, scale=1.0 / np.sqrt(cin * 9))
    b = _rand(cout, seed=13, scale=0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    if act == 1:
        want = F.relu(want)
    elif act == 2:
        want = F.leaky_relu(want, 0.1)
    res = _rand(n, cout, h, w, seed=14) if use_res else None
    mul = _rand(n, 1, h, w, seed=15).abs() + 0.5 if use_mul else None
    if res is not None:
        want = want + res.double()
    if mul is not None:
        want = want * mul.double()
    srcs, o = [], 0
    for i, c in enumerate(cins):
        srcs.append(_to_act(x[:, o:o + c], dev, ld=c + (8 if i == 0 else 0), off=8 if i == 0 else 0))
        o += c
    pc = pack_conv(wt, b, dev, cins)
    pc.wino = pack_winograd(wt, dev)
    assert ops.winograd_ok(srcs, pc)
    r_act = None if res is None else _to_act(res, dev, ld=cout + 4, off=4)
    m_act = None if mul is None else ops.Act(mul.reshape(-1).to(dev), n, h, w, 1, 1, 0)
    out = _to_act(torch.zeros(n, cout, h, w), dev, ld=cout + 12, off=4)
    got = ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act, out=out, winograd=True)
    _close(got.nchw(), want, 2e-5, f"winograd {case}")
    assert float((out.buf.view(n, h, w, cout + 12)[..., :4] - 7.0).abs().max()) == 0.0       # neighbours of the slice untouched
    direct = ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act)
    _close(got.nchw(), direct.nchw(), 2e-5, "winograd vs direct")
    again = _to_act(torch.zeros(n, cout, h, w), dev, ld=cout + 12, off=4)
    ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act, out=again, winograd=True)
    assert torch.equal(out.buf, again.buf), "not bit-stable run to run"


WINO4_CASES = [
    # (n, cins, cout, h, w, act, residual, pixmul)
    (1, (128,), 64, 16, 32, 0, False, False),   # exactly one 16 x 32 tile, 16 chunks
    (2, (128,), 128, 37, 70, 1, False, False),  # ragged tiles both ways, odd sizes (blocks across the right / bottom edge)
    (1, (512,), 512, 32, 64, 0, False, False),  # the VQGAN 512-channel layers: 64 chunks, 8 cout blocks
    (2, (64, 128, 64), 64, 17, 33, 2, False, False),   # concat of three sources
    (3, (256,), 128, 48, 40, 0, False, False),  # several images x tiles x cout blocks
    (1, (8,), 64, 20, 36, 0, False, False),     # ONE chunk (prologue only)
    (1, (16,), 64, 16, 32, 0, False, False),    # two chunks (no image DMA inside the loop)
    (2, (24,), 64, 21, 40, 1, False, False),    # three chunks (the ring of three raw images filled by the prologue)
    (1, (32,), 128, 33, 65, 0, False, False),   # four chunks: the first DMA issued inside the loop
    (2, (64,), 64, 32, 64, 0, True, False),     # ResidualBlockNoBN conv2: + residual, whole tiles
    (3, (64,), 64, 37, 70, 2, True, True),      # + residual and the per-pixel multiplier (MPF mask), ragged tiles
]


@pytest.mark.parametrize("case", WINO4_CASES)
def test_conv2d_winograd_f4x4_form(case):
    """gpemsr_conv_desc.transposed = 5: the Winograd F(4x4, 3x3) form of the many-channel 3x3 stride-1 layers (csrc/conv_wino4.hip) against
    torch in float64 (tolerance 6e-5 of the result's scale: the transforms' constants reach 8, fp32 throughout; F(2x2) sits at 2e-5), against
    the direct kernel, sources / output embedded in wider buffers, the GroupNorm partial sums of its epilogue against the statistics pass,
    and run-to-run bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd, pack_winograd4
    n, cins, cout, h, w, act, use_res, use_mul = case
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=31)
    wt = _rand(cout, cin, 3, 3, seed=32, scale=1.0 / np.sqrt(cin * 9))
    b = _rand(cout, seed=33, scale=0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    if act == 1:
        want = F.relu(want)
    elif act == 2:
        want = F.leaky_relu(want, 0.1)
    srcs, o = [], 0
    for i, c in enumerate(cins):
        srcs.append(_to_act(x[:, o:o + c], dev, ld=c + (8 if i == 0 else 4 if i == 1 else 0), off=8 if i == 0 else 0))   # (row pitches = 0 and 4 mod 8 floats)
        o += c
    pc = pack_conv(wt, b, dev, cins)
    pc.wino = pack_winograd(wt, dev)
    pc.wino4 = pack_winograd4(wt, dev)
    res = _rand(n, cout, h, w, seed=36) if use_res else None
    mul = _rand(n, 1, h, w, seed=37).abs() + 0.5 if use_mul else None
    if res is not None:
        want = want + res.double()
    if mul is not None:
        want = want * mul.double()
    r_act = None if res is None else _to_act(res, dev, ld=cout + 4, off=4)
    m_act = None if mul is None else ops.Act(mul.reshape(-1).to(dev), n, h, w, 1, 1, 0)
    assert ops.winograd_ok(srcs, pc, residual=r_act) and ops.winograd4_ok(srcs, pc, r_act, m_act)
    out = _to_act(torch.zeros(n, cout, h, w), dev, ld=cout + 12, off=4)
    got = ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act, out=out, winograd=True)
    _close(got.nchw(), want, 6e-5, f"winograd F(4x4) {case}")
    assert float((out.buf.view(n, h, w, cout + 12)[..., :4] - 7.0).abs().max()) == 0.0       # neighbours of the slice untouched
    direct = ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act)
    _close(got.nchw(), direct.nchw(), 6e-5, "winograd F(4x4) vs direct")
    f2 = ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act, winograd=True, winograd4=False)
    assert not torch.equal(f2.nchw(), got.nchw()), "the F(4x4) kernel did not run"
    again = _to_act(torch.zeros(n, cout, h, w), dev, ld=cout + 12, off=4)
    ops.conv2d(srcs, pc, act, residual=r_act, pixmul=m_act, out=again, winograd=True)
    assert torch.equal(out.buf, again.buf), "not bit-stable run to run"
    if act == 0 and cout % 32 == 0 and not use_res:
        g = (1.0 + 0.2 * _rand(cout, seed=34)).to(dev); be = (0.2 * _rand(cout, seed=35)).to(dev)
        plain = ops.conv2d(srcs, pc, ops.ACT_NONE, winograd=True)
        y0 = ops.groupnorm_relu(plain, g, be, True).nchw().clone()
        st = ops.conv2d(srcs, pc, ops.ACT_NONE, gn_stats=True, winograd=True)
        assert st.gn is not None and st.gn[1] == -(-h // 16) * -(-w // 32)
        assert torch.equal(st.nchw(), plain.nchw())
        sums = st.gn[0].clone()
        y1 = ops.groupnorm_relu(st, g, be, True)
        wantg = torch.relu(F.group_norm(F.conv2d(x.double(), wt.double(), b.double(), 1, 1), 32, g.double().cpu(), be.double().cpu(), 1e-6))
        _close(y1.nchw(), wantg.float(), tol=6e-5, what="GN from the F(4x4) epilogue sums vs fp64")
        _close(y1.nchw(), y0, tol=2e-6, what="GN from the F(4x4) epilogue sums vs the statistics pass")
        assert torch.equal(ops.conv2d(srcs, pc, ops.ACT_NONE, gn_stats=True, winograd=True).gn[0], sums), "partial sums not bit-stable"


@pytest.mark.parametrize("n,cins,cout,h,w,act,use_res", [(2, (64, 64), 216, 32, 64, 0, False), (1, (128,), 216, 37, 70, 2, True), (3, (64,), 72, 16, 32, 1, False)])
def test_conv2d_winograd_f4x4_form_with_padded_couts(n, cins, cout, h, w, act, use_res):
    """cout % 64 != 0 (the 216-channel offset convolutions of the DCN packs, R:model/GPEMSR.py:98-140 via basicsr DCNv2Pack): U carries zero rows
    up to the next multiple of 64 and the kernel stores only the real channels -- against fp64 and the direct kernel, neighbours of the output
    slice untouched."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd4
    dev = _dev()
    cin = sum(cins)
    x = _rand(n, cin, h, w, seed=51)
    wt = _rand(cout, cin, 3, 3, seed=52, scale=1.0 / np.sqrt(cin * 9)); b = _rand(cout, seed=53, scale=0.1)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    want = F.relu(want) if act == 1 else F.leaky_relu(want, 0.1) if act == 2 else want
    res = _rand(n, cout, h, w, seed=54) if use_res else None
    if res is not None:
        want = want + res.double()
    srcs, o = [], 0
    for c in cins:
        srcs.append(_to_act(x[:, o:o + c], dev)); o += c
    pc = pack_conv(wt, b, dev, cins)
    pc.wino4 = pack_winograd4(wt, dev)
    assert tuple(pc.wino4.shape) == (cin // 8, 36, 2, -(-cout // 64) * 64, 4) and not ops.winograd_ok(srcs, pc) and ops.winograd4_padded_ok(srcs, pc)
    r_act = None if res is None else _to_act(res, dev, ld=cout + 4, off=4)
    out = _to_act(torch.zeros(n, cout, h, w), dev, ld=cout + 12, off=4)
    got = ops.conv2d(srcs, pc, act, residual=r_act, out=out, winograd=True)
    _close(got.nchw(), want, 6e-5, "padded-cout F(4x4) vs fp64")
    buf = out.buf.view(n, h, w, cout + 12)
    assert float((buf[..., :4] - 7.0).abs().max()) == 0.0 and float((buf[..., 4 + cout:] - 7.0).abs().max()) == 0.0      # nothing stored beside the slice
    direct = ops.conv2d(srcs, pc, act, residual=r_act)
    _close(got.nchw(), direct.nchw(), 6e-5, "padded-cout F(4x4) vs direct")
    assert not torch.equal(got.nchw(), direct.nchw())


@pytest.mark.parametrize("n,c,cout,h,w", [(2, 128, 128, 32, 64), (3, 64, 64, 37, 70), (1, 512, 512, 16, 32), (2, 256, 64, 48, 33)])
def test_groupnorm_relu_folded_into_the_f4x4_input_transform(n, c, cout, h, w):
    """gpemsr_conv_desc.a_scale / a_shift with transposed = 5 (csrc/conv_wino4.hip, W4_AFF): conv(relu(GroupNorm(t))) without the normalised
    tensor in memory (the first Normalize + ReLU of a VQGAN ResidualBlock, R:model/blocks.py:5-29) -- against fp64, against apply + convolution,
    with ragged edge tiles (padding must stay zero: relu(shift) is not), the GroupNorm sums of its own epilogue, bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd, pack_winograd4
    dev = _dev()
    t = _rand(n, c, h, w, seed=41) * 2.0 + 0.3
    g = (1.0 + 0.2 * _rand(c, seed=42)).to(dev); be = (0.2 * _rand(c, seed=43)).to(dev)
    wt = _rand(cout, c, 3, 3, seed=44, scale=1.0 / np.sqrt(c * 9)); b = _rand(cout, seed=45, scale=0.1)
    pc = pack_conv(wt, b, dev)
    pc.wino = pack_winograd(wt, dev)
    pc.wino4 = pack_winograd4(wt, dev)
    ta = _to_act(t, dev)
    assert ops.conv_affine_source_ok32(ta, pc)
    sc, sh = ops.groupnorm_scale_shift(ta, g, be)
    got = ops.conv2d([ta], pc, ops.ACT_NONE, winograd=True, gn_stats=True, a_affine=(sc, sh, True))
    assert got.gn is not None
    sums = got.gn[0].clone()
    norm64 = torch.relu(F.group_norm(t.double(), 32, g.double().cpu(), be.double().cpu(), 1e-6))
    want = F.conv2d(norm64, wt.double(), b.double(), 1, 1)
    _close(got.nchw(), want, 6e-5, "folded GroupNorm + ReLU -> F(4x4) conv vs fp64")
    tn = ops.groupnorm_relu(_to_act(t, dev), g, be, True)
    ref = ops.conv2d([tn], pc, ops.ACT_NONE, winograd=True, gn_stats=True)
    _close(got.nchw(), ref.nchw(), 2e-5, "folded vs apply + convolution")
    _close(sums, ref.gn[0], 2e-5, "GroupNorm sums of the folded launch")
    again = ops.conv2d([ta], pc, ops.ACT_NONE, winograd=True, gn_stats=True, a_affine=(sc, sh, True))
    assert torch.equal(again.nchw(), got.nchw()) and torch.equal(again.gn[0], sums), "not bit-stable"
    assert float((ta.nchw() - t.to(dev)).abs().max()) == 0.0                       # the source is left as stored


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (1, 13, 21)])
def test_conv2d_winograd_form_pixel_shuffle(n, h, w):
    """upconv1-3 (R:model/GPEMSR.py:304-316,442-448: conv 64 -> 256 + PixelShuffle(2) + LeakyReLU) in the Winograd form: the store map
    of the 64-cout kernel against torch's pixel_shuffle."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_winograd
    dev = _dev()
    x = _rand(n, 64, h, w, seed=21)
    wt = _rand(256, 64, 3, 3, seed=22, scale=1.0 / 24)
    b = _rand(256, seed=23, scale=0.1)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x.double(), wt.double(), b.double(), 1, 1), 2), 0.1)
    pc = pack_conv(wt, b, dev, pixel_shuffle=True)
    pc.wino = pack_winograd(wt, dev, pixel_shuffle=True)
    xa = _to_act(x, dev)
    assert ops.winograd_ok([xa], pc)
    got = ops.conv2d([xa], pc, 2, winograd=True)
    assert (got.n, got.h, got.w, got.c) == (n, 2 * h, 2 * w, 64)
    _close(got.nchw(), want, 2e-5, "winograd + pixel shuffle")
    _close(got.nchw(), ops.conv2d([xa], pc, 2).nchw(), 2e-5, "vs direct")
    from gpemsr_amd.packing import pack_winograd4
    pc.wino4 = pack_winograd4(wt, dev, pixel_shuffle=True)              # upconv1-3 on the F(4x4) form (csrc/conv_wino4.hip, W4_PS)
    assert ops.winograd4_ok([xa], pc)
    g4 = ops.conv2d([xa], pc, 2, winograd=True)
    assert (g4.n, g4.h, g4.w, g4.c) == (n, 2 * h, 2 * w, 64) and not torch.equal(g4.nchw(), got.nchw())
    _close(g4.nchw(), want, 6e-5, "winograd F(4x4) + pixel shuffle")
