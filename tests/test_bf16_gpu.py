"""bf16 data path (precision = "bf16", BASELINE configs[2]): every entry point of the "bf16 data path" section of
include/gpemsr_hip.h against torch CPU functional ops.  The comparison operands are the bf16-ROUNDED inputs and weights
evaluated in fp32 (so only accumulation order and the final bf16 rounding of the result differ): fp32 outputs must match
to 1e-4, bf16 outputs to 2^-8 relative (half an ulp is 2^-9)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = 2.0 ** -8


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda", 0)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _r(x):
    """round to bf16 and back (what the device tensor holds)"""
    return x.to(torch.bfloat16).to(torch.float32)


def _act16(x_nchw, dev, ld=None, off=0):
    from gpemsr_amd import ops
    n, c, h, w = x_nchw.shape
    ld = c if ld is None else ld
    buf = torch.full((n, h, w, ld), 7.0)
    buf[..., off:off + c] = x_nchw.permute(0, 2, 3, 1)
    return ops.Act(buf.to(torch.bfloat16).to(dev).contiguous().view(-1), n, h, w, c, ld, off)


def _act32(x_nchw, dev):
    from gpemsr_amd import ops
    return ops.from_nhwc(x_nchw.permute(0, 2, 3, 1).contiguous().to(dev))


def _close(got, want, tol, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(want.abs().max().item(), 1e-6)
    assert err <= tol * ref + 1e-6, f"{what}: max err {err:.3e} vs ref max {ref:.3e} (tol {tol:.1e})"


def _apply_act(t, act):
    return {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.1), 3: torch.sigmoid,
            4: lambda v: torch.sigmoid(F.leaky_relu(v, 0.1))}[act](t)


def _pc(wt, b, dev, splits=None, pixel_shuffle=False):
    from gpemsr_amd.packing import pack_conv, pack_conv_bf16
    pc = pack_conv(wt, b, dev, splits, pixel_shuffle=pixel_shuffle)
    pc.wb = pack_conv_bf16(wt, dev, splits, pixel_shuffle=pixel_shuffle)
    return pc


CASES = [
    # (n, cins, cout, k, stride, h, w, act, residual: 0 none / 1 bf16 / 2 fp32, pixmul, out_f32, variant)
    (2, (64,), 64, 3, 1, 16, 32, 1, 1, False, False, 0),
    (2, (64,), 64, 3, 1, 16, 32, 1, 1, False, False, 3),          # variant 3: the ring kernel instead of the weights-resident one
    (3, (64,), 64, 3, 1, 50, 70, 2, 1, True, False, 0),           # several tiles per workgroup, ragged edges
    (1, (64,), 256, 3, 1, 33, 40, 0, 0, False, False, 0),         # four cout slabs
    (1, (64,), 32, 3, 1, 20, 33, 1, 2, False, True, 0),           # half-filled slab, fp32 out + fp32 residual
    (1, (64,), 64, 3, 1, 37, 70, 2, 2, True, False, 0),
    (1, (64,), 64, 3, 1, 37, 70, 0, 0, False, True, 1),
    (2, (64, 64), 64, 3, 1, 17, 19, 0, 1, True, False, 0),
    (1, (64, 128, 64), 64, 3, 1, 16, 40, 0, 0, False, False, 0),
    (1, (64, 64, 64), 64, 3, 1, 9, 33, 2, 0, False, False, 0),
    (1, (128,), 256, 3, 1, 20, 36, 0, 1, False, False, 0),
    (1, (128,), 256, 3, 1, 20, 36, 1, 0, False, False, 1),
    (1, (128,), 256, 3, 1, 20, 36, 1, 0, False, True, 2),
    (1, (128,), 256, 3, 1, 20, 36, 1, 1, False, False, 4),         # variant 4: the 8-wave big tile without loader waves
    (2, (128, 64), 128, 3, 1, 35, 45, 2, 1, True, False, 0),      # loader-wave form: two sources, six chunks, ragged tiles
    (1, (256,), 256, 3, 1, 16, 32, 0, 0, False, False, 0),
    (1, (64,), 216, 3, 1, 16, 16, 0, 0, False, True, 0),
    (1, (32,), 32, 3, 1, 12, 50, 1, 0, False, False, 0),
    (1, (16,), 32, 3, 1, 12, 50, 1, 0, False, False, 0),
    (1, (48,), 64, 3, 1, 10, 34, 0, 0, False, False, 0),
    (2, (64,), 64, 3, 2, 32, 64, 2, 0, False, False, 0),
    (1, (256,), 512, 3, 2, 16, 32, 0, 0, False, False, 0),
    (1, (64, 64), 64, 3, 2, 18, 66, 0, 0, False, False, 0),
    (1, (16,), 32, 7, 1, 32, 40, 1, 0, False, False, 0),
    (1, (32,), 64, 7, 1, 16, 48, 1, 0, False, False, 0),
    (1, (64,), 32, 7, 1, 9, 37, 1, 0, False, False, 0),
    (1, (32,), 16, 7, 1, 12, 32, 1, 0, False, False, 0),
    (2, (512,), 512, 1, 1, 8, 16, 0, 1, False, False, 0),
    (1, (64, 128, 64), 64, 1, 1, 16, 16, 0, 0, False, False, 0),
    (1, (320,), 64, 1, 1, 12, 20, 2, 1, False, False, 0),
    (1, (576,), 64, 1, 1, 10, 13, 2, 0, False, False, 0),
    (1, (512,), 512, 1, 1, 8, 8, 0, 0, False, True, 0),
    (1, (64,), 20, 3, 1, 16, 16, 3, 0, False, True, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_bf16(case):
    from gpemsr_amd import ops
    n, cins, cout, k, stride, h, w, act, res_mode, use_mul, out_f32, variant = case
    dev = _dev()
    cin = sum(cins)
    x = _r(_rand(n, cin, h, w, seed=1))
    wt = _r(_rand(cout, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k)))
    b = _rand(cout, seed=3, scale=0.1)
    want = _apply_act(F.conv2d(x, wt, b, stride, k // 2), act)
    oh, ow = want.shape[2:]
    res = _rand(n, cout, oh, ow, seed=4) if res_mode else None
    if res_mode == 1:
        res = _r(res)
    mul = torch.rand(n, 1, oh, ow, generator=torch.Generator().manual_seed(5)) if use_mul else None
    if res is not None:
        want = want + res
    if mul is not None:
        want = want * mul
    srcs, o = [], 0
    for i, c in enumerate(cins):
        srcs.append(_act16(x[:, o:o + c], dev, ld=c + (8 if i == 0 else 0), off=8 if i == 0 else 0))
        o += c
    pc = _pc(wt, b, dev, cins)
    r_act = None if res is None else (_act16(res, dev) if res_mode == 1 else _act32(res, dev))
    m_act = None if mul is None else ops.Act(mul.reshape(-1).to(dev), n, oh, ow, 1, 1, 0)
    got = ops.conv2d(srcs, pc, act, stride=stride, residual=r_act, pixmul=m_act, precision="bf16", out_f32=out_f32, variant=variant,
                     force_mfma=True)
    assert got.bf16 == (not out_f32)
    _close(got.nchw(), want, 1e-4 if out_f32 else BF, str(case))


STREAM_CASES = [
    # more tiles than the chip holds workgroups (512): every workgroup streams several tiles, the DMA cursors cross tile
    # (and image, and cout-tile) boundaries while the previous tile is still being multiplied
    # (n, cins, cout, k, stride, h, w, transposed)
    (10, (64, 64), 64, 3, 1, 96, 96, False),        # 2 sources x 2 chunks, ring == 3 stages of a 12-stage tile
    (6, (32,), 64, 3, 1, 128, 96, False),           # ONE chunk per tile: single halo image, refilled at every tile end
    (3, (128,), 256, 3, 1, 96, 128, False),         # two cout tiles per pixel tile
    (8, (128,), 256, 3, 1, 96, 128, False),         # loader-wave form, three tiles per workgroup
    (9, (48,), 128, 3, 1, 80, 96, False),           # loader-wave form with 16-channel chunks (three of them)
    (12, (64,), 64, 3, 1, 64, 64, True),            # transposed: 4 stages per tile == ring depth
    (40, (64,), 64, 3, 1, 96, 80, True),            # transposed, weights-resident kernel: ~5 tiles per workgroup, ragged tile columns
    (10, (64,), 128, 3, 1, 72, 96, True),           # transposed, weights-resident kernel: two cout slabs
    (6, (128,), 64, 3, 1, 64, 96, True),            # transposed ring kernel, four chunks
    (5, (64,), 64, 1, 1, 192, 160, False),          # 1x1: two chunks, two stages per tile
    (6, (64,), 64, 3, 2, 192, 128, False),          # stride 2
    (4, (32,), 64, 7, 1, 128, 160, False),          # 7x7, single chunk
    (40, (64,), 128, 3, 1, 64, 96, False),          # weights-resident kernel: several tiles per workgroup and cout slab
]


@pytest.mark.parametrize("case", STREAM_CASES)
def test_conv2d_bf16_streams_many_tiles_per_workgroup(case):
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_convT, pack_convT_bf16
    n, cins, cout, k, stride, h, w, tr = case
    dev = _dev()
    cin = sum(cins)
    x = _r(_rand(n, cin, h, w, seed=1))
    b = _rand(cout, seed=3, scale=0.1)
    if tr:
        wt = _r(_rand(cin, cout, 3, 3, seed=2, scale=1.0 / np.sqrt(cin * 2.25)))
        want = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
        pc = pack_convT(wt, b, dev)
        pc.wb = pack_convT_bf16(wt, dev)
    else:
        wt = _r(_rand(cout, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k)))
        want = F.conv2d(x, wt, b, stride, k // 2)
        pc = _pc(wt, b, dev, cins)
    srcs, o = [], 0
    for c in cins:
        srcs.append(_act16(x[:, o:o + c], dev))
        o += c
    got = ops.conv2d(srcs, pc, 0, stride=stride, precision="bf16", force_mfma=True)
    _close(got.nchw(), want, BF, str(case))
    again = ops.conv2d(srcs, pc, 0, stride=stride, precision="bf16", force_mfma=True)
    assert torch.equal(got.buf, again.buf), "not bit-stable run to run"


def test_conv2d_bf16_dual_output_and_gn_partials():
    """out32 = the un-rounded result; gn partial sums + finish == the statistics torch's group_norm uses; apply == group_norm."""
    from gpemsr_amd import ops
    dev = _dev()
    for (cin, cout, h, w, variant) in ((64, 64, 24, 40, 0), (128, 256, 16, 48, 0), (128, 256, 16, 48, 1), (64, 128, 9, 33, 2)):
        n = 2
        x = _r(_rand(n, cin, h, w, seed=11))
        wt = _r(_rand(cout, cin, 3, 3, seed=12, scale=1.0 / np.sqrt(cin * 9)))
        b = _rand(cout, seed=13, scale=0.3)
        conv = F.conv2d(x, wt, b, 1, 1)
        pc = _pc(wt, b, dev)
        o32 = ops.new_act(n, h, w, cout, device=dev)
        got = ops.conv2d([_act16(x, dev)], pc, 0, precision="bf16", out32=o32, gn_stats=True, variant=variant)
        _close(o32.nchw(), conv, 1e-4, "out32")
        _close(got.nchw(), conv, BF, "bf16 out")
        assert got.gn is not None
        gamma, beta = _rand(cout, seed=14) + 1.5, _rand(cout, seed=15)
        res = _r(_rand(n, cout, h, w, seed=16))
        y = ops.groupnorm_relu(got, gamma.to(dev), beta.to(dev), True, residual=_act16(res, dev))
        want = F.relu(F.group_norm(conv, 32, gamma, beta, eps=1e-6)) + res
        # the normalised tensor is computed from the bf16-rounded conv output: 2^-8 of |conv| scaled by rstd*gamma
        _close(y.nchw(), want, 3 * BF, f"groupnorm after conv {cin}->{cout} v{variant}")
        # standalone statistics pass (no producing conv)
        z = ops.groupnorm_relu(_act16(_r(conv), dev), gamma.to(dev), beta.to(dev), False)
        _close(z.nchw(), F.group_norm(_r(conv), 32, gamma, beta, eps=1e-6), 2 * BF, "groupnorm standalone")


@pytest.mark.parametrize("n,c,cout,h,w", [(3, 64, 64, 24, 40), (2, 64, 64, 37, 70), (2, 128, 128, 20, 36), (1, 256, 256, 35, 33), (5, 512, 512, 16, 16),
                                          (80, 64, 64, 16, 32)])
def test_conv_with_folded_groupnorm_apply(n, c, cout, h, w):
    """Normalize + ReLU between the two convolutions of a VQGAN block (R:model/blocks.py:5-20) applied by the
    SECOND convolution while it stages its source (gpemsr_conv16_desc.a_scale / a_shift), so the normalised tensor never exists in HBM.
    Both kernel families that have the register-staged loader (64-channel weights-resident, wide 3x3 loader-wave tile; ragged tile
    edges, several tiles per workgroup, image changes inside a workgroup's stream) against (a) torch: conv(relu(group_norm(x))) on the
    bf16-rounded normalised tensor, and (b) the two-kernel path (gpemsr_groupnorm_apply_bf16, then the plain convolution), which
    it must reproduce to one bf16 rounding of the normalised tensor."""
    from gpemsr_amd import ops
    dev = _dev()
    x = _r(_rand(n, c, h, w, seed=91) * 2.0 + 0.3 * _rand(n, c, 1, 1, seed=92))
    gamma, beta = 1.0 + 0.5 * _rand(c, seed=93), 0.3 * _rand(c, seed=94)
    gamma[::7] *= -1.0                                   # negative scales too
    wt = _r(_rand(cout, c, 3, 3, seed=95, scale=1.0 / (3.0 * c ** 0.5)))
    b = _rand(cout, seed=96, scale=0.1)
    pc = _pc(wt, b, dev)
    xa = _act16(x, dev)
    assert ops.conv_affine_source_ok(xa, pc)
    sc, sh = ops.groupnorm_scale_shift(xa, gamma.to(dev), beta.to(dev))
    res = _r(_rand(n, cout, h, w, seed=97))
    got = ops.conv2d([xa], pc, 0, precision="bf16", a_affine=(sc, sh, True), gn_stats=True)
    torch.cuda.synchronize()
    y = _r(F.relu(F.group_norm(x, 32, gamma, beta, eps=1e-6)))
    want = F.conv2d(y, wt, b, 1, 1)
    _close(got.nchw(), want, 3 * BF, f"conv with folded GroupNorm apply {c}->{cout}")
    # the two-kernel path on the same input
    t = ops.groupnorm_relu(_act16(x, dev), gamma.to(dev), beta.to(dev), True)
    two = ops.conv2d([t], pc, 0, precision="bf16", gn_stats=True)
    _close(got.nchw(), two.nchw(), 2 * BF, "fused vs two kernels")
    # the statistics its epilogue leaves feed the block's second GroupNorm as before
    z = ops.groupnorm_relu(got, gamma[:cout].to(dev) if cout <= c else torch.ones(cout, device=dev), (beta[:cout].to(dev) if cout <= c else torch.zeros(cout, device=dev)),
                           True, residual=_act16(res, dev))
    g2, b2 = (gamma[:cout], beta[:cout]) if cout <= c else (torch.ones(cout), torch.zeros(cout))
    _close(z.nchw(), F.relu(F.group_norm(_r(want), 32, g2, b2, eps=1e-6)) + res, 4 * BF, "second GroupNorm from the epilogue sums")
    # without ReLU (the affine map alone) and with a residual + LeakyReLU epilogue
    got2 = ops.conv2d([xa], pc, 2, precision="bf16", a_affine=(sc, sh, False), residual=_act16(res, dev))
    y2 = _r(F.group_norm(x, 32, gamma, beta, eps=1e-6))
    _close(got2.nchw(), F.leaky_relu(F.conv2d(y2, wt, b, 1, 1), 0.1) + res, 3 * BF, "affine only + residual epilogue")


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 16, 32), (128, 64, 9, 20), (512, 256, 8, 8), (64, 32, 5, 33), (64, 64, 13, 45), (64, 128, 9, 20), (64, 64, 1, 1)])
def test_conv_transpose_bf16(cin, cout, h, w):
    """ConvTranspose2d(k3,s2,p1,op1) (R:model/blocks.py:32-38, R:model/GPEMSR.py:252-253) against torch; 64 -> 64k goes to the
    weights-resident kernel (ragged tiles, one pixel, two cout slabs), everything else to the ring kernel; all three activations."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_convT, pack_convT_bf16
    dev = _dev()
    x = _r(_rand(2, cin, h, w, seed=21))
    wt = _r(_rand(cin, cout, 3, 3, seed=22, scale=1.0 / np.sqrt(cin * 2.25)))
    b = _rand(cout, seed=23, scale=0.1)
    lin = F.conv_transpose2d(x, wt, b, stride=2, padding=1, output_padding=1)
    pc = pack_convT(wt, b, dev)
    pc.wb = pack_convT_bf16(wt, dev)
    for act, want in ((2, F.leaky_relu(lin, 0.1)), (0, lin), (1, F.relu(lin))):
        got = ops.conv2d([_act16(x, dev)], pc, act, precision="bf16")
        _close(got.nchw(), want, BF, f"convT act {act}")
    if cin == 64 and cout % 64 == 0:
        # the same layer on the ring kernel (variant 3 keeps it off the resident one): both must agree to bf16 rounding of the result
        ring = ops.conv2d([_act16(x, dev)], pc, 2, precision="bf16", variant=3)
        _close(ring.nchw(), F.leaky_relu(lin, 0.1), BF, "convT ring kernel")
        # a result written into a channel slice of a wider buffer (out_ld > cout)
        wide = ops.new_act(2, 2 * h, 2 * w, cout + 64, device=dev, bf16=True, zero=True)
        ops.conv2d([_act16(x, dev)], pc, 0, precision="bf16", out=wide.slice(64, cout))
        _close(wide.slice(64, cout).nchw(), lin, BF, "convT into a slice")
        assert float(wide.slice(0, 64).nchw().abs().max()) == 0.0


def test_pixel_shuffle_bf16():
    from gpemsr_amd import ops
    dev = _dev()
    x = _r(_rand(2, 64, 12, 34, seed=31))
    wt = _r(_rand(256, 64, 3, 3, seed=32, scale=1.0 / 24))
    b = _rand(256, seed=33, scale=0.1)
    want = F.leaky_relu(F.pixel_shuffle(F.conv2d(x, wt, b, 1, 1), 2), 0.1)
    got = ops.conv2d([_act16(x, dev)], _pc(wt, b, dev, pixel_shuffle=True), 2, precision="bf16")
    _close(got.nchw(), want, BF, "pixel shuffle")


def test_spynet_flow_conv_fp32_out_with_fp32_residual():
    """SpyNet's 16 -> 2 7x7 output conv: bf16 in, fp32 flow out, + the fp32 up-sampled flow (2-channel rows, scalar path)."""
    from gpemsr_amd import ops
    dev = _dev()
    x = _r(_rand(2, 16, 24, 40, seed=41))
    wt = _r(_rand(2, 16, 7, 7, seed=42, scale=1.0 / 28))
    b = _rand(2, seed=43, scale=0.1)
    up = _rand(2, 2, 24, 40, seed=44, scale=3.0)
    want = F.conv2d(x, wt, b, 1, 3) + up
    got = ops.conv2d([_act16(x, dev)], _pc(wt, b, dev), 0, residual=_act32(up, dev), precision="bf16", out_f32=True)
    _close(got.nchw(), want, 1e-4, "flow conv")


def test_attention_products_bf16():
    """The three products of model/blocks.py:75-80 with per-image B operands in the kpack layout: S = q.k^T, v^T = W_v.hn^T,
    A = P.v (+ v bias), against torch.bmm on the bf16-rounded operands."""
    from gpemsr_amd import ops
    dev = _dev()
    n, h, w, c = 2, 8, 12, 64
    T = h * w
    hn = _r(_rand(n, c, h, w, seed=51))
    wq, wk, wv = (_r(_rand(c, c, 1, 1, seed=s, scale=1.0 / 8)) for s in (52, 53, 54))
    bq, bk, bv = (_rand(c, seed=s, scale=0.1) for s in (55, 56, 57))
    hn_a = _act16(hn, dev)
    q = ops.conv2d([hn_a], _pc(wq, bq, dev), 0, precision="bf16")
    kp = ops.conv2d([hn_a], _pc(wk, bk, dev), 0, precision="bf16", kpack=True)
    q_ref = _r(F.conv2d(hn, wq, bq)).permute(0, 2, 3, 1).reshape(n, T, c)
    k_ref = _r(F.conv2d(hn, wk, bk)).permute(0, 2, 3, 1).reshape(n, T, c)
    # kpack layout [n][c/8][T][8]
    _close(kp.float().permute(0, 2, 1, 3).reshape(n, T, c), k_ref, 1e-6, "kpack of k")
    S = ops.conv2d([q], ops.PackedConv(None, None, 1, T, (c,), 32, wb=kp), 0, weight_image_stride=T * c, precision="bf16", out_f32=True)
    _close(S.torch().reshape(n, T, T), torch.bmm(q_ref, k_ref.transpose(1, 2)), 1e-4, "q.k^T")
    Sb = ops.conv2d([q], ops.PackedConv(None, None, 1, T, (c,), 32, wb=kp), 0, weight_image_stride=T * c, precision="bf16")
    P = ops.softmax_rows_bf16(Sb)
    p_ref = torch.softmax(_r(torch.bmm(q_ref, k_ref.transpose(1, 2))), dim=2)
    _close(P.torch().float().reshape(n, T, T), p_ref, 2 * BF, "softmax")
    hnp = ops.pack_rows_bf16(hn_a)
    wa = ops.Act(wv.reshape(c, c).to(torch.bfloat16).to(dev).contiguous().view(-1), n, 1, c, c, c, 0)
    vtp = ops.conv2d([wa], ops.PackedConv(None, None, 1, T, (c,), 32, wb=hnp), 0, weight_image_stride=T * c, src_image_stride=[0],
                     kpack=True, precision="bf16")
    v_ref = _r(F.conv2d(hn, wv)).permute(0, 2, 3, 1).reshape(n, T, c)                 # without bias
    _close(vtp.float().permute(0, 2, 1, 3).reshape(n, c, T), v_ref.transpose(1, 2), 1e-6, "v^T kpack")
    p_dev = _r(P.torch().float().reshape(n, T, T).cpu())
    out = ops.conv2d([P], ops.PackedConv(None, bv.to(dev), 1, c, (T,), 32, wb=vtp), 0, weight_image_stride=c * T, precision="bf16", out_f32=True)
    _close(out.torch().reshape(n, T, c), torch.bmm(p_dev, v_ref) + bv, 1e-4, "P.v")


@pytest.mark.parametrize("n,h,w,spike", [(3, 16, 16, False), (2, 32, 32, False), (9, 8, 16, False), (2, 16, 24, True), (1, 64, 64, True),
                                         (1, 144, 128, False)])        # 18,432 tokens: beyond the layered form's 16,384-column softmax
def test_flash_attention_bf16(n, h, w, spike):
    """the NonLocalBlock products + softmax (R:model/blocks.py:75-80) as ONE kernel, score matrix on chip.  Against
    torch on the bf16-rounded operands (fp32 softmax); operands come from the same kpack / perm16 producers the engine uses.  `spike`
    forces the cold path of the lazy online softmax (guide rule 26): one key far down the sequence scores ~60 above everything before
    it for some queries (its tile maximum outgrows the reference by more than 2^24), another EARLY key dominates other queries, so
    both "rescale O and l" and "later tiles underflow to zero" happen in one launch; every element is checked against the full
    reference.  n = 9 covers two image groups of the XCD mapping (images 8.. and the early exit of unused slots)."""
    from gpemsr_amd import ops
    dev = _dev()
    c, T = 512, h * w
    assert ops.flash_attention_ok(T, c)
    q = _r(_rand(n, T, c, seed=101, scale=0.25))
    k = _r(_rand(n, T, c, seed=102, scale=0.25))
    v = _r(_rand(n, T, c, seed=103, scale=2.0))
    bv = _rand(c, seed=104, scale=0.5)
    if spike:
        k[:, T - 37] = _r(12.0 * q[:, 5])            # late key: huge score for query 5 (and whatever correlates with it)
        k[:, 3] = _r(9.0 * q[:, 77])                 # early key: dominates query 77 from the first tile on
    def act(t):
        return ops.Act(t.reshape(n, h, w, c).to(torch.bfloat16).to(dev).contiguous().view(-1), n, h, w, c, c, 0)
    qa = act(q)
    kp = ops.pack_rows_bf16(act(k))                                     # [n][C/8][T][8]
    # v^T [n][T/8][C][8] in the perm16 key order: pack v^T's columns (= tokens) through the same permutation the engine's product applies
    pos = torch.arange(T)
    pp = pos & 15
    src = (pos & ~15) + 8 * ((pp & 7) >> 2) + 4 * (pp >> 3) + (pp & 3)
    vtp = v[:, src].reshape(n, T // 8, 8, c).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).to(dev)
    got = ops.flash_attention_bf16(qa, kp, vtp, bv.to(dev))
    torch.cuda.synchronize()
    ref_dev = torch.device("cuda") if T > 8192 else torch.device("cpu")            # (the long case: 2.7 GB of float64 scores)
    S = torch.bmm(q.double().to(ref_dev), k.double().to(ref_dev).transpose(1, 2))
    want = (torch.bmm(torch.softmax(S, dim=2), v.double().to(ref_dev)) + bv.double().to(ref_dev)).cpu()
    S = S.cpu() if spike else None
    assert torch.isfinite(got.torch().float()).all()
    _close(got.torch().float().reshape(n, T, c), want.float(), 2 * BF, "flash attention")
    if spike:
        assert float(S[:, 5].max()) - float(S[:, 5, :T - 64].max()) > 30.0          # the cold path really was needed
    # the engine's producers give the same operand: pack_rows(perm16) as the B operand of the v^T product
    hnp = ops.pack_rows_bf16(act(v), perm16=True)                        # rows = tokens of "hn" (here v itself), perm16 order
    assert torch.equal(hnp.cpu().float().permute(0, 2, 1, 3).reshape(n, T, c), _r(v)[:, src])


@pytest.mark.parametrize("cols", [9216, 16384])
def test_softmax_rows_bf16_long_rows(cols):
    """x8 at LR 192x192 gives 9216 latent tokens, x16 at LR 128x128 gives 16384 -- beyond the 8192 columns the bf16 row
    softmax used to hold in registers.  It now takes what the fp32 path takes (16384); longer rows are routed to the fp32 block by
    the engine (Engine.BF16_SOFTMAX_MAX_COLS)."""
    from gpemsr_amd import ops
    from gpemsr_amd.engine import Engine
    dev = _dev()
    assert cols <= Engine.BF16_SOFTMAX_MAX_COLS
    s = _r(_rand(3, cols, seed=71, scale=4.0))
    a = ops.Act(s.to(torch.bfloat16).to(dev).contiguous().view(-1), 1, 1, 3, cols, cols, 0)
    P = ops.softmax_rows_bf16(a)
    _close(P.torch().float().reshape(3, cols).cpu(), torch.softmax(s, dim=1), 2 * BF, "long-row softmax")
    with pytest.raises(RuntimeError):
        ops.softmax_rows_bf16(ops.Act(torch.zeros(2 * 16392, dtype=torch.bfloat16, device=dev), 1, 1, 2, 16392, 16392, 0))


def test_elementwise_bf16_kernels():
    from gpemsr_amd import ops
    dev = _dev()
    # bilinear (x2 up, x1/2 down, align_corners, mul) and the 3x3 stride-2 max|avg pool
    x = _r(_rand(2, 64, 9, 14, seed=61))
    a = _act16(x, dev)
    _close(ops.bilinear(a, 18, 28).nchw(), F.interpolate(x, size=(18, 28), mode="bilinear", align_corners=False), BF, "bilinear up")
    _close(ops.bilinear(a, 18, 28, mul=2.0).nchw(), 2 * F.interpolate(x, size=(18, 28), mode="bilinear", align_corners=False), BF, "bilinear mul")
    p = ops.pool3s2_maxavg(a).nchw()
    _close(p[:, :64], F.max_pool2d(x, 3, 2, 1), 1e-6, "maxpool")
    _close(p[:, 64:], F.avg_pool2d(x, 3, 2, 1), BF, "avgpool")
    # casts and channel copies
    y = _rand(2, 16, 5, 7, seed=62)
    _close(ops.cast_bf16(_act32(y, dev)).nchw(), _r(y), 1e-7, "cast to bf16")
    _close(ops.cast_f32(_act16(_r(y), dev)).nchw(), _r(y), 1e-7, "cast to f32")
    dst = ops.new_act(2, 5, 7, 64, device=dev, bf16=True, zero=True)
    ops.copy_channels(_act16(_r(y), dev), dst.slice(16, 16))
    ops.copy_channels_f32_bf16(_act32(y[:, :1], dev), dst.slice(33, 1))
    full = dst.nchw()
    _close(full[:, 16:32], _r(y), 1e-7, "copy slice")
    _close(full[:, 33:34], _r(y[:, :1]), 1e-7, "copy f32 -> bf16 slice")
    assert float(full[:, :16].abs().max()) == 0 and float(full[:, 34:].abs().max()) == 0
    # codebook gather
    table = _rand(50, 64, seed=63)
    idx = torch.randint(0, 50, (2 * 3 * 4,), generator=torch.Generator().manual_seed(64), dtype=torch.int32)
    g = ops.gather_rows_bf16(table.to(dev), idx.to(dev), 2, 3, 4)
    _close(g.torch().float().reshape(-1, 64), _r(table[idx.long()]), 1e-7, "gather rows")
    # image regrouping is a raw copy
    imgs = _act16(_r(_rand(6, 64, 4, 8, seed=65)), dev)
    gi = ops.gather_images(imgs, torch.tensor([5, 0, 0, 3], dtype=torch.int32, device=dev))
    _close(gi.nchw(), imgs.nchw()[[5, 0, 0, 3]], 0.0, "gather images")
    ci = ops.copy_images(imgs, 2, 1, 3, 2)
    _close(ci.nchw(), imgs.nchw()[[2, 5]], 0.0, "copy images")
    # ThreeDA pieces
    b, t, hh, ww, c = 2, 5, 6, 10, 64
    al = _r(_rand(b * t, c, hh, ww, seed=66)); em = _r(_rand(b * t, c, hh, ww, seed=67, scale=0.3)); er = _r(_rand(b, c, hh, ww, seed=68, scale=0.3))
    af = ops.temporal_gate(_act16(al, dev), _act16(em, dev), _act16(er, dev), b, t)
    cor = torch.sigmoid((em.view(b, t, c, hh, ww) * er.view(b, 1, c, hh, ww)).sum(2, keepdim=True))
    want_af = (al.view(b, t, c, hh, ww) * cor).reshape(b, t * c, hh, ww)
    _close(af.nchw(), want_af, BF, "temporal gate")
    m = _rand(t, t, seed=69, scale=0.5); mb = _rand(t, seed=70, scale=0.1)
    fm = ops.frame_mix_lrelu(_act16(_r(want_af), dev), t, m.to(dev), mb.to(dev))
    wf = F.leaky_relu(torch.einsum("ik,bkchw->bichw", m, _r(want_af).view(b, t, c, hh, ww)) + mb.view(1, t, 1, 1, 1), 0.1).reshape(b, t * c, hh, ww)
    _close(fm.nchw(), wf, BF, "frame mix")
    f5 = [_r(_rand(b, c, hh, ww, seed=71 + i)) for i in range(5)]
    tc = ops.threeda_combine(*[_act16(v, dev) for v in f5])
    _close(tc.nchw(), f5[0] * torch.sigmoid(f5[1]) * 2 + f5[2] + f5[3] + f5[4], BF, "threeda combine")
    # 16x16 patch cosine
    fa, fb = _r(_rand(2, 64, 32, 48, seed=80).abs()), _r(_rand(2, 64, 32, 48, seed=81).abs())
    pcs = ops.patch_cosine(_act16(fa, dev), _act16(fb, dev)).nchw()
    ua = F.unfold(fa, 16, stride=16); ub = F.unfold(fb, 16, stride=16)
    wantc = (F.normalize(ua, dim=1) * F.normalize(ub, dim=1)).sum(1).view(2, 1, 2, 3)
    _close(pcs, wantc, 1e-5, "patch cosine")


def test_stem_direct_and_dcn_bf16():
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv
    dev = _dev()
    # 1 -> 64 stem: fp32 image in, bf16 out
    img = torch.rand(2, 1, 20, 36, generator=torch.Generator().manual_seed(90))
    wt = _rand(64, 1, 3, 3, seed=91, scale=0.3); b = _rand(64, seed=92, scale=0.1)
    got = ops.conv2d([ops.Act(img.reshape(-1).to(dev), 2, 20, 36, 1, 1, 0)], pack_conv(wt, b, dev), 2, precision="bf16")
    assert got.bf16
    _close(got.nchw(), F.leaky_relu(F.conv2d(img, wt, b, 1, 1), 0.1), BF, "stem")
    # 64 -> 1 (bf16 in, fp32 out, fp32 residual, sigmoid(lrelu)) and flow down-convs (fp32 in -> bf16 slice; bf16 -> bf16 stride 2)
    x = _r(_rand(2, 64, 16, 24, seed=93))
    w1 = _rand(1, 64, 3, 3, seed=94, scale=1.0 / 24); b1 = _rand(1, seed=95)
    base = _rand(2, 1, 16, 24, seed=96)
    g1 = ops.conv2d([_act16(x, dev)], pack_conv(w1, b1, dev), 0, residual=_act32(base, dev), precision="bf16")
    assert not g1.bf16
    _close(g1.nchw(), F.conv2d(x, w1, b1, 1, 1) + base, 1e-4, "64 -> 1 with residual")
    g2 = ops.conv2d([_act16(x, dev)], pack_conv(w1, b1, dev), 4, precision="bf16")
    _close(g2.nchw(), torch.sigmoid(F.leaky_relu(F.conv2d(x, w1, b1, 1, 1), 0.1)), 1e-4, "mask head")
    flow = _rand(2, 2, 32, 48, seed=97, scale=2.0)
    wf = _rand(16, 2, 3, 3, seed=98, scale=0.3); bf = _rand(16, seed=99, scale=0.1)
    buf = ops.new_act(2, 8, 12, 64, device=dev, bf16=True, zero=True)
    ops.conv2d([_act32(flow, dev)], pack_conv(wf, bf, dev), 0, stride=4, out=buf.slice(16, 16), precision="bf16")
    want = F.conv2d(flow, wf, bf, 4, 1)
    _close(buf.nchw()[:, 16:32], want, BF, "flow conv stride 4 into a slice")
    w2 = _rand(16, 16, 3, 3, seed=100, scale=0.1)
    buf2 = ops.new_act(2, 4, 6, 64, device=dev, bf16=True, zero=True)
    ops.conv2d([buf.slice(16, 16)], pack_conv(w2, None, dev), 0, stride=2, out=buf2.slice(0, 16), precision="bf16")
    _close(buf2.nchw()[:, :16], F.conv2d(buf.nchw()[:, 16:32].cpu(), w2, None, 2, 1), BF, "flow conv stride 2 bf16 -> bf16")
    # deformable columns: bf16 features, fp32 offsets / mask logits; compare with the fp32 kernel on the same rounded features
    xf = _r(_rand(2, 64, 12, 16, seed=101))
    om = _rand(2, 216, 12, 16, seed=102, scale=2.0)
    c16 = ops.dcn_columns(_act16(xf, dev), _act32(om, dev), 8)
    c32 = ops.dcn_columns(_act32(xf, dev), _act32(om, dev), 8)
    assert c16.bf16 and not c32.bf16
    _close(c16.nchw(), c32.nchw(), BF, "dcn columns")


@pytest.mark.parametrize("geom", [(2, 12, 16, 64, 0, 0), (3, 8, 8, 64, 2, 2.0), (1, 7, 9, 72, 1, 12.0), (5, 32, 32, 64, 0, 1.0)])
def test_dcn_sampling_and_contraction_in_one_kernel(geom):
    """gpemsr_dcn_conv_bf16 (csrc/dcn_bf16.hip): deformable sampling into LDS + the 64 x 576 contraction == the oracle's restatement of
    basicsr DCNv2Pack -> torchvision deform_conv2d (R:model/GPEMSR.py:79-94) on the bf16-rounded features and weights, and == the
    two-kernel form (gpemsr_dcn_columns_bf16 + the 1x1 product) it replaces.  Ragged pixel counts (7 x 9 = 63 per image), a strided
    source, offsets far outside the image, all three activations."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv_bf16, pack_dcn, pack_dcn_rows_bf16
    from oracle import gpemsr_oracle as orc
    n, h, w, ld, act, big = geom
    dev = _dev()
    x = _r(_rand(n, 64, h, w, seed=300 + h))
    om = _rand(n, 216, h, w, seed=301 + h, scale=2.0)
    if big:
        om[:, :144] *= big                                  # sampling points up to +-24 pixels away: the out-of-image rule
    wt = _r(_rand(64, 64, 3, 3, seed=302, scale=0.05)); b = _rand(64, seed=303, scale=0.1)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    want = _apply_act(orc.deform_conv2d_v2(x, torch.cat((o1, o2), 1), torch.sigmoid(m), wt, b), act)
    pc = pack_dcn(wt, b, dev)
    pc.wb = pack_conv_bf16(wt.permute(0, 2, 3, 1).reshape(64, -1, 1, 1), dev)
    pc.wrows = pack_dcn_rows_bf16(wt, dev)
    xa, oma = _act16(x, dev, ld=ld), _act32(om, dev)
    assert ops.dcn_conv_ok(xa, oma, pc, 8)
    got = ops.dcn_conv_bf16(xa, oma, pc, act)
    assert got.bf16
    _close(got.nchw(), want, 1.5 * BF, "fused DCN vs deform_conv2d semantics")
    two = ops.conv2d([ops.dcn_columns(xa, oma, 8)], pc, act, precision="bf16")
    _close(got.nchw(), two.nchw(), 1.5 * BF, "fused DCN vs columns + product")
    got2 = ops.dcn_conv_bf16(xa, oma, pc, act)               # bit-stable run to run (no atomics, fixed order)
    assert torch.equal(got.buf, got2.buf)


def test_spynet_prep_bf16_matches_fp32_kernel():
    from gpemsr_amd import ops
    dev = _dev()
    ref = torch.rand(2, 1, 16, 24, generator=torch.Generator().manual_seed(110))
    sup = torch.rand(2, 1, 16, 24, generator=torch.Generator().manual_seed(111))
    fc = _rand(2, 2, 8, 12, seed=112, scale=1.5)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    A = lambda t: ops.Act(t.permute(0, 2, 3, 1).contiguous().reshape(-1).to(dev), t.shape[0], t.shape[2], t.shape[3], t.shape[1], t.shape[1], 0)   # noqa: E731
    up16, in16 = ops.spynet_prep_bf16(A(ref), A(sup), A(fc), mean, std)
    up32, in32 = ops.spynet_prep(A(ref), A(sup), A(fc), mean, std, pad16=True)
    _close(up16.nchw(), up32.nchw(), 0.0, "up flow")
    _close(in16.nchw(), in32.nchw(), BF, "level input")


@pytest.mark.parametrize("n,h,w,scale", [(2, 16, 24, 8), (1, 32, 32, 8), (3, 8, 12, 16), (1, 10, 6, 8), (4, 32, 48, 8)])      # the last: 768 super-tiles, 3 per workgroup
def test_vgg_mask_fused_matches_the_layered_form(n, h, w, scale):
    """gpemsr_vgg_mask_bf16 (conv1_1 -> conv1_2 -> 16x16 patch cosine in one kernel, R:model/GPEMSR.py:385-395) against torch:
    relu1_2 of the image expanded to 3 identical channels, F.interpolate for the LR slice, unfold + normalize + sum."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv_bf16
    dev = _dev()
    H, W = h * scale, w * scale
    g = torch.Generator().manual_seed(7)
    ref = torch.rand(n, 1, H, W, generator=g)
    lr = torch.rand(n, 1, h, w, generator=g)
    w1 = _rand(64, 3, 3, 3, seed=8, scale=0.4); b1 = _rand(64, seed=9, scale=0.2)
    w2 = _r(_rand(64, 64, 3, 3, seed=10, scale=1.0 / 24)); b2 = _rand(64, seed=11, scale=0.1)
    w1s = _r(w1.sum(dim=1, keepdim=True))                                   # the kernel folds the 3 channels and rounds W1 to bf16

    def relu1_2(img):
        f = _r(F.relu(F.conv2d(img, w1s, b1, 1, 1)))                        # conv1_2's input is bf16 in the kernel
        return F.relu(F.conv2d(f, w2, b2, 1, 1))
    fa = relu1_2(ref)
    fb = relu1_2(F.interpolate(lr, scale_factor=scale, mode="bilinear", align_corners=False))
    ua, ub = F.unfold(fa, 16, stride=16), F.unfold(fb, 16, stride=16)
    want = (F.normalize(ua, dim=1) * F.normalize(ub, dim=1)).sum(1).view(n, 1, H // 16, W // 16)
    A = lambda t: ops.Act(t.reshape(-1).to(dev), t.shape[0], t.shape[2], t.shape[3], 1, 1, 0)   # noqa: E731
    got = ops.vgg_mask_bf16(A(ref), A(lr), scale, w1.sum(dim=1).reshape(64, 9).contiguous().to(dev), b1.to(dev), pack_conv_bf16(w2, dev), b2.to(dev))
    _close(got.nchw(), want, 2e-3, "fused vgg mask")
    again = ops.vgg_mask_bf16(A(ref), A(lr), scale, w1.sum(dim=1).reshape(64, 9).contiguous().to(dev), b1.to(dev), pack_conv_bf16(w2, dev), b2.to(dev))
    assert torch.equal(got.buf, again.buf)
    # the form the engine uses since round 3: the LR slice up-sampled once by gpemsr_bilinear, both images read the same way (scale = 1)
    up = ops.bilinear(A(lr), H, W)
    got2 = ops.vgg_mask_bf16(A(ref), up, 1, w1.sum(dim=1).reshape(64, 9).contiguous().to(dev), b1.to(dev), pack_conv_bf16(w2, dev), b2.to(dev))
    _close(got2.nchw(), want, 2e-3, "fused vgg mask, pre-up-sampled LR")
    _close(got2.nchw(), got.nchw(), 2e-4, "both forms")


@pytest.mark.parametrize("n,h,w,ld,off", [(2, 16, 24, 64, 0), (1, 37, 131, 96, 16), (3, 5, 62, 64, 0), (1, 70, 63, 64, 0)])
def test_tap_sum_conv_64_to_1(n, h, w, ld, off):
    """gpemsr_conv_c64_cout1_bf16 (csrc/tap_sum.hip): hi + lo weight halves keep the weights at fp32 accuracy, so the result is the
    fp32 convolution of the bf16-rounded input to 1e-5; tiles own 16 x 62 pixels, the cases cross both tile edges."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_cout1_taps
    dev = _dev()
    x = _r(_rand(n, 64, h, w, seed=300 + h))
    wt = _rand(1, 64, 3, 3, seed=301, scale=1.0 / 24); b = _rand(1, seed=302)
    base = _rand(n, 1, h, w, seed=303)
    pc = pack_conv(wt, b, dev)
    pc.wtap = pack_cout1_taps(wt, dev)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 1)
    got = ops.conv2d([_act16(x, dev, ld, off)], pc, 0, residual=_act32(base, dev), precision="bf16")
    assert not got.bf16
    _close(got.nchw(), want + base.double(), 1e-5, "tap-sum 64 -> 1 with residual")
    got = ops.conv2d([_act16(x, dev, ld, off)], pc, 4, precision="bf16")
    _close(got.nchw(), torch.sigmoid(F.leaky_relu(want, 0.1)), 1e-5, "tap-sum 64 -> 1 mask head")
    pc.b = None
    got = ops.conv2d([_act16(x, dev, ld, off)], pc, 0, precision="bf16")
    _close(got.nchw(), want - b.double(), 1e-5, "tap-sum 64 -> 1 without bias")


@pytest.mark.parametrize("n,h,w,ld,off", [(2, 8, 12, 64, 0), (1, 19, 67, 96, 32), (2, 3, 62, 64, 0), (1, 33, 125, 64, 0), (1, 1, 1, 64, 0)])
def test_upconv_out_composed_operator(n, h, w, ld, off):
    """gpemsr_upconv_out_c64_bf16: ConvTranspose2d(64 -> 64, k3 s2 p1 op1) + Conv2d(64 -> 1, 3x3) as one operator must equal the
    layered fp64 evaluation on the same bf16-rounded input (no intermediate rounding in either): 1e-5, borders included."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_upconv_out
    dev = _dev()
    x = _r(_rand(n, 64, h, w, seed=310 + h))
    w1 = _rand(64, 64, 3, 3, seed=311, scale=1.0 / 12); b1 = _rand(64, seed=312, scale=0.5)
    w2 = _rand(1, 64, 3, 3, seed=313, scale=1.0 / 24); b2 = _rand(1, seed=314)
    want = F.conv2d(F.conv_transpose2d(x.double(), w1.double(), b1.double(), stride=2, padding=1, output_padding=1), w2.double(), b2.double(), padding=1)
    frag, consts = pack_upconv_out(w1, b1, w2, b2, dev)
    got = ops.upconv_out_bf16(_act16(x, dev, ld, off), frag, consts)
    assert not got.bf16 and (got.h, got.w, got.c) == (2 * h, 2 * w, 1)
    g, wv = got.nchw().cpu().double(), want
    for name, sl in (("top row", (slice(None), slice(None), slice(0, 1))), ("left column", (slice(None), slice(None), slice(None), slice(0, 1))),
                     ("bottom row", (slice(None), slice(None), slice(-1, None))), ("right column", (slice(None), slice(None), slice(None), slice(-1, None)))):
        err = (g[sl] - wv[sl]).abs().max().item()
        assert err <= 1e-5 * wv.abs().max().item() + 1e-6, f"{name}: {err:.3e}"
    _close(got.nchw(), want, 1e-5, "composed up-block + output layer")


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (1, 37, 70), (3, 5, 9), (1, 64, 33)])
def test_spynet_flow_update_row_sums(n, h, w):
    """gpemsr_conv7_c16_cout2_bf16 (rowsum7_kernel): Conv2d(16 -> 2, 7x7) + fp32 residual == the fp64 convolution of the
    bf16-rounded input with the fp32 weights (hi + lo halves) to 1e-5; 16 x 32 tiles, ragged edges, images smaller than a tile."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv, pack_rowsum7
    dev = _dev()
    x = _r(_rand(n, 16, h, w, seed=400 + h))
    wt = _rand(2, 16, 7, 7, seed=401, scale=1.0 / 28); b = _rand(2, seed=402)
    up = _rand(n, 2, h, w, seed=403, scale=3.0)
    pc = pack_conv(wt, b, dev)
    pc.wrow7 = pack_rowsum7(wt, dev)
    want = F.conv2d(x.double(), wt.double(), b.double(), 1, 3) + up.double()
    got = ops.conv2d([_act16(x, dev)], pc, 0, residual=_act32(up, dev), precision="bf16", out_f32=True)
    assert not got.bf16 and got.c == 2
    _close(got.nchw(), want, 1e-5, "flow update with residual")
    pc.b = None
    got = ops.conv2d([_act16(x, dev)], pc, 0, precision="bf16", out_f32=True)
    _close(got.nchw(), want - up.double() - b.double().view(1, 2, 1, 1), 1e-5, "flow update, no bias, no residual")


@pytest.mark.parametrize("rows_hw,cin,cout", [((2, 16, 32), 512, 1024), ((1, 5, 12), 64, 96)])
def test_linear_as_three_bf16_products_keeps_fp32_precision(rows_hw, cin, cout):
    """The indexer's nn.Linear in bf16 mode (engine.indexer_logits): ops.split_hi_lo_bf16 + packing.pack_linear_bf16x3 + the 1x1 bf16
    kernel over [hi, lo, hi].  hi + lo reproduces the fp32 input to 2^-16; the logits match the fp64 product of the UNROUNDED fp32
    operands to 2e-5 of the logit scale (fp32 kernel: ~1e-6; one bf16 rounding of either operand alone would be ~2e-3)."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_linear_bf16x3
    dev = _dev()
    n, h, w = rows_hw
    x = _rand(n, cin, h, w, seed=500 + cin, scale=3.0)
    wt = _rand(cout, cin, seed=501, scale=1.0 / cin ** 0.5); b = _rand(cout, seed=502)
    hi, lo = ops.split_hi_lo_bf16(_act32(x, dev))
    assert hi.bf16 and lo.bf16
    rec = hi.nchw().double() + lo.nchw().double()
    assert float((rec.cpu() - x.double()).abs().max()) <= 2.0 ** -16 * float(x.abs().max())
    assert torch.equal(hi.nchw().cpu().float(), _r(x))
    pc = pack_linear_bf16x3(wt, b, dev)
    got = ops.conv2d([hi, lo, hi], pc, 0, precision="bf16", out_f32=True)
    assert not got.bf16 and got.c == cout
    want = torch.einsum("nchw,oc->nohw", x.double(), wt.double()) + b.double().view(1, -1, 1, 1)
    _close(got.nchw(), want, 2e-5, "three-product linear")


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (3, 37, 70), (1, 5, 5), (9, 64, 96)])
def test_conv7_c32_cout16_on_the_16x16x32_shape(n, h, w):
    """SpyNet's 32 -> 16 7x7 layers (basicsr BasicModule via R:model/GPEMSR.py:67,99) on v_mfma_f32_16x16x32_bf16 with resident weights
    (csrc/conv7_bf16.hip): against torch on the bf16-rounded operands, against the ring kernel (variant 9), ragged tiles, several
    tiles per workgroup (the last case: 9 x 4 x 3 tiles... on a 256-CU chip one each; the persistent loop is exercised by n = 80 in the
    forward), all three activations, a result written into a channel slice, bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv7_c32_cout16
    dev = _dev()
    x = _r(_rand(n, 32, h, w, seed=41))
    wt = _r(_rand(16, 32, 7, 7, seed=42, scale=1.0 / np.sqrt(32 * 49)))
    b = _rand(16, seed=43, scale=0.2)
    lin = F.conv2d(x, wt, b, 1, 3)
    pc = _pc(wt, b, dev)
    pc.w7c16 = pack_conv7_c32_cout16(wt, dev)
    xa = _act16(x, dev)
    for act, want in ((1, F.relu(lin)), (0, lin), (2, F.leaky_relu(lin, 0.1))):
        got = ops.conv2d([xa], pc, act, precision="bf16")
        _close(got.nchw(), want, BF, f"conv7 32->16 act {act}")
    ring = ops.conv2d([xa], pc, 1, precision="bf16", variant=9)
    _close(ring.nchw(), F.relu(lin), BF, "ring kernel")
    a1 = ops.conv2d([xa], pc, 1, precision="bf16")
    a2 = ops.conv2d([xa], pc, 1, precision="bf16")
    assert torch.equal(a1.buf, a2.buf), "not bit-stable run to run"
    wide = ops.new_act(n, h, w, 48, device=dev, bf16=True, zero=True)
    ops.conv2d([xa], pc, 0, precision="bf16", out=wide.slice(16, 16))
    _close(wide.slice(16, 16).nchw(), lin, BF, "into a slice")
    assert float(wide.slice(0, 16).nchw().abs().max()) == 0.0 and float(wide.slice(32, 16).nchw().abs().max()) == 0.0
    # a source that is a channel slice of a wider buffer (ld > 32)
    xw = _act16(torch.cat([x, _r(_rand(n, 32, h, w, seed=44))], 1), dev)
    got = ops.conv2d([xw.slice(0, 32)], pc, 0, precision="bf16")
    _close(got.nchw(), lin, BF, "sliced source")


@pytest.mark.parametrize("n,h,w,cout", [(3, 16, 24, 1024), (2, 9, 7, 1024), (1, 64, 64, 1000), (5, 32, 32, 256)])
def test_argmax_in_the_logits_gemm_epilogue(n, h, w, cout):
    """R:model/indexer.py:100 + R:model/codebook.py:34-43 without the logits in memory: gpemsr_conv16_desc.rowmax + gpemsr_rowmax_finish must
    return exactly the indices gpemsr_argmax_rows finds in the stored logits of the same three-product GEMM (same accumulators, same
    bias add), incl. ragged row tiles, a column count that is not a multiple of the 128-column tile, and exact ties (lowest column)."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_linear_bf16x3
    dev = _dev()
    cin = 512
    x = _rand(n, cin, h, w, seed=700 + cout, scale=2.0)
    wt = _rand(cout, cin, seed=701, scale=1.0 / cin ** 0.5); b = _rand(cout, seed=702)
    wt[5] = wt[900 % cout]; b[5] = b[900 % cout]                   # two identical columns: an exact tie wherever they hold the maximum
    wt[5] *= 3.0; wt[900 % cout] *= 3.0                            # ... which they often do
    hi, lo = ops.split_hi_lo_bf16(_act32(x, dev))
    pc = pack_linear_bf16x3(wt, b, dev)
    logits = ops.conv2d([hi, lo, hi], pc, 0, precision="bf16", out_f32=True)
    want = ops.argmax_rows(logits)
    got = ops.conv2d([hi, lo, hi], pc, 0, precision="bf16", argmax=True)
    assert got.dtype == torch.int32 and got.numel() == n * h * w
    assert torch.equal(got, want.to(torch.int32).reshape(-1)), int((got != want.reshape(-1)).sum())
    ties = int((want.reshape(-1) == 5).sum())
    assert ties > 0, "the tie columns never won: the test does not exercise the tie rule"
    assert int((got == 900 % cout).sum()) == 0 or cout <= 5
    ref = logits.torch().reshape(-1, cout).argmax(dim=1).to(torch.int32)
    assert float((ref == got).float().mean()) > 0.999               # (torch's CUDA argmax may break ties differently)


GEMM_DIRECT_CASES = [
    # (n, cins, cout, h, w, act, residual bf16, out_f32, kpack)      -- gemm_bf16.hip: cout % 256 == 0, 64-channel chunks, an even number >= 4 of them
    (2, (512,), 512, 8, 16, 0, True, False, False),          # one ragged 128-row tile per image and cout block
    (3, (512,), 512, 24, 24, 1, False, False, False),        # 576 rows: two full tiles + a ragged one, several tiles per workgroup stream
    (1, (256,), 256, 20, 13, 2, False, True, False),         # four chunks (the minimum), fp32 result, LeakyReLU
    (2, (512,), 1024, 16, 16, 0, False, False, True),        # B-operand ("kpack") store, four cout blocks
    (1, (512, 512, 512), 1024, 17, 19, 0, False, True, False),   # three sources (the three-product logits GEMM's shape), 24 chunks
    (70, (512,), 512, 32, 32, 0, True, False, False),        # 560 tiles on 256 workgroups: the chunk stream crosses tiles, images and cout blocks
]


@pytest.mark.parametrize("case", GEMM_DIRECT_CASES)
def test_matrix_products_with_activations_straight_into_registers(case):
    """gemm_direct_bf16_kernel (csrc/gemm_bf16.hip; R:model/blocks.py:61-83 q / k / v / proj_out, R:model/indexer.py:89-100): against torch on the
    bf16-rounded operands, and bit-for-bit against nothing -- but the ring kernel (variant 14) must agree to the same tolerance, and the
    library must really have picked the new kernel for the default variant."""
    import ctypes as C
    from gpemsr_amd import _abi, ops
    n, cins, cout, h, w, act, use_res, out_f32, kpack = case
    dev = _dev()
    cin = sum(cins)
    x = _r(_rand(n, cin, h, w, seed=11))
    wt = _r(_rand(cout, cin, 1, 1, seed=12, scale=1.0 / np.sqrt(cin)))
    b = _rand(cout, seed=13, scale=0.1)
    want = _apply_act(F.conv2d(x, wt, b), act)
    res = _r(_rand(n, cout, h, w, seed=14)) if use_res else None
    if res is not None:
        want = want + res
    srcs, o = [], 0
    for i, c in enumerate(cins):
        srcs.append(_act16(x[:, o:o + c], dev, ld=c + (8 if i == 0 else 0), off=8 if i == 0 else 0))
        o += c
    pc = _pc(wt, b, dev, cins)
    r_act = None if res is None else _act16(res, dev)
    # the instantiation the library picks for this descriptor
    d = _abi.ConvDesc16()
    d.n, d.h, d.w, d.nsrc = n, h, w, len(srcs)
    for i, s_ in enumerate(srcs):
        d.src[i].ptr, d.src[i].ld, d.src[i].c = s_.ptr, s_.ld, s_.c
        d.src_image_stride[i] = -1
    d.cout, d.ksize, d.stride, d.weight, d.out, d.out_ld, d.out_f32, d.kpack = cout, 1, 1, pc.wb.data_ptr(), srcs[0].ptr, cout, int(out_f32), int(kpack)
    buf = C.create_string_buffer(160)
    assert _abi.load().gpemsr_conv2d_bf16_kernel_name(C.byref(d), buf, 160) == 0 and buf.value.decode().startswith("gemm_direct_bf16_kernel"), buf.value
    kw = dict(residual=r_act, precision="bf16", out_f32=out_f32, kpack=kpack, force_mfma=True)
    got = ops.conv2d(srcs, pc, act, **kw)
    ring = ops.conv2d(srcs, pc, act, variant=14, **kw)
    if kpack:       # [n][cout/8][pixels][8]
        unpack = lambda t: t.float().permute(0, 1, 3, 2).reshape(n, cout, h, w)      # noqa: E731
        _close(unpack(got), want, BF, str(case))
        _close(unpack(got), unpack(ring), BF, "new kernel vs ring kernel")
    else:
        assert got.bf16 == (not out_f32)
        _close(got.nchw(), want, 1e-4 if out_f32 else BF, str(case))
        _close(got.nchw(), ring.nchw(), 1e-4 if out_f32 else BF, "new kernel vs ring kernel")
    again = ops.conv2d(srcs, pc, act, **kw)                   # bit-stable run to run (screens the DMA / barrier discipline for races)
    assert torch.equal(got if kpack else got.buf, again if kpack else again.buf)


def test_per_image_weights_on_the_direct_product_kernel():
    """v^T = W_v . hn^T of the NonLocalBlock at C = 512 (R:model/blocks.py:72-80): the A operand is ONE weight matrix shared by all images
    (src_image_stride 0), the B operand a per-image activation in the kpack layout (weight_image_stride), the result again kpack."""
    from gpemsr_amd import ops
    dev = _dev()
    n, h, w, c = 3, 16, 32, 512
    T = h * w
    hn = _r(_rand(n, c, h, w, seed=21))
    wv = _r(_rand(c, c, 1, 1, seed=22, scale=1.0 / np.sqrt(c)))
    hn_a = _act16(hn, dev)
    hnp = ops.pack_rows_bf16(hn_a)
    wa = ops.Act(wv.reshape(c, c).to(torch.bfloat16).to(dev).contiguous().view(-1), n, 1, c, c, c, 0)
    vtp = ops.conv2d([wa], ops.PackedConv(None, None, 1, T, (c,), 32, wb=hnp), 0, weight_image_stride=T * c, src_image_stride=[0], kpack=True, precision="bf16")
    v_ref = F.conv2d(hn, wv).permute(0, 2, 3, 1).reshape(n, T, c)
    _close(vtp.float().permute(0, 2, 1, 3).reshape(n, c, T), v_ref.transpose(1, 2), BF, "v^T kpack, C = 512")


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (3, 37, 70), (1, 5, 5), (20, 64, 96)])
def test_conv7_c8_cout32_four_taps_per_mfma(n, h, w):
    """SpyNet's 8 -> 32 7x7 stems (basicsr BasicModule's first convolution, R:model/GPEMSR.py:67,99) with four taps per
    v_mfma_f32_16x16x32_bf16 (csrc/conv7_bf16.hip): x is the 16-channel tensor spynet_prep_bf16 makes (channels 8..15 zero -- here set to
    GARBAGE to prove the kernel never reads them), against torch, against the ring kernel on the zero-padded input, ragged tiles,
    several tiles per workgroup (three halo buffers in rotation), activations, bit-stable."""
    from gpemsr_amd import ops
    from gpemsr_amd.packing import pack_conv7_c8_cout32
    dev = _dev()
    x8 = _r(_rand(n, 8, h, w, seed=51))
    wt = _r(_rand(32, 8, 7, 7, seed=52, scale=1.0 / np.sqrt(8 * 49)))
    b = _rand(32, seed=53, scale=0.2)
    lin = F.conv2d(x8, wt, b, 1, 3)
    wpad = F.pad(wt, (0, 0, 0, 0, 0, 8))
    pc = _pc(wpad, b, dev)
    pc.w7c8 = pack_conv7_c8_cout32(wt, dev)
    xg = _act16(torch.cat([x8, 50.0 * _rand(n, 8, h, w, seed=54)], 1), dev)          # garbage in the padding channels
    xz = _act16(torch.cat([x8, torch.zeros(n, 8, h, w)], 1), dev)
    for act, want in ((1, F.relu(lin)), (0, lin), (2, F.leaky_relu(lin, 0.1))):
        got = ops.conv2d([xg], pc, act, precision="bf16")
        _close(got.nchw(), want, BF, f"conv7 8->32 act {act}")
    ring = ops.conv2d([xz], pc, 1, precision="bf16", variant=9)
    _close(ring.nchw(), F.relu(lin), BF, "ring kernel")
    a1 = ops.conv2d([xg], pc, 1, precision="bf16")
    a2 = ops.conv2d([xg], pc, 1, precision="bf16")
    assert torch.equal(a1.buf, a2.buf), "not bit-stable run to run"
