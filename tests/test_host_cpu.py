"""CPU: host logic -- architecture inventory vs the reference manifest, synthetic weights,
drop-in module behaviour, weight packing, option files, sharding helpers, C-ABI surface."""
import ctypes
import hashlib
import json
import math
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(scale):
    from gpemsr_amd.config import load_options
    return load_options(os.path.join(ROOT, "option", f"output_GPEMSR_x{scale}.yml"))


def build_model(*a, **k):
    from gpemsr_amd.config import build_model as bm
    return bm(*a, **k)


@pytest.mark.parametrize("scale", [8, 16])
def test_param_specs_match_reference_manifest(scale, golden_dir):
    """Key names, shapes, parameter counts == the reference model's state_dict (565 / 575 tensors)."""
    from gpemsr_amd.arch import param_specs
    man = json.load(open(os.path.join(golden_dir, f"state_manifest_x{scale}.json")))
    net = _opt(scale)["network"]
    specs = param_specs(scale=scale, **{k: v for k, v in net.items() if k not in ("ref_path_G", "ref_path_Indexer")})
    assert [k for k, _ in man["keys"]] == list(specs.keys())            # same ORDER as the reference too
    for k, shp in man["keys"]:
        assert list(specs[k].shape) == shp, k
    n_params = sum(int(np.prod(s.shape)) for s in specs.values() if not s.is_buffer)
    n_train = sum(int(np.prod(s.shape)) for s in specs.values() if s.trainable)
    assert (n_params, n_train) == (man["n_params"], man["n_trainable"])
    assert man["n_tensors"] == (565 if scale == 8 else 575)


def test_synthetic_weights_are_deterministic():
    from gpemsr_amd.arch import ParamSpec
    from gpemsr_amd.synth import synth_lr_tiles, synth_tensor
    t = synth_tensor("recon_trunk.3.conv1.weight", ParamSpec((64, 64, 3, 3), "conv_w", True), 0)
    assert hashlib.sha256(t.numpy().tobytes()).hexdigest()[:16] == hashlib.sha256(
        synth_tensor("recon_trunk.3.conv1.weight", ParamSpec((64, 64, 3, 3), "conv_w", True), 0).numpy().tobytes()).hexdigest()[:16]
    assert not torch.equal(t, synth_tensor("recon_trunk.4.conv1.weight", ParamSpec((64, 64, 3, 3), "conv_w", True), 0))
    assert abs(float(t.std()) - 1.4 / np.sqrt(576)) < 0.01
    x = synth_lr_tiles(2, 5, 16, 16, seed=3)
    assert x.shape == (2, 5, 1, 16, 16) and 0 <= float(x.min()) and float(x.max()) < 1
    assert torch.equal(x, synth_lr_tiles(2, 5, 16, 16, seed=3))


def test_module_is_a_drop_in(golden_dir):
    from gpemsr_amd.config import build_model
    from gpemsr_amd.synth import synth_state_dict
    m = build_model(_opt(8), load_prior_files=False)
    sd = m.state_dict()
    man = json.load(open(os.path.join(golden_dir, "state_manifest_x8.json")))
    assert list(sd.keys()) == [k for k, _ in man["keys"]]
    frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
    assert all(k.startswith(("vgg.", "refmodel.", "align_module.spynet.")) for k in frozen)
    assert all(p.requires_grad for k, p in m.named_parameters() if not k.startswith(("vgg.", "refmodel.", "align_module.spynet.")))
    # strict load of a full stage-3 dict (output_GPEMSR.py:52), tolerant of missing spynet mean/std buffers only
    sd2 = synth_state_dict(m._specs, seed=7)
    m.load_state_dict(sd2, strict=True)
    assert torch.equal(m.state_dict()["conv_first.weight"], sd2["conv_first.weight"])
    sd3 = {k: v for k, v in sd2.items() if not k.endswith(("spynet.mean", "spynet.std"))}
    m.load_state_dict(sd3, strict=True)
    bad = dict(sd2); bad.pop("HRconv.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    assert m.eval() is m and m.train() is m
    with pytest.raises(RuntimeError, match="cuda/HIP"):
        m(torch.rand(1, 5, 1, 16, 16))                      # no CPU fallback, by design
    with pytest.raises(ValueError):
        from gpemsr_amd.model import GPEMSR
        GPEMSR(None, None, _opt(8)["network"]["argref"], mode="8to1", scale=4)


def test_option_files_accepted_like_the_reference():
    for scale in (8, 16):
        o = _opt(scale)
        assert o["scale"] == scale and o["dataset"]["N_frames"] == 5
        for k in ("save_path", "pretrain_path"):
            assert k in o
        net = o["network"]
        assert net["mode"] == f"{scale}to1" and net["ref_fusion_feat_RBs"] == 1
        assert ("Indexer16" if scale == 16 else "Indexer8") in net["argref"]
    from gpemsr_amd.config import dict_to_nonedict
    nd = dict_to_nonedict({"a": {"b": 1}})
    assert nd["zzz"] is None and nd["a"]["nope"] is None and nd["a"]["b"] == 1


def test_additive_option_keys_reach_the_model():
    """`precision`, `indexer_precision` and `winograd` are option keys a reference user finds in the YAML, under `network:`
    or at the top level; a bare YAML `off` (which the loader reads as False) still means the direct form."""
    import copy
    from gpemsr_amd.config import build_model
    base = _opt(8)
    m = build_model(copy.deepcopy(base), load_prior_files=False)
    assert (m.precision, m.indexer_precision, m.winograd) == ("fp32", "bf16", None)       # None = the engine's default, "f4x4"
    for where in ("top", "network"):
        o = copy.deepcopy(base)
        tgt = o if where == "top" else o["network"]
        tgt["winograd"], tgt["precision"], tgt["indexer_precision"] = "decoder_f4x4", "bf16", "fp32:all"
        m = build_model(o, load_prior_files=False)
        assert (m.precision, m.indexer_precision, m.winograd) == ("bf16", "fp32:all", "decoder_f4x4")
    o = copy.deepcopy(base)
    o["winograd"] = False
    assert build_model(o, load_prior_files=False).winograd == "off"
    # the keyword wins over the file (bench.py passes precision=...)
    o = copy.deepcopy(base)
    o["winograd"] = "f2x2"
    assert build_model(o, load_prior_files=False, winograd="off").winograd == "off"


def _unpack_conv(pc, splits):
    """Inverse of packing: [tap][cout][cin_pad] -> OIHW (drops the per-source zero padding)."""
    k = pc.ksize
    w = pc.w.cpu()
    pieces, off = [], 0
    for c in splits:
        cp = -(-c // pc.ck) * pc.ck
        pieces.append(w[:, :, off:off + c]); off += cp
        assert float(w[:, :, off - (cp - c):off].abs().sum()) == 0.0
    w = torch.cat(pieces, dim=2)
    return w.reshape(k, k, pc.cout, -1).permute(2, 3, 0, 1).contiguous()


def test_weight_packing_layouts():
    from gpemsr_amd.packing import pack_conv, pack_convT, pack_dcn, pack_vgg_first
    g = torch.Generator().manual_seed(0)
    w, b = torch.randn(64, 162, 3, 3, generator=g), torch.randn(64, generator=g)
    pc = pack_conv(w, b, "cpu", (64, 64, 32, 2))
    assert pc.w.shape == (9, 64, 64 + 64 + 32 + 8) and pc.ck == 8
    assert torch.equal(_unpack_conv(pc, (64, 64, 32, 2)), w)
    w1 = torch.randn(64, 320, 1, 1, generator=g)
    assert pack_conv(w1, None, "cpu").w.shape == (1, 64, 320) and pack_conv(w1, None, "cpu").ck == 32
    # pixel shuffle: conv with permuted rows, stored as (2i+j)*C/4 + c, equals F.pixel_shuffle of the plain conv
    wp, bp = torch.randn(16, 8, 3, 3, generator=g), torch.randn(16, generator=g)
    x = torch.randn(1, 8, 5, 6, generator=g)
    pcp = pack_conv(wp, bp, "cpu", pixel_shuffle=True)
    y = F.conv2d(x, _unpack_conv(pcp, (8,)), pcp.b, 1, 1)             # [1,16,5,6] in permuted row order
    cq = 4
    shuffled = torch.zeros(1, cq, 10, 12)
    for q in range(4):
        shuffled[:, :, (q >> 1)::2, (q & 1)::2] = y[:, q * cq:(q + 1) * cq]
    assert torch.allclose(shuffled, F.pixel_shuffle(F.conv2d(x, wp, bp, 1, 1), 2), atol=1e-6)
    # transposed conv: phase-stacked 2x2-tap form reproduces F.conv_transpose2d
    wt, bt = torch.randn(8, 32, 3, 3, generator=g), torch.randn(32, generator=g)
    pt = pack_convT(wt, bt, "cpu")
    assert pt.transposed and pt.w.shape == (4, 128, 8)
    xt = torch.randn(1, 8, 5, 7, generator=g)
    xp = F.pad(xt, (0, 1, 0, 1))                                  # in(i+1, j+1) beyond the image is zero
    stacked = sum(torch.einsum("nc,bchw->bnhw", pt.w[2 * dy + dx], xp[:, :, dy:dy + 5, dx:dx + 7])
                  for dy in range(2) for dx in range(2))          # [1,128,5,7]
    yt = torch.zeros(1, 32, 10, 14)
    for q in range(4):
        yt[:, :, (q >> 1)::2, (q & 1)::2] = stacked[:, q * 32:(q + 1) * 32]
    assert torch.allclose(yt + bt.view(1, -1, 1, 1), F.conv_transpose2d(xt, wt, bt, stride=2, padding=1, output_padding=1), atol=1e-5)
    # DCN: tap-major column order
    wd = torch.randn(64, 64, 3, 3, generator=g)
    pd = pack_dcn(wd, torch.zeros(64), "cpu")
    assert pd.w.shape == (1, 64, 576) and torch.equal(pd.w[0, :, 5 * 64:6 * 64], wd[:, :, 1, 2])
    # VGG conv1_1 on three identical channels == 1-channel conv with summed weights
    wv, bv = torch.randn(64, 3, 3, 3, generator=g), torch.randn(64, generator=g)
    img = torch.rand(1, 1, 7, 7, generator=g)
    a = F.conv2d(img.expand(-1, 3, -1, -1), wv, bv, 1, 1)
    c = F.conv2d(img, _unpack_conv(pack_vgg_first(wv, bv, "cpu"), (1,)), bv, 1, 1)
    assert torch.allclose(a, c, atol=1e-5)


def test_shard_ranges_cover_everything():
    from gpemsr_amd.dist import shard_range
    for total in (1, 7, 16, 128, 130):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_c_abi_library_exports_every_declared_symbol():
    """The built shared object loads without a GPU and exports every function include/gpemsr_hip.h declares."""
    from gpemsr_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "gpemsr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gpemsr_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_abi.SYMBOLS), (set(declared) ^ set(_abi.SYMBOLS))
    if not os.path.exists(_abi.lib_path()):
        from gpemsr_amd.build import build_library
        build_library()
    lib = ctypes.CDLL(_abi.lib_path())
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.gpemsr_abi_version() == 1
    lib.gpemsr_last_error.restype = ctypes.c_char_p
    assert lib.gpemsr_conv2d(None, None) == -1 and b"null descriptor" in lib.gpemsr_last_error()   # validation only, no compute
    assert ctypes.sizeof(_abi.ConvDesc) == 248        # matches the C struct layout (static_assert in conv_mfma.hip)
    assert ctypes.sizeof(_abi.ConvDesc16) == 272      # static_assert in conv_bf16.hip
    assert lib.gpemsr_conv2d_bf16(None, None) == -1


def test_image_helpers_match_reference_semantics(golden_dir):
    from gpemsr_amd.imgutil import calculate_psnr, tensor2img
    d = np.load(os.path.join(golden_dir, "x8_lr16_b1_uniform.npz"))
    u8 = tensor2img(torch.from_numpy(d["out"]))
    assert np.array_equal(u8, d["out_u8"])
    assert calculate_psnr(u8, u8) == float("inf")
    x = torch.tensor([[0.5 / 255, 1.5 / 255, 2.5 / 255, -1.0, 2.0]])
    assert tensor2img(x).tolist() == [0, 2, 2, 0, 255]      # round-half-even, clamp


def test_cli_window_order_matches_reference_edges(tmp_path):
    """output_GPEMSR.py:54-128: slices 0,1 and n-2,n-1 use replicated neighbours; files are 0..n-1."""
    from PIL import Image
    import importlib.util
    spec = importlib.util.spec_from_file_location("cli", os.path.join(ROOT, "output_GPEMSR.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    n = 9
    for d in ("GT", "LQ"):
        os.makedirs(tmp_path / d)
        for i in range(n):
            Image.fromarray(np.full((8, 8), i * 10, dtype=np.uint8)).save(tmp_path / d / f"{i}.png")
    ds = cli.CREMIWindows({"N_frames": 5, "dataroot_GT": str(tmp_path / "GT"), "dataroot_LQ": str(tmp_path / "LQ")})
    assert len(ds) == n - 4
    wins = cli.build_windows(ds)
    ids = [[int(round(float(w[j, 0, 0, 0]) * 255 / 10)) for j in range(5)] for w in wins]
    assert ids[0] == [0, 0, 0, 1, 2] and ids[1] == [0, 0, 1, 2, 3]
    assert ids[2] == [0, 1, 2, 3, 4] and ids[n - 3] == [n - 5, n - 4, n - 3, n - 2, n - 1]
    assert ids[n - 2] == [n - 4, n - 3, n - 2, n - 1, n - 1] and ids[n - 1] == [n - 3, n - 2, n - 1, n - 1, n - 1]
    assert len(wins) == n and wins[0].shape == (5, 1, 8, 8) and wins[0].dtype == torch.float32


def test_cli_index_windows_dedupes_slices(tmp_path):
    """Volume mode bookkeeping: distinct LQ files in first-use order + [Wn,5] frame numbers that reproduce the windows."""
    from PIL import Image
    import importlib.util
    spec = importlib.util.spec_from_file_location("cli", os.path.join(ROOT, "output_GPEMSR.py"))
    cli = importlib.util.module_from_spec(spec); spec.loader.exec_module(cli)
    n = 8
    for d in ("GT", "LQ"):
        os.makedirs(tmp_path / d)
        for i in range(n):
            Image.fromarray(np.full((8, 8), i * 10, dtype=np.uint8)).save(tmp_path / d / f"{i}.png")
    ds = cli.CREMIWindows({"N_frames": 5, "dataroot_GT": str(tmp_path / "GT"), "dataroot_LQ": str(tmp_path / "LQ")})
    wp = cli.window_paths(ds)
    assert len(wp) == n
    files, win = cli.index_windows(wp)
    assert len(files) == n and win.shape == (n, 5) and win.dtype == torch.int32
    frames = cli.load_frames(files)
    full = cli.build_windows(ds)
    for w in range(n):
        assert torch.equal(frames[win[w].long()], full[w])
    # a block in the middle of the volume only needs its own slices (+ halo)
    files2, win2 = cli.index_windows(wp[3:5])
    assert len(files2) == 6 and int(win2.max()) == 5 and win2[0].tolist() == [0, 1, 2, 3, 4]


def test_missing_prior_file_raises_like_the_reference(tmp_path):
    """R:model/GPEMSR.py:275-276 torch.load()s the prior files unconditionally: a mistyped path must not silently train
    against the synthetic prior."""
    import pytest
    opt = _opt(8)
    net = dict(opt["network"])
    net["ref_path_G"] = str(tmp_path / "no_such_stage1.pth")
    opt = dict(opt, network=net)
    with pytest.raises(FileNotFoundError):
        build_model(opt, load_prior_files=True)
    build_model(opt, load_prior_files=False)        # explicit opt-out keeps the synthetic prior


def test_scheduler_defaults_do_not_index_past_t_period():
    """A trainer built from an option dict without `restarts`: T_period = [1 << 30], no restarts."""
    from gpemsr_amd.train import CosineAnnealingLRRestart, MultiStepLRRestart
    s = CosineAnnealingLRRestart(4e-4, [1 << 30], None, None, eta_min=1e-7)
    lrs = [s.step() for _ in range(5)]
    assert all(0 < v <= 4e-4 for v in lrs) and lrs[0] == pytest.approx(4e-4, rel=1e-6)
    m = MultiStepLRRestart(1e-3, [2, 4], None, None, gamma=0.5)
    assert [round(m.step(), 9) for _ in range(5)] == [1e-3, 5e-4, 5e-4, 2.5e-4, 2.5e-4]
    # with restarts the recurrence still follows R:model/lr_scheduler.py:36-68
    c = CosineAnnealingLRRestart(1.0, [4, 4], [4], [0.5], eta_min=0.0)
    seq = [c.step() for _ in range(6)]
    assert seq[4] == pytest.approx(0.5)              # e = 5 = restarts[0] + 1 -> base_lr * weight
    # `restarts` absent on a MULTI-period schedule: the reference's own default ([0] -> [1], weight 1) switches T_max to
    # T_period[1] at step 1 (R:model/lr_scheduler.py:40-42,50-53) -- ADVICE r2
    d = CosineAnnealingLRRestart(1.0, [4, 8], None, None, eta_min=0.0)
    seq = [d.step() for _ in range(3)]
    assert d.T_max == 8 and seq[0] == pytest.approx(1.0)
    assert seq[1] == pytest.approx((1 + math.cos(math.pi * 1 / 8)) / 2)     # cosine over T_period[1] counted from the restart at step 1


def test_packed_weights_follow_in_place_parameter_updates():
    """validation between optimizer steps must see the current weights.  Packing runs on the CPU too
    (only the kernels need a GPU), so the life-cycle is checked here: in-place update, p.data rebinding, sub-module
    load_state_dict and the trainers' HIP-written marker all refresh the packed copies; untouched entries are not repacked."""
    import torch
    m = build_model(_opt(8), load_prior_files=False)
    dev = torch.device("cpu")
    eng = m._get_engine(dev)
    w0 = eng.pc["conv_last"].w.clone()
    keep = eng.pc["HRconv"].w
    with torch.no_grad():
        m.conv_last.weight.mul_(2.0)                                   # what torch.optim does
    eng2 = m._get_engine(dev)
    assert eng2 is eng and torch.equal(eng.pc["conv_last"].w, 2.0 * w0)
    assert eng.pc["HRconv"].w is keep                                   # unchanged layers keep their packs
    m.conv_last.weight.data = m.conv_last.weight.data * 0.5             # rebinding the storage
    assert torch.equal(m._get_engine(dev).pc["conv_last"].w, w0)
    g0 = eng.par["refmodel.indexer.feat_extract.0.block.1.weight"].clone()
    sd = {k: v * 3.0 for k, v in m.refmodel.indexer.state_dict().items()}
    m.refmodel.indexer.load_state_dict(sd)                              # bypasses GPEMSR.load_state_dict
    assert torch.equal(m._get_engine(dev).par["refmodel.indexer.feat_extract.0.block.1.weight"], 3.0 * g0)
    # a HIP kernel writing through the raw pointer leaves torch's counters alone: the trainers mark such names
    m.HRconv.bias.data.view(-1)[0] = 123.0                              # .data writes do not bump _version
    m.mark_weights_written(["HRconv.bias"])
    assert float(m._get_engine(dev).pc["HRconv"].b[0]) == 123.0


def test_composed_decoder_tail_is_exact_in_float64():
    """packing.compose_upconv_out: ConvTranspose2d(k3 s2 p1 op1) followed by Conv2d(3x3, pad 1, one output channel) as ONE 5x5 stride-2
    operator + border routes + bias-through-taps table == the layered evaluation (float64, every border class, odd sizes).  This is the
    algebra csrc/tap_sum.hip (tap_sum_kernel<true, .>) implements; the GPU tests check the kernel, this checks the composition."""
    import torch.nn.functional as F
    from gpemsr_amd.packing import compose_upconv_out
    torch.manual_seed(0)
    C, M = 8, 6
    w1 = torch.randn(C, M, 3, 3, dtype=torch.float64); b1 = torch.randn(M, dtype=torch.float64)
    w2 = torch.randn(1, M, 3, 3, dtype=torch.float64); b2 = torch.randn(1, dtype=torch.float64)
    for (h, w) in ((5, 7), (1, 1), (2, 3)):
        x = torch.randn(2, C, h, w, dtype=torch.float64)
        ref = F.conv2d(F.conv_transpose2d(x, w1, b1, stride=2, padding=1, output_padding=1), w2, b2, padding=1)
        taps, s, bb, wy0, wx0, wc = compose_upconv_out(w1, b1, w2, b2)
        W5 = taps.reshape(5, 5, C).permute(2, 0, 1).unsqueeze(1)                      # [C][1][5][5]: o = 2 i + t - 2
        out = F.conv_transpose2d(x, W5, None, stride=2, padding=2, output_padding=1) + bb
        OH, OW = 2 * h, 2 * w
        S = s.reshape(3, 3)
        for oy in range(OH):
            for ox in range(OW):
                for dy in range(3):
                    for dx in range(3):
                        if 0 <= oy + dy - 1 < OH and 0 <= ox + dx - 1 < OW:
                            out[:, 0, oy, ox] += S[dy, dx]
        for ox in range(OW):                        # routes through intermediate row -1
            for t in range(ox & 1, 5, 2):
                jj = (ox + 2 - t) >> 1
                if 0 <= jj < w:
                    out[:, 0, 0, ox] -= x[:, :, 0, jj] @ wy0[t]
        for oy in range(OH):                        # ... column -1
            for t in range(oy & 1, 5, 2):
                ii = (oy + 2 - t) >> 1
                if 0 <= ii < h:
                    out[:, 0, oy, 0] -= x[:, :, ii, 0] @ wx0[t]
        out[:, 0, 0, 0] += x[:, :, 0, 0] @ wc       # subtracted twice
        assert float((out - ref).abs().max()) <= 1e-12 * max(1.0, float(ref.abs().max())), (h, w)


def test_mfma_fragment_packers_place_every_weight_where_the_kernel_reads_it():
    """packing._tap_fragments / _tap_fragments_f32 / pack_rowsum7: rebuild the weight rows from the lane layout the kernels document
    (bf16 32x32x16: lane l = row l % 32, channels 16 ks + 8 (l // 32) .. + 8; fp32 32x32x2: lane l = row l % 32, channel 2 ks + l // 32)."""
    from gpemsr_amd.packing import _tap_fragments, _tap_fragments_f32, pack_cout1_taps, pack_rowsum7
    g = torch.Generator().manual_seed(3)
    rows = (torch.rand(25, 64, generator=g) - 0.5).to(torch.bfloat16).to(torch.float32)
    f = _tap_fragments(rows).to(torch.float32)                                         # [4][64][8]
    back = torch.zeros(32, 64)
    for ks in range(4):
        for l in range(64):
            back[l % 32, 16 * ks + 8 * (l // 32): 16 * ks + 8 * (l // 32) + 8] = f[ks, l]
    assert torch.equal(back[:25], rows) and float(back[25:].abs().max()) == 0.0
    r32 = torch.rand(9, 64, generator=g) - 0.5
    f32 = _tap_fragments_f32(r32)                                                        # [32][64]
    back = torch.zeros(32, 64)
    for ks in range(32):
        for l in range(64):
            back[l % 32, 2 * ks + l // 32] = f32[ks, l]
    assert torch.equal(back[:9], r32)
    w = torch.rand(1, 64, 3, 3, generator=g) - 0.5
    ft = pack_cout1_taps(w, "cpu").to(torch.float32)                                      # hi rows 0..8, lo rows 16..24
    back = torch.zeros(32, 64)
    for ks in range(4):
        for l in range(64):
            back[l % 32, 16 * ks + 8 * (l // 32): 16 * ks + 8 * (l // 32) + 8] = ft[ks, l]
    want = w[0].permute(1, 2, 0).reshape(9, 64)
    assert float((back[0:9] + back[16:25] - want).abs().max()) <= 2.0 ** -16 * float(want.abs().max())
    w7 = torch.rand(2, 16, 7, 7, generator=g) - 0.5
    f7 = pack_rowsum7(w7, "cpu").to(torch.float32)                                        # [7 kx][64 lanes][8]
    for kx in range(7):
        back = torch.zeros(32, 16)
        for l in range(64):
            back[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = f7[kx, l]
        want = w7[:, :, :, kx].permute(2, 0, 1).reshape(14, 16)                            # row 2 ky + co
        assert float((back[0:14] + back[16:30] - want).abs().max()) <= 2.0 ** -16 * float(want.abs().max()), kx


def test_volume_window_table_matches_the_reference_script():
    """dist.volume_window_rows == the index arithmetic of R:output_GPEMSR.py:54-128 (first / last two outputs replicate edge slices)."""
    from gpemsr_amd.dist import volume_window_rows
    rows = volume_window_rows(7)
    assert rows == [[0, 0, 0, 1, 2], [0, 0, 1, 2, 3], [0, 1, 2, 3, 4], [1, 2, 3, 4, 5], [2, 3, 4, 5, 6], [3, 4, 5, 6, 6], [4, 5, 6, 6, 6]]


def test_winograd_f27_transforms_are_an_exact_identity():
    """The 1-D Winograd F(2, 7) form of the fp32 SpyNet 7x7 layers (csrc/conv7_wino.hip, packing.pack_winograd7): with G (host, float64) and the
    B^T / A^T constants the kernel hard-codes, [y0, y1] = A^T [(G g) (.) (B^T d)] is the correlation of 8 inputs with 7 taps -- in float64 to
    1e-12; and a whole 7x7 convolution assembled from the packed U tensor (chunk / filter-row / position / quad layout) equals F.conv2d."""
    import torch
    import torch.nn.functional as F
    from gpemsr_amd.packing import WINO7_AT, WINO7_G, pack_winograd7, wino7_bt
    bt = wino7_bt()
    # the constants of conv7_wino.hip, row by row
    kernel_bt = torch.tensor([
        [1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0],
        [0, -2 / 9, -2 / 9, 17 / 18, 17 / 18, -2 / 9, -2 / 9, 0],
        [0, 2 / 9, -2 / 9, -17 / 18, 17 / 18, 2 / 9, -2 / 9, 0],
        [0, 1 / 180, 1 / 360, -1 / 36, -1 / 72, 1 / 45, 1 / 90, 0],
        [0, -1 / 180, 1 / 360, 1 / 36, -1 / 72, -1 / 45, 1 / 90, 0],
        [0, 64 / 45, 128 / 45, -16 / 9, -32 / 9, 16 / 45, 32 / 45, 0],
        [0, -64 / 45, 128 / 45, 16 / 9, -32 / 9, -16 / 45, 32 / 45, 0],
        [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]], dtype=torch.float64)
    assert float((bt - kernel_bt).abs().max()) < 1e-12
    g = torch.Generator().manual_seed(5)
    d = torch.rand(8, generator=g, dtype=torch.float64) - 0.5
    taps = torch.rand(7, generator=g, dtype=torch.float64) - 0.5
    y = WINO7_AT @ ((WINO7_G @ taps) * (kernel_bt @ d))
    want = torch.stack([(d[0:7] * taps).sum(), (d[1:8] * taps).sum()])
    assert float((y - want).abs().max()) < 1e-12
    # whole layer through the packed tensor: U[chunk][ky][nu][quad][cout][4]
    cin, cout, h, w = 16, 32, 6, 12
    x = torch.rand(1, cin, h, w, generator=g, dtype=torch.float64) - 0.5
    wt = torch.rand(cout, cin, 7, 7, generator=g, dtype=torch.float64) - 0.5
    U = pack_winograd7(wt, "cpu").double()                                # (rounded to fp32 on the way: 1e-7)
    assert tuple(U.shape) == (cin // 8, 7, 8, 2, cout, 4)
    Uc = U.permute(1, 2, 4, 0, 3, 5).reshape(7, 8, cout, cin)             # [ky][nu][cout][cin]
    xp = F.pad(x, (3, 3, 3, 3))[0]                                         # [cin][h+6][w+6]
    out = torch.zeros(cout, h, w, dtype=torch.float64)
    for yy in range(h):
        for j in range(w // 2):
            M = torch.zeros(8, cout, dtype=torch.float64)
            for ky in range(7):
                V = torch.einsum("pi,ci->pc", kernel_bt, xp[:, yy + ky, 2 * j:2 * j + 8])       # [nu][cin]
                M += torch.einsum("pc,poc->po", V, Uc[ky])
            out[:, yy, 2 * j:2 * j + 2] = (WINO7_AT @ M).T
    ref = F.conv2d(x, wt, None, 1, 3)[0]
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-6


def test_winograd_f2x2_7x7_packing_is_an_exact_identity():
    """The 2-D form F(2x2, 7x7) of the same layers (csrc/conv7_wino2d.hip, packing.pack_winograd77): Y = A^T [(G g G^T) (.) (B^T d B)] A with the
    1-D form's matrices in both directions; a whole convolution assembled from the packed U tensor exactly as the kernel indexes it -- tile of
    4 x 8 blocks of 2 x 2 outputs, 8 x 8 patch starting at (2 br, 2 bc) of the pad-3 halo, position p = 8 xi + nu, chunk / quad / element of a
    channel -- equals F.conv2d (U is rounded to fp32 on the way: 1e-6)."""
    import torch
    import torch.nn.functional as F
    from gpemsr_amd.packing import WINO7_AT, pack_winograd77, wino7_bt
    g = torch.Generator().manual_seed(9)
    cin, cout, h, w = 16, 32, 8, 16
    x = torch.rand(1, cin, h, w, generator=g, dtype=torch.float64) * 2 - 1
    wt = (torch.rand(cout, cin, 7, 7, generator=g, dtype=torch.float64) * 2 - 1) / (7 * cin ** 0.5)
    U = pack_winograd77(wt, "cpu").double()
    assert tuple(U.shape) == (cin // 8, 64, 2, cout, 4)
    bt, at = wino7_bt(), WINO7_AT
    xp = F.pad(x, (3, 3, 3, 3))[0]
    out = torch.zeros(cout, h, w, dtype=torch.float64)
    for br in range(4):
        for bc in range(8):
            V = torch.einsum("ai,cij,bj->cab", bt, xp[:, 2 * br:2 * br + 8, 2 * bc:2 * bc + 8], bt)        # [cin][xi][nu]
            M = torch.zeros(8, 8, cout, dtype=torch.float64)
            for ch in range(cin):
                M += V[ch][:, :, None] * U[ch // 8, :, (ch % 8) // 4, :, ch % 4].reshape(8, 8, cout)
            out[:, 2 * br:2 * br + 2, 2 * bc:2 * bc + 2] = torch.einsum("ia,abo,jb->oij", at, M, at)
    ref = F.conv2d(x, wt, None, 1, 3)[0]
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-6


def test_winograd_f4x4_transforms_are_an_exact_identity():
    """The Winograd F(4x4, 3x3) form of the many-channel fp32 layers (csrc/conv_wino4.hip, packing.pack_winograd4): with G (host, float64), the
    integer B^T the kernel evaluates as sums (w4_bt) and A^T (w4_at), Y = A^T [(G g G^T) (.) (B^T d B)] A is the 4x4 block of the pad-1
    correlation of a 6x6 patch with 3x3 taps -- in float64 to 1e-12; the kernel's factored sums equal the matrices; and a whole layer
    assembled from the packed U tensor (chunk / position / quad layout) equals F.conv2d."""
    import torch
    import torch.nn.functional as F
    from gpemsr_amd.packing import WINO4_AT, WINO4_BT, WINO4_G, pack_winograd4
    g = torch.Generator().manual_seed(9)
    d = torch.rand(6, 6, generator=g, dtype=torch.float64) - 0.5
    taps = torch.rand(3, 3, generator=g, dtype=torch.float64) - 0.5
    y = WINO4_AT @ ((WINO4_G @ taps @ WINO4_G.T) * (WINO4_BT @ d @ WINO4_BT.T)) @ WINO4_AT.T
    want = F.conv2d(d[None, None], taps[None, None])[0, 0]
    assert float((y - want).abs().max()) < 1e-12

    def kernel_bt(v):       # w4_bt of conv_wino4.hip, operation by operation
        a, b, c, e = v[4] - 4 * v[2], v[3] - 4 * v[1], v[4] - v[2], v[3] - v[1]
        return torch.stack([4 * v[0] + (-5 * v[2] + v[4]), a + b, a - b, 2 * e + c, -2 * e + c, 4 * v[1] + (-5 * v[3] + v[5])])

    def kernel_at(m):       # w4_at
        s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
        return torch.stack([(m[0] + s12) + s34, 2 * d34 + d12, 4 * s34 + s12, 8 * d34 + d12 + m[5]])
    v = torch.rand(6, generator=g, dtype=torch.float64)
    assert float((kernel_bt(v) - WINO4_BT @ v).abs().max()) < 1e-12 and float((kernel_at(v) - WINO4_AT @ v).abs().max()) < 1e-12
    # whole layer through the packed tensor: U[chunk][p = 6 xi + nu][quad][cout][4]
    cin, cout, h, w = 16, 64, 8, 12
    x = torch.rand(1, cin, h, w, generator=g, dtype=torch.float64) - 0.5
    wt = torch.rand(cout, cin, 3, 3, generator=g, dtype=torch.float64) - 0.5
    U = pack_winograd4(wt, "cpu").double()                                 # (rounded to fp32 on the way: 1e-7)
    assert tuple(U.shape) == (cin // 8, 36, 2, cout, 4)
    Uc = U.permute(1, 3, 0, 2, 4).reshape(6, 6, cout, cin)                 # [xi][nu][cout][cin]
    xp = F.pad(x, (1, 1, 1, 1))[0]
    out = torch.zeros(cout, h, w, dtype=torch.float64)
    for by in range(h // 4):
        for bx in range(w // 4):
            V = torch.einsum("xi,cij,yj->xyc", WINO4_BT, xp[:, 4 * by:4 * by + 6, 4 * bx:4 * bx + 6], WINO4_BT)
            M = torch.einsum("xyc,xyoc->xyo", V, Uc)
            out[:, 4 * by:4 * by + 4, 4 * bx:4 * bx + 4] = torch.einsum("ix,xyo,jy->oij", WINO4_AT, M, WINO4_AT)
    ref = F.conv2d(x, wt, None, 1, 1)[0]
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-6


def test_winograd_f4x4_packing_pads_couts_and_permutes_pixel_shuffle_rows():
    """packing.pack_winograd4: cout % 64 != 0 -> zero rows up to the next multiple of 64 (the kernel never stores them); PixelShuffle layers ->
    the row permutation of pack_conv ((2 i + j) C / 4 + c), so that a cout block of 64 lands in ONE sub-pixel."""
    import torch
    from gpemsr_amd.packing import WINO4_G, pack_winograd4
    g = torch.Generator().manual_seed(3)
    wt = torch.rand(216, 16, 3, 3, generator=g) - 0.5
    U = pack_winograd4(wt, "cpu")
    assert tuple(U.shape) == (2, 36, 2, 256, 4) and float(U[:, :, :, 216:].abs().max()) == 0.0
    ref = torch.einsum("xa,ocab,yb->xyoc", WINO4_G, wt.double(), WINO4_G).float()          # [xi][nu][cout][cin]
    got = U.permute(1, 3, 0, 2, 4).reshape(6, 6, 256, 16)[:, :, :216]
    assert float((got - ref).abs().max()) < 1e-6
    wp = torch.rand(256, 8, 3, 3, generator=g) - 0.5
    Up = pack_winograd4(wp, "cpu", pixel_shuffle=True).permute(1, 3, 0, 2, 4).reshape(6, 6, 256, 8)
    refp = torch.einsum("xa,ocab,yb->xyoc", WINO4_G, wp.double(), WINO4_G).float()
    perm = torch.tensor([4 * c + q for q in range(4) for c in range(64)])
    assert float((Up - refp[:, :, perm]).abs().max()) < 1e-6
