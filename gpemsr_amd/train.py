"""Stage-3 training step on the HIP kernels: the reference's
``train_EMSR_onestep`` (/root/reference/GPEMSR-CREMI/GPEMSR/train_stage3.py:343-366)

    SR, ref_img = model(LR)
    loss = rec_loss_factor * L1(GT, SR) + ref_loss_factor * CX(VGG relu3_4(SR x t copies), VGG relu3_4(ref_img frames))
    loss.backward(); optimizer_G.step(); scheduler_G.step()

with the forward of gpemsr_amd/engine.py, a recorded tape of that forward for the backward, Adam on one flat parameter
buffer and (world > 1) one RCCL all-reduce of the flat gradient buffer (what DistributedDataParallel does, :141).

How the backward is built (DESIGN_HISTORY.md §3.6):
  * The engine issues every operator through ``self.o``; in training that is ``TapeOps``, which runs the same HIP kernel
    and, if an input depends on a trainable parameter, appends a closure that launches the operator's gradient kernels.
    Gradients live in a zero-initialised buffer per allocation (``Act.grad()``); every gradient kernel ACCUMULATES, which
    is what makes fan-out (a tensor with several consumers) and channel-slice writes (virtual ``cat``) free.
  * The frozen sub-networks (VQGAN prior, VGG for the mask, SpyNet) see inputs without gradient and hold no trainable
    parameter, so they are never recorded -- exactly the pruning autograd does in the reference.
  * Convolution data gradients are convolutions on the forward kernels with re-packed weights (stride 1: rotated taps and
    swapped channels; stride 2: the transposed form and vice versa); weight gradients are gpemsr_conv2d_wgrad.
  * Fusions whose backward needs an intermediate are un-fused in training only: the MPF mask multiply, and
    activation-then-residual (ThreeDA feat_fusion / spatial_attn3).

torch is used for allocation, streams and torch.distributed only; there is no CPU path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import os

import torch

from . import ops
from .dist import average_gradients
from .engine import Engine
from .ops import ACT_NONE, ACT_RELU, Act
from .packing import _pad_split, pack_conv, pack_conv_split, pack_winograd

VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)
# torchvision vgg19 features up to relu3_4 as sliced by model/VGG.py:17-25: (slice, index, kind)
_VGG_TO_RELU3_4 = ((1, 0, "conv"), (1, 2, "conv"), (2, 4, "pool"), (2, 5, "conv"), (2, 7, "conv"),
                   (3, 9, "pool"), (3, 10, "conv"), (3, 12, "conv"), (3, 14, "conv"), (3, 16, "conv"))


# ------------------------------------------------------------------------------------------------- schedulers
class CosineAnnealingLRRestart:
    """model/lr_scheduler.py:36-68 as a scalar recurrence (same update order: the value after ``step()`` number k is the
    reference's ``get_lr()`` at ``last_epoch == k``)."""

    def __init__(self, base_lr: float, T_period, restarts=None, weights=None, eta_min: float = 0.0):
        self.base_lr, self.lr = float(base_lr), float(base_lr)
        self.T_period = list(T_period)
        self.T_max = self.T_period[0]
        self.eta_min = float(eta_min)
        # `restarts` absent: the reference defaults to [0] -> [1] with weights [1] (model/lr_scheduler.py:40-42), i.e. at step 1 it
        # switches T_max to T_period[1] -- kept for a multi-period schedule.  For a ONE-period schedule that default indexes
        # T_period[1] and raises (:52); there, and only there, "no restarts" is used instead.
        if not restarts and len(self.T_period) > 1:
            restarts = [0]
        self.restarts = [v + 1 for v in restarts] if restarts else []
        self.restart_weights = (list(weights) if weights else [1] * len(self.restarts))[:len(self.restarts)] if self.restarts else []
        assert len(self.restarts) == len(self.restart_weights), 'restarts and their weights do not match.'
        assert not self.restarts or len(self.T_period) > len(self.restarts), 'T_period needs one entry more than restarts'
        self.last_restart = 0
        self.last_epoch = 0

    def step(self) -> float:
        self.last_epoch += 1
        e = self.last_epoch
        if e in self.restarts:
            self.last_restart = e
            self.T_max = self.T_period[self.restarts.index(e) + 1]
            self.lr = self.base_lr * self.restart_weights[self.restarts.index(e)]
        elif (e - self.last_restart - 1 - self.T_max) % (2 * self.T_max) == 0:
            self.lr = self.lr + (self.base_lr - self.eta_min) * (1 - math.cos(math.pi / self.T_max)) / 2
        else:
            self.lr = ((1 + math.cos(math.pi * (e - self.last_restart) / self.T_max)) /
                       (1 + math.cos(math.pi * ((e - self.last_restart) - 1) / self.T_max)) * (self.lr - self.eta_min) + self.eta_min)
        return self.lr


class MultiStepLRRestart:
    """model/lr_scheduler.py:8-33 (without clear_state) as a scalar recurrence."""

    def __init__(self, base_lr: float, milestones, restarts=None, weights=None, gamma: float = 0.1):
        self.base_lr, self.lr, self.gamma = float(base_lr), float(base_lr), float(gamma)
        self.milestones = {}
        for m in milestones:
            self.milestones[m] = self.milestones.get(m, 0) + 1
        self.restarts = [v + 1 for v in restarts] if restarts else []
        self.restart_weights = (list(weights) if weights else [1] * len(self.restarts)) if self.restarts else []
        self.last_epoch = 0

    def step(self) -> float:
        self.last_epoch += 1
        e = self.last_epoch
        if e in self.restarts:
            self.lr = self.base_lr * self.restart_weights[self.restarts.index(e)]
        elif e in self.milestones:
            self.lr = self.lr * self.gamma ** self.milestones[e]
        return self.lr


def make_scheduler(opt_train: dict):
    """The scheduler R:train_stage3.py:165-181 builds from the option file's ``train:`` block (lr_G, lr_scheme, T_period / lr_steps,
    restarts, restart_weights, eta_min / lr_gamma).  Host-only (no device): tests/test_train_cpu.py drives it with the reference's blocks."""
    o = opt_train
    lr = float(o.get("lr_G", 4e-4))
    scheme = o.get("lr_scheme") or "CosineAnnealingLR_Restart"
    if scheme == "MultiStepLR":
        return MultiStepLRRestart(lr, o["lr_steps"], o.get("restarts"), o.get("restart_weights"), o.get("lr_gamma", 0.1))
    if scheme == "CosineAnnealingLR_Restart":
        return CosineAnnealingLRRestart(lr, o.get("T_period", [1 << 30]), o.get("restarts"), o.get("restart_weights"), o.get("eta_min", 0.0))
    raise NotImplementedError(f"lr_scheme {scheme!r} (the reference knows MultiStepLR and CosineAnnealingLR_Restart)")


# ------------------------------------------------------------------------------------------------- the tape
class TapeOps:
    """The engine's operator namespace in training: same calls as ``gpemsr_amd.ops``; operators on the trainable part of
    the path also record their backward.  Anything not overridden here falls through to ``ops`` (no gradient)."""

    def __init__(self, eng: "TrainEngine"):
        self._eng = eng

    def __getattr__(self, k):
        return getattr(ops, k)

    def _rec(self, fn):
        self._eng.tape.append(fn)

    def bilinear(self, x: Act, oh: int, ow: int, align_corners: bool = False, mul: float = 1.0, out: Optional[Act] = None) -> Act:
        r = ops.bilinear(x, oh, ow, align_corners, mul, out)
        if self._eng.tape is not None and x.requires_grad:
            r.mark_grad()
            self._rec(lambda: ops.bilinear_bwd(r.grad(), x.grad(), align_corners, mul))
        return r

    def copy_channels(self, src: Act, dst: Act):
        ops.copy_channels(src, dst)
        if self._eng.tape is not None and src.requires_grad:
            dst.mark_grad()
            self._rec(lambda: ops.axpy(dst.grad(), src.grad()))

    def gather_images(self, src: Act, idx: torch.Tensor) -> Act:
        r = ops.gather_images(src, idx)
        if self._eng.tape is not None and src.requires_grad:
            r.mark_grad()
            self._rec(lambda: ops.scatter_add_images(r.grad(), idx, src.grad()))
        return r

    def copy_images(self, src: Act, n_dst: int, div: int, mul: int, add: int) -> Act:
        r = ops.copy_images(src, n_dst, div, mul, add)
        if self._eng.tape is not None and src.requires_grad:
            r.mark_grad()
            idx = ((torch.arange(n_dst, device=src.buf.device) // div) * mul + add).to(torch.int32)
            self._rec(lambda: ops.scatter_add_images(r.grad(), idx, src.grad()))
        return r

    def dcn_columns(self, x: Act, om: Act, groups: int) -> Act:
        col = ops.dcn_columns(x, om, groups)
        if self._eng.tape is not None and (x.requires_grad or om.requires_grad):
            col.mark_grad()
            self._rec(lambda: ops.dcn_columns_bwd(x, om, groups, col.grad(), x.grad() if x.requires_grad else None,
                                                  om.grad() if om.requires_grad else None))
        return col

    def temporal_gate(self, aligned: Act, emb: Act, emb_ref: Act, b: int, t: int) -> Act:
        af = ops.temporal_gate(aligned, emb, emb_ref, b, t)
        if self._eng.tape is not None:
            af.mark_grad()
            self._rec(lambda: ops.temporal_gate_bwd(aligned, emb, emb_ref, af.grad(), b, t, aligned.grad(), emb.grad(), emb_ref.grad()))
        return af

    def frame_mix_lrelu(self, af: Act, t: int, m: torch.Tensor, bias: torch.Tensor) -> Act:
        out = ops.frame_mix_lrelu(af, t, m, bias)
        eng = self._eng
        if eng.tape is not None:
            out.mark_grad()
            name = eng.par_name(m)                      # "ThreeDA.conv3D_k"
            self._rec(lambda: ops.frame_mix_lrelu_bwd(af, out, out.grad(), t, m, af.grad(), eng.gw[name], eng.gb[name]))
        return out

    def pool3s2_maxavg(self, x: Act) -> Act:
        y = ops.pool3s2_maxavg(x)
        if self._eng.tape is not None and x.requires_grad:
            y.mark_grad()
            self._rec(lambda: ops.pool3s2_maxavg_bwd(x, y.grad(), x.grad()))
        return y

    def threeda_combine(self, feat: Act, attn: Act, attn_add: Act, f2: Act, f3: Act) -> Act:
        out = ops.threeda_combine(feat, attn, attn_add, f2, f3)
        if self._eng.tape is not None:
            out.mark_grad()
            self._rec(lambda: ops.threeda_combine_bwd(feat, attn, out.grad(), feat.grad(), attn.grad(), attn_add.grad(), f2.grad(), f3.grad()))
        return out

    def maxpool2(self, x: Act) -> Act:
        y = ops.maxpool2(x)
        if self._eng.tape is not None and x.requires_grad:
            y.mark_grad()
            self._rec(lambda: ops.maxpool2_bwd(x, y.grad(), x.grad()))
        return y


def _pack_convT_dev(w: torch.Tensor, device) -> ops.PackedConv:
    """packing.pack_convT on the device (no host round trip): ConvTranspose2d(k3,s2,p1,op1) [Cin,Cout,3,3], Cout % 32 == 0."""
    cin, cout = w.shape[0], w.shape[1]
    assert cout % 32 == 0
    wf = w.detach().to(torch.float32)
    out = torch.zeros(4, 4 * cout, cin, dtype=torch.float32, device=device)
    co = torch.arange(cout, device=device)
    for dy in range(2):
        for dx in range(2):
            for py in range(dy, 2):
                for px in range(dx, 2):
                    rows = (co // 32) * 128 + (2 * py + px) * 32 + (co % 32)
                    out[2 * dy + dx, rows] = wf[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()
    return ops.PackedConv(_pad_split(out, (cin,), 8), None, 3, cout, (cin,), 8, transposed=True)


class TrainEngine(Engine):
    """Engine + tape.  ``trainable``: names (without .weight/.bias) of the parameters that receive gradients;
    ``gw`` / ``gb``: name -> gradient tensor in the reference's own layout (views of the flat gradient buffer)."""

    def __init__(self, sd, device, scale, nframes, groups, nf, dec_nrb, trainable, gw: Dict[str, torch.Tensor],
                 gb: Dict[str, torch.Tensor], precision: str = "fp32", wino_train: int = 0):
        self.trainable = set(trainable)                        # before the base constructor packs the weights
        # Winograd form (fp32 path) of FROZEN 3x3 layers, opt-in (`winograd_frozen`; GPEMSR_WINO_TRAIN overrides): bit 0 the prior / mask /
        # flow networks' forward, bit 1 the loss network's forward, bit 2 data gradients through frozen layers.  Off by default: the
        # direct kernel's gradients lie CLOSER to the float64 gradients than the reference's own fp32 step (median 1.5e-5 vs 1.4e-4,
        # test_gradient_distance_to_fp64_against_the_references_own); with the Winograd forms they lie at 2.2e-4 (fp32-grade, 1.5x the
        # reference's distance) for a step 7 % shorter (profiles/r04_winograd_training_sweep.log).
        self.wino_train = int(os.environ.get("GPEMSR_WINO_TRAIN", str(int(wino_train))))
        self.fuse_tail_f32 = False                             # training keeps the layered decoder tail / VALU 64 -> 1 convs: the tape's
        #                                                        backward kernels were validated against exactly that forward (DESIGN_HISTORY.md §3.6)
        # precision "bf16": the FROZEN sub-networks whose inputs carry no gradient -- the VQGAN prior (indexer, codebook, decoder), the
        # VGG relu1_2 mask and SpyNet: about two thirds of the step's convolution time -- run on the bf16 DATA PATH of the inference
        # engine (bf16 activations in HBM, fused kernels, flash attention); everything that is trained or differentiated through
        # (trainable layers, their gradients, the loss network) stays as in "bf16x3": fp32 activations, fp32-grade products.
        self._frozen16 = None
        if precision == "bf16":
            frozen = {k: v for k, v in sd.items() if k.startswith(("refmodel.", "vgg.slice1.", "align_module.spynet."))}
            self._frozen16 = Engine(frozen, device, scale, nframes, groups, nf, dec_nrb, frame_chunk=1 << 20, tile_chunk=1 << 20, precision="bf16")
            precision = "bf16x3"
        super().__init__(sd, device, scale, nframes, groups, nf, dec_nrb, frame_chunk=1 << 20, tile_chunk=1 << 20, precision=precision)
        self.gw, self.gb = gw, gb
        self.tape: Optional[list] = None
        self.o = TapeOps(self)
        self._dpc: Dict[tuple, ops.PackedConv] = {}
        self._dpc_frozen: Dict[tuple, ops.PackedConv] = {}
        self.wscale: Dict[str, float] = {}                     # layers whose packed forward weights carry a folded scale
        for sl, idx, kind in _VGG_TO_RELU3_4:                  # the loss network sees RGB: pack slices 1-3 as they are
            if kind == "conv":
                key = f"vgg.slice{sl}.{idx}"
                if (key + ".weight") not in sd:              # stage-2 generator objects carry no VGG
                    continue
                self.pc[key + "@rgb"] = pack_conv(sd[key + ".weight"], sd[key + ".bias"], device)
                wv = sd[key + ".weight"]
                if self.winograd and (self.wino_train & 2) and wv.shape[2] == 3 and wv.shape[0] % 32 == 0 and wv.shape[1] % 8 == 0:
                    self.pc[key + "@rgb"].wino = pack_winograd(wv, device)      # frozen loss network: Winograd form (fp32 path)
                if precision != "fp32" and sd[key + ".weight"].shape[1] % 16 == 0:
                    self.pc[key + "@rgb"].w16 = pack_conv_split(self.pc[key + "@rgb"], sd[key + ".weight"], device)

    # -- frozen sub-networks on the bf16 data path (precision "bf16") ------------------------------------------------------
    def ref_extract(self, xf, forced_idx, trace):
        if self._frozen16 is None:
            return super().ref_extract(xf, forced_idx, trace)
        feats = self._frozen16.ref_extract(xf, forced_idx, trace)       # constants of the step: nothing to record
        return [ops.cast_f32(f) if f.bf16 else f for f in feats]

    def vgg_mask_fused(self, ref_img, xf):
        return self._frozen16.vgg_mask_fused(ref_img, xf)

    def spynet(self, ref, supp):
        if self._frozen16 is None:
            return super().spynet(ref, supp)
        return self._frozen16.spynet(ref, supp)

    def par_name(self, t: torch.Tensor) -> str:
        for k, v in self.par.items():
            if v.data_ptr() == t.data_ptr() and k.endswith(".weight"):
                return k[:-7]
        raise KeyError("parameter tensor not found")

    def _pack_one(self, k: str, w: torch.Tensor):
        name = k[:-7] if k.endswith(".weight") else k
        if k.endswith(".weight") and w.dim() == 4 and name.startswith("reffea_L") and name in getattr(self, "trainable", ()):
            pc = _pack_convT_dev(w, self.dev)                  # per-step repack stays on the device
            pc.b = self.sd[name + ".bias"].detach().to(torch.float32).clone()
            self.pc[name] = pc
            return
        if name in getattr(self, "trainable", ()):             # trainable layers stay on the exact f32 kernel (their packed copy is
            prec, split = self.precision, self.split           # rebuilt every step; the split-bf16 packing is a host-side routine that
            self.precision, self.split = "fp32", False         # cost 440 ms per step when `split` alone still asked for it: round 2's
            try:                                               # "bf16x3" training extras ran at 24 instead of 118 samples/s)
                super()._pack_one(k, w)
            finally:
                self.precision, self.split = prec, split
            return
        super()._pack_one(k, w)

    def refresh_weights(self):
        """After an optimizer step: the packed copies of the trainable parameters follow the master weights."""
        self._dpc.clear()
        if getattr(self, "_ridx", None) is not None:           # one gather for all of them (enable_fast_refresh)
            torch.index_select(self._rflat, 0, self._ridx, out=self._rtmp)
            torch.mul(self._rtmp, self._rmask, out=self._rpacked)
            return
        for name in self.trainable:
            self._pack_one(name + ".weight", self.sd[name + ".weight"])

    def enable_fast_refresh(self, flat_p: torch.Tensor, verify: bool = True):
        """Every packed form of a trainable layer is a permutation of its master weights times a per-layer constant (1, or the folded
        C^-1/2 of an attention q projection) plus zero padding, so the per-step repack (93 layers x ~10 small torch launches in stage 3,
        ~1,400 launches in stage 1: host time serial with the device) collapses into ONE gather from the flat parameter buffer.  The index
        map is found by packing index-valued tensors through the very same routines -- as TWO planes (index + 1 = 4096 * hi + lo, both
        exact in fp32 whatever constant multiplies them), so the map covers the 42.6 M parameters of the stage-1 generator (one fp32 plane
        stops at 2^24) -- and the multiplier by packing ones.  The packed tensors become views of one flat buffer.  `verify`: the gathered
        buffer is compared once with the layer-by-layer repack (a packing routine that SUMS weights would not be a permutation)."""
        base, n_flat = flat_p.data_ptr(), flat_p.numel()
        assert n_flat < (1 << 31), "int32 gather indices"
        real, names = {}, sorted(self.trainable)
        offs = {}
        for name in names:
            for leaf in ("weight", "bias"):
                k = f"{name}.{leaf}"
                t = self.sd.get(k)
                if t is None:
                    continue
                off = (t.data_ptr() - base) // 4
                assert 0 <= off and off + t.numel() <= n_flat, f"{k} is not a view of the flat parameter buffer"
                real[k], offs[k] = t, off

        def packed_plane(plane: str) -> torch.Tensor:
            """All packed slots (16-byte aligned each) after packing the index plane `plane` in place of the weights."""
            for k, t in real.items():
                i1 = torch.arange(t.numel(), device=t.device, dtype=torch.int64) + (offs[k] + 1)
                v = (i1 & 4095) if plane == "lo" else ((i1 >> 12) if plane == "hi" else torch.ones_like(i1))
                self.sd[k] = v.to(torch.float32).view(t.shape)
            for name in names:
                self._pack_one(name + ".weight", self.sd[name + ".weight"])
            return torch.cat([_pad4(t) for _, _, t in self._refresh_slots(names)])

        def _pad4(t):                                        # every view must stay 16-byte aligned (the kernels' vector loads)
            t = t.reshape(-1)
            return torch.cat([t, t.new_zeros((-t.numel()) % 4)])
        try:
            one = packed_plane("one")
            lo, hi = packed_plane("lo"), packed_plane("hi")
        finally:
            self.sd.update(real)
        live = one != 0
        safe = torch.where(live, one, torch.ones_like(one))
        idx1 = (torch.round(hi / safe).to(torch.int64) << 12) + torch.round(lo / safe).to(torch.int64)
        idx1 = torch.where(live, idx1, torch.zeros_like(idx1))
        assert int(idx1.max()) <= n_flat and bool((idx1[live] > 0).all()), "index planes did not decode"
        self._rmask = one.clone()                            # multiplier: 0 in the padding
        self._ridx = (idx1 - 1).clamp_(min=0).to(torch.int32)
        self._rflat = flat_p
        self._rtmp = torch.empty_like(self._rmask)
        self._rpacked = torch.empty_like(self._rmask)
        for name in names:                                   # the slots are re-created from the real weights, then re-pointed into the buffer
            self._pack_one(name + ".weight", self.sd[name + ".weight"])
        slots = self._refresh_slots(names)
        want = torch.cat([_pad4(t) for _, _, t in slots]) if verify else None
        off = 0
        for obj, key, t in slots:
            view = self._rpacked[off:off + t.numel()].view(t.shape)
            if isinstance(obj, dict):
                obj[key] = view
            else:
                setattr(obj, key, view)
            off += (t.numel() + 3) // 4 * 4
        self.refresh_weights()
        if verify:
            assert torch.equal(self._rpacked, want), "one-gather repack differs from the layer-by-layer repack"

    def _refresh_slots(self, names):
        """(owner, key, tensor) of every packed tensor the per-step repack rewrites, in a fixed order."""
        slots = []
        for name in names:
            pc = self.pc.get(name)
            if pc is not None:
                for attr in ("w", "b"):
                    t = getattr(pc, attr, None)
                    if t is not None:
                        slots.append((pc, attr, t))
            for leaf in ("weight", "bias"):
                if f"{name}.{leaf}" in self.par:
                    slots.append((self.par, f"{name}.{leaf}", self.par[f"{name}.{leaf}"]))
        return slots

    # -- recorded convolution ------------------------------------------------------------------------------------------
    def conv(self, srcs, name, act=ACT_NONE, **kw) -> Act:
        if self.tape is None:
            return super().conv(srcs, name, act, **kw)
        if isinstance(srcs, Act):
            srcs = [srcs]
        residual, pixmul = kw.get("residual"), kw.get("pixmul")
        trainable = name in self.trainable
        if not (trainable or any(s.requires_grad for s in srcs) or (residual is not None and residual.requires_grad)
                or (pixmul is not None and pixmul.requires_grad)):
            return super().conv(srcs, name, act, **kw)
        kw = dict(kw)
        residual, pixmul, out = kw.pop("residual", None), kw.pop("pixmul", None), kw.pop("out", None)
        stride = kw.get("stride", 1)
        fuse_res = residual is not None and act == ACT_NONE
        direct_out = out if (pixmul is None and (residual is None or fuse_res)) else None
        y = super().conv(srcs, name, act, residual=residual if fuse_res else None, out=direct_out, **kw)
        y.mark_grad()
        self.tape.append(lambda: self._conv_backward(srcs, name, act, stride, y, residual if fuse_res else None, trainable))
        res = y
        if residual is not None and not fuse_res:              # act(conv) + residual: the activation needs its own output
            r2 = out if pixmul is None and out is not None else ops.new_act(y.n, y.h, y.w, y.c, device=self.dev)
            ops.copy_channels(y, r2)
            ops.axpy(residual, r2)
            r2.mark_grad()

            def _add_bwd(y=y, r2=r2, residual=residual):
                ops.axpy(r2.grad(), y.grad())
                if residual.requires_grad:
                    ops.axpy(r2.grad(), residual.grad())
            self.tape.append(_add_bwd)
            res = r2
        if pixmul is not None:
            pre = res
            r3 = ops.mul_pix(pre, pixmul, out=out)
            r3.mark_grad()
            self.tape.append(lambda: ops.mul_pix_bwd(r3.grad(), pre, pixmul, pre.grad(), pixmul.grad() if pixmul.requires_grad else None))
            res = r3
        return res

    def _dgrad_pc(self, name: str, si: int, c0: int, c1: int, kind: str) -> ops.PackedConv:
        key = (name, si)
        frozen = name.split("@")[0] not in self.trainable      # constant weights: packed once, not once per step
        pc = (self._dpc_frozen if frozen else self._dpc).get(key)
        if pc is not None:
            return pc
        w = self.sd[name.split("@")[0] + ".weight"].detach().to(torch.float32)
        if w.dim() == 2:                                       # nn.Linear on NHWC == 1x1 conv
            w = w[:, :, None, None]
        if name in self.wscale:                                # weights packed with a folded scale (attention q)
            w = w * self.wscale[name]
        if kind == "s1":          # stride-1 conv: rotate the taps, swap in/out channels
            wd = w[:, c0:c1].permute(1, 0, 2, 3).flip(2, 3)
            pc = pack_conv(wd, None, self.dev)
            if frozen and self.winograd and (self.wino_train & 4) and wd.shape[2] == 3 and wd.shape[0] % 32 == 0 and wd.shape[1] % 8 == 0:
                pc.wino = pack_winograd(wd.contiguous(), self.dev)      # data gradient through a frozen 3x3 layer: Winograd form too
        elif kind == "s2":        # stride-2 conv k3 p1: its data gradient is ConvTranspose2d(k3,s2,p1,op1) with the same tensor
            wt = w[:, c0:c1]
            if wt.shape[1] % 32:
                wt = torch.nn.functional.pad(wt, (0, 0, 0, 0, 0, 32 - wt.shape[1] % 32))
            pc = _pack_convT_dev(wt, self.dev)
        elif kind == "T":         # ConvTranspose2d forward: data gradient = stride-2 conv with the same tensor read as OIHW
            pc = pack_conv(w, None, self.dev)
        elif kind == "dcn":       # 1x1 over the tap-major column tensor
            cout = w.shape[0]
            w1 = w.permute(0, 2, 3, 1).reshape(cout, -1)
            pc = pack_conv(w1.t().reshape(-1, cout, 1, 1), None, self.dev)
        else:
            raise ValueError(kind)
        (self._dpc_frozen if frozen else self._dpc)[key] = pc
        return pc

    def _wino_layer(self, name: str) -> bool:
        return bool(self.wino_train & 1) and name not in getattr(self, "trainable", ())

    def sync_weights(self, live, force=()):
        """Engine.sync_weights + the data-gradient packs: those of frozen layers are kept across steps (`_dpc_frozen`), so a parameter that
        changed behind the engine's back (load_state_dict of a sub-module, the torch.autograd path) must drop them too."""
        n = super().sync_weights(live, force)
        if n:
            self._dpc.clear()
            self._dpc_frozen.clear()
        return n

    def _conv_backward(self, srcs: List[Act], name: str, act: int, stride: int, y: Act, residual: Optional[Act], trainable: bool):
        pc = self.pc[name]
        wkey = name.split("@")[0]
        w = self.sd[wkey + ".weight"]
        if w.dim() == 2:
            w = w[:, :, None, None]
        dY = y.grad()
        ps = pc.pixel_shuffle
        is_dcn = name.endswith("dcnpack")
        if pc.transposed:
            zn, zh, zw, zc = y.n, y.h, y.w, y.c
        elif ps:
            zn, zh, zw, zc = y.n, y.h // 2, y.w // 2, 4 * y.c
        else:
            zn, zh, zw, zc = y.n, y.h, y.w, y.c
        if act != ACT_NONE or ps:
            dZ = ops.act_bwd(dY, y, zn, zh, zw, zc, act, ps)
        else:
            dZ = dY
        if residual is not None and residual.requires_grad:
            ops.axpy(dY, residual.grad())
        sc = self.wscale.get(name)
        if trainable:
            gb = self.gb.get(wkey)
            if gb is not None:
                if sc is None:
                    ops.bias_grad(dZ, gb)
                else:                                          # y = (sc*W) x + sc*b: d/db = sc * sum dZ
                    tmp = torch.zeros_like(gb)
                    ops.bias_grad(dZ, tmp)
                    gb.add_(tmp, alpha=sc)
        if is_dcn:                                            # srcs = [col]; weight [cout][cin][3][3] <-> 1x1 over [9*cin]
            col = srcs[0]
            if trainable:
                cout, cin = w.shape[0], w.shape[1]
                dw1 = torch.zeros(cout, 9 * cin, dtype=torch.float32, device=self.dev)
                ops.conv2d_wgrad(col, dZ, 1, 1, dw1, 9 * cin, 0, tag=name)
                self.gw[wkey] += dw1.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
            if col.requires_grad:
                g = col.grad()
                ops.conv2d([dZ], self._dgrad_pc(name, 0, 0, 0, "dcn"), ACT_NONE, residual=g, out=g, tag=name + ".dgrad")
            return
        if pc.transposed:                                     # y = ConvTranspose(x): weight [Cin][Cout][3][3]
            x = srcs[0]
            if trainable:
                ops.conv2d_wgrad(dZ, x, 3, 2, self.gw[wkey], w.shape[1], 0, tag=name)
            if x.requires_grad:
                g = x.grad()
                ops.conv2d([dZ], self._dgrad_pc(name, 0, 0, 0, "T"), ACT_NONE, stride=2, residual=g, out=g, tag=name + ".dgrad")
            return
        cin_total, k = w.shape[1], w.shape[2]
        c0 = 0
        for si, s in enumerate(srcs):
            ci = min(s.c, cin_total - c0)                     # logical channels of this source (a padded buffer may be wider)
            xs = s if ci == s.c else s.slice(0, ci)
            if trainable:
                if sc is None:
                    ops.conv2d_wgrad(xs, dZ, k, stride, self.gw[wkey], cin_total, c0, tag=name)
                else:
                    tmpw = torch.zeros_like(self.gw[wkey])
                    ops.conv2d_wgrad(xs, dZ, k, stride, tmpw, cin_total, c0, tag=name)
                    self.gw[wkey].add_(tmpw, alpha=sc)
            if s.requires_grad:
                g = xs.grad()
                if stride == 1:
                    pcd = self._dgrad_pc(name, si, c0, c0 + ci, "s1")
                    ops.conv2d([dZ], pcd, ACT_NONE, residual=g, out=g, tag=name + ".dgrad",
                               winograd=pcd.wino is not None and self.precision == "fp32" and self.wino_geometry_ok(dZ, pcd.cout))
                elif stride == 2:
                    assert k == 3 and xs.h == 2 * dZ.h and xs.w == 2 * dZ.w, "stride-2 data gradient needs even input sizes"
                    pcd = self._dgrad_pc(name, si, c0, c0 + ci, "s2")
                    if pcd.cout == ci:
                        ops.conv2d([dZ], pcd, ACT_NONE, residual=g, out=g, tag=name + ".dgrad")
                    else:                                     # channel count padded to 32 for the transposed kernel
                        tmp = ops.conv2d([dZ], pcd, ACT_NONE, tag=name + ".dgrad")
                        ops.axpy(tmp.slice(0, ci), g)
                else:
                    raise NotImplementedError(f"data gradient of a stride-{stride} convolution ({name})")
            c0 += ci
        assert c0 == cin_total, (name, c0, cin_total)

    # -- forward of the SR network with the tape on ------------------------------------------------------------------------
    def forward_train(self, x: torch.Tensor, forced_idx: Optional[torch.Tensor] = None, forced_flow: Optional[torch.Tensor] = None):
        """GPEMSR.forward with recording -> (out Act [B,sH,sW,1], ref_img tensor [B*N,1,sH,sW])."""
        B, N, C, H, W = x.shape
        assert N == self.N and C == 1
        self._check_lr(H, W)
        x = x.to(torch.float32).contiguous()
        xa = Act(x, B * N, H, W, 1, 1, 0)
        pyr, ref_img = self._front_all(xa, forced_idx, None)
        windows = torch.arange(B * N, dtype=torch.int32, device=self.dev).view(B, N)
        if forced_flow is not None:                           # [B,N,2,4H,4W] (reference layout) -> NHWC pairs
            self._forced_flow = ops.from_nhwc(forced_flow.to(torch.float32).reshape(B * N, 2, 4 * H, 4 * W).permute(0, 2, 3, 1).contiguous())
        try:
            self._back(xa, pyr, windows, None)
        finally:
            self._forced_flow = None
        return self._last_out_act, ref_img


def flatten_parameters(named, device):
    """One flat fp32 buffer for the given (name, Parameter) list plus same-sized gradient / Adam-moment buffers; the
    Parameters become views of the first (state_dict() / checkpoints always show the current weights).  Returns
    (flat_p, flat_g, flat_m, flat_v, gw, gb, names): gw / gb map a layer name to the gradient view of its weight / bias."""
    sizes = [(p.numel() + 3) // 4 * 4 for _, p in named]
    total = sum(sizes)
    flat_p = torch.zeros(total, dtype=torch.float32, device=device)
    flat_g = torch.zeros(total, dtype=torch.float32, device=device)
    flat_m = torch.zeros(total, dtype=torch.float32, device=device)
    flat_v = torch.zeros(total, dtype=torch.float32, device=device)
    gw, gb, off, names = {}, {}, 0, set()
    for (k, p), sz in zip(named, sizes):
        view = flat_p[off:off + p.numel()].view(p.shape)
        view.copy_(p.detach().to(device=device, dtype=torch.float32))
        p.data = view
        g = flat_g[off:off + p.numel()].view(p.shape)
        base, leaf = k.rsplit(".", 1)
        names.add(base)
        if leaf == "weight":
            gw[base] = g.view(g.shape[0], g.shape[1]) if g.dim() == 5 else g          # Conv3d [t,t,1,1,1] -> [t,t]
        else:
            gb[base] = g
        off += sz
    return flat_p, flat_g, flat_m, flat_v, gw, gb, names


def contextual_loss_taped(fx: Act, fy: Act, t: int, scale_ref: list, band_width: float, tape: list, dev):
    """contextual_loss(x, y) (model/contextual.py:8-52) for x = every image of ``fx`` repeated t times, y = ``fy`` ([b*t] images).
    Appends to ``tape`` the closure that adds d loss / d fx times ``scale_ref[0]`` (read at backward time) into ``fx.grad()``.
    Returns (loss [1], c [n, Py])."""
    b, n, c = fx.n, fy.n, fx.c
    px, py = fx.h * fx.w, fy.h * fy.w
    if px % 32 or py % 32 or c % 32:
        raise RuntimeError(f"gpemsr_amd.train: contextual loss needs Hx*Wx, Hy*Wy and C to be multiples of 32 (got {px}, {py}, {c})")
    lib = ops._abi.load()
    nblk = (fy.pixels + 1023) // 1024
    ws = torch.empty(nblk * c, dtype=torch.float32, device=dev)
    mu = torch.empty(c, dtype=torch.float32, device=dev)
    ops._abi.check(lib.gpemsr_cx_channel_mean(fy.ptr, fy.pixels, c, fy.ld, ws.data_ptr(), ws.numel(), mu.data_ptr(), ops._stream()), "cx_channel_mean")
    xn = ops.new_act(b, fx.h, fx.w, c, device=dev)
    yn = ops.new_act(n, fy.h, fy.w, c, device=dev)
    ops._abi.check(lib.gpemsr_cx_center_normalize(fx.ptr, mu.data_ptr(), fx.pixels, c, fx.ld, xn.ptr, xn.ld, ops._stream()), "cx_center_normalize")
    ops._abi.check(lib.gpemsr_cx_center_normalize(fy.ptr, mu.data_ptr(), fy.pixels, c, fy.ld, yn.ptr, yn.ld, ops._stream()), "cx_center_normalize")
    xr = ops.copy_images(xn, n, t, 1, 0)                            # SR features repeated for the t reference frames
    sim = ops.conv2d([xr.reshape_hw(px // 32, 32)], ops.PackedConv(yn.buf, None, 1, py, (c,), 32), ACT_NONE,
                     weight_image_stride=py * c, tag="cx.sim")
    simt = sim.buf.view(n, px, py)
    cx = torch.empty_like(simt)
    ops._abi.check(lib.gpemsr_cx_rows(simt.data_ptr(), n * px, py, float(band_width), cx.data_ptr(), ops._stream()), "cx_rows")
    nslab = (px + 127) // 128
    ws2 = torch.empty(2 * n * nslab * py, dtype=torch.float32, device=dev)
    rmax = torch.empty(n, py, dtype=torch.float32, device=dev)
    cw = torch.empty(n, py, dtype=torch.float32, device=dev)
    cxn = torch.empty(n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ops._abi.check(lib.gpemsr_cx_reduce(cx.data_ptr(), simt.data_ptr(), n, px, py, float(band_width), ws2.data_ptr(), ws2.numel(),
                                        rmax.data_ptr(), cw.data_ptr(), cxn.data_ptr(), loss.data_ptr(), ops._stream()), "cx_reduce")
    del ws2

    def _cx_bwd():
        idx = torch.empty(n * py, dtype=torch.int32, device=dev)
        coef = torch.empty(2 * n * py * (1 + (px + 127) // 128), dtype=torch.float32, device=dev)
        dsim = ops.new_act(n, px // 32, 32, py, device=dev)
        ops._abi.check(lib.gpemsr_cx_backward(simt.data_ptr(), cx.data_ptr(), rmax.data_ptr(), cw.data_ptr(), cxn.data_ptr(), n, px, py,
                                              float(band_width), float(scale_ref[0]), idx.data_ptr(), coef.data_ptr(), dsim.ptr,
                                              ops._stream()), "cx_backward")
        ynT = ops.transpose_images(yn)                              # [n][C][Py]: dX^ = dS . Y^ as a 1x1 with per-image weights
        dxr = ops.conv2d([dsim], ops.PackedConv(ynT.buf, None, 1, c, (py,), 32), ACT_NONE, weight_image_stride=c * py, tag="cx.dgrad")
        dxn = ops.new_act(b, fx.h, fx.w, c, device=dev)
        dxn.buf.zero_()
        rep = (torch.arange(n, device=dev) // t).to(torch.int32)
        ops.scatter_add_images(dxr.reshape_hw(fx.h, fx.w), rep, dxn)
        g = fx.grad()
        ops._abi.check(lib.gpemsr_cx_center_normalize_bwd(fx.ptr, mu.data_ptr(), dxn.ptr, fx.pixels, c, fx.ld, dxn.ld, g.ptr, g.ld,
                                                          ops._stream()), "cx_center_normalize_bwd")
    tape.append(_cx_bwd)
    return loss, cw


class _TrainerState:
    """Resume support shared by the trainers: what the reference keeps in its ``training_state`` files
    (train_stage3.py:183-184: optimizer and scheduler state dicts) -- Adam moments, step counter, scheduler position.
    The weights themselves travel in ``model.state_dict()`` as usual."""

    def adam_hparams(self):
        """(beta1, beta2, eps, weight_decay) this trainer's ``step()`` really uses -- ONE place, so that the exported torch state dict
        (``torch_optimizer_state_dict``) can never disagree with the run.  Absent keys take torch.optim.Adam's defaults;
        the reference reads beta1 / beta2 from the option file without a default (R:train_stage3.py:153-158)."""
        o = self.opt
        return float(o.get("beta1", 0.9)), float(o.get("beta2", 0.999)), 1e-8, float(o.get("weight_decay_G") or 0.0)

    def state_dict(self) -> dict:
        names = [k for k, p in self.model.named_parameters() if p.data_ptr() >= self.flat_p.data_ptr()
                 and p.data_ptr() < self.flat_p.data_ptr() + 4 * self.flat_p.numel()]
        return {"step_count": self.step_count, "lr": self.lr, "scheduler": dict(vars(self.sched)),
                "exp_avg": self.flat_m.detach().clone(), "exp_avg_sq": self.flat_v.detach().clone(), "param_names": names}

    def load_state_dict(self, st: dict):
        assert st["exp_avg"].numel() == self.flat_m.numel(), "optimizer state belongs to a different parameter set"
        mine = self.state_dict()["param_names"] if "param_names" in st else None
        assert mine is None or list(st["param_names"]) == mine, "optimizer state was saved for other parameters / another order"
        self.step_count, self.lr = int(st["step_count"]), float(st["lr"])
        for k, v in st["scheduler"].items():
            setattr(self.sched, k, v)
        self.flat_m.copy_(st["exp_avg"].to(self.flat_m.device))
        self.flat_v.copy_(st["exp_avg_sq"].to(self.flat_v.device))
        self.eng.refresh_weights()


    # ---- exchange with the reference's checkpoints (train_stage3.py:183-184 saves optimizer.state_dict() / scheduler.state_dict()) ----
    def _flat_params(self):
        """(name, offset, shape) of every trainable parameter inside the flat buffers, in optimizer order."""
        base, out = self.flat_p.data_ptr(), []
        for k, p in self.model.named_parameters():
            off = (p.data_ptr() - base) // 4
            if 0 <= off < self.flat_p.numel() and p.requires_grad:
                out.append((k, int(off), tuple(p.shape)))
        return out

    def torch_optimizer_state_dict(self) -> dict:
        """The Adam state in ``torch.optim.Adam.state_dict()`` layout (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``, one param
        group over the trainable parameters in ``named_parameters()`` order, which is how train_stage3.py:153-158 builds it), so a run
        can be resumed by the reference and vice versa."""
        state, params = {}, self._flat_params()
        for i, (k, off, shape) in enumerate(params):
            n = 1
            for d in shape:
                n *= d
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.flat_m[off:off + n].detach().clone().view(shape),
                        "exp_avg_sq": self.flat_v[off:off + n].detach().clone().view(shape)}
        b1, b2, eps, wd = self.adam_hparams()
        group = {"lr": float(self.lr), "betas": (b1, b2), "eps": eps,
                 "weight_decay": wd, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "initial_lr": float(self.opt.get("lr_G", self.lr)),
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_torch_optimizer_state_dict(self, sd: dict, scheduler_state: dict = None):
        """Inverse of ``torch_optimizer_state_dict``: accepts what ``torch.optim.Adam.state_dict()`` returns for the same parameter
        list (and optionally the reference scheduler's ``state_dict()``, of which ``last_epoch`` positions this scheduler)."""
        params = self._flat_params()
        assert len(sd["param_groups"]) == 1 and len(sd["param_groups"][0]["params"]) == len(params), "optimizer state for another parameter list"
        steps = set()
        for i, (k, off, shape) in enumerate(params):
            st = sd["state"].get(i)
            if st is None:                   # a parameter torch never stepped (no gradient): moments stay zero
                continue
            assert tuple(st["exp_avg"].shape) == shape, (k, tuple(st["exp_avg"].shape), shape)
            n = st["exp_avg"].numel()
            self.flat_m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(self.flat_m.device, torch.float32))
            self.flat_v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.flat_v.device, torch.float32))
            steps.add(int(float(st["step"])))
        assert len(steps) <= 1, f"per-parameter step counters differ: {sorted(steps)}"
        if steps:
            self.step_count = steps.pop()
        self.lr = float(sd["param_groups"][0]["lr"])
        if scheduler_state is not None and "last_epoch" in scheduler_state:
            e = int(scheduler_state["last_epoch"])
            sc = self.sched
            sc.last_epoch, sc.lr = e, self.lr           # the optimizer's lr IS the scheduler's last value (its recurrence continues from it)
            if hasattr(sc, "last_restart"):             # cosine: which period we are in (the reference's state dict carries these)
                past = [r for r in sc.restarts if r <= e]
                sc.last_restart = int(scheduler_state.get("last_restart", past[-1] if past else 0))
                sc.T_max = int(scheduler_state.get("T_max", sc.T_period[len(past)] if past else sc.T_period[0]))
        self.eng.refresh_weights()


class Stage3Trainer(_TrainerState):
    """``train_EMSR_onestep`` (train_stage3.py:343-366).  ``opt_train`` is the ``train:`` block of
    option/train_stage3_x{8,16}.yml (lr_G, beta1, beta2, lr_scheme, T_period, restarts, restart_weights, eta_min,
    rec_loss_factor, ref_loss_factor, weight_decay_G)."""

    def __init__(self, model, opt_train: dict, device, world: int = 1, band_width: float = 0.5):
        from . import _abi
        _abi.load()
        self.model, self.dev, self.world = model, device, world
        assert all(p.is_cuda for p in model.parameters()), "move the model to the device first (model.to(device))"
        # model.precision "fp32" (default, exact) or "bf16x3": the FORWARD convolutions (incl. the frozen prior / VGG, about two
        # thirds of the step) run on the split-bf16 kernel (fp32-grade, DESIGN_HISTORY.md §3.3); data / weight gradients stay on the f32 pipe
        assert model.precision in ("fp32", "bf16x3", "bf16"), "training supports precision fp32, bf16x3 or bf16 (bf16 data path for the frozen sub-networks)"
        self.opt = dict(opt_train)
        self.band_width = band_width
        # one flat buffer for the trainable parameters (and their gradient / Adam moments): the model's Parameters
        # become views of it, so state_dict() / checkpoints always show the current weights
        named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        self.flat_p, self.flat_g, self.flat_m, self.flat_v, gw, gb, names = flatten_parameters(named, device)
        self.n_params = sum(p.numel() for _, p in named)
        self._param_keys = [k for k, _ in named]
        model._engine = None
        model._train_state = None
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        self.eng = TrainEngine(sd, device, model.scale, model.nframes, model.groups, model.nf, model._dec_nrb, names, gw, gb,
                               precision=model.precision, wino_train=7 if opt_train.get("winograd_frozen") else 0)
        self.gw, self.gb = gw, gb
        if os.environ.get("GPEMSR_FAST_REFRESH", "1") != "0":
            self.eng.enable_fast_refresh(self.flat_p)
        self.step_count = 0
        o = self.opt
        self.lr = float(o.get("lr_G", 4e-4))
        self.sched = make_scheduler(o)

    # -- loss network --------------------------------------------------------------------------------------------------
    def _vgg_relu3_4(self, a: Act) -> Act:
        eng = self.eng
        for sl, idx, kind in _VGG_TO_RELU3_4:
            if kind == "pool":
                a = eng.o.maxpool2(a)
            else:
                a = eng.conv(a, f"vgg.slice{sl}.{idx}@rgb", ACT_RELU)
        return a

    def _vgg_relu3_4_target16(self, a: Act) -> Act:
        """VGG relu3_4 of the constant target frames (R:train_stage3.py:358-359) with bf16 activations: conv1_1 (3 input channels) on
        the fp32 kernel, everything after it on the bf16 kernels of the frozen engine; the features return as fp32."""
        from .packing import pack_conv_bf16
        eng, e16 = self.eng, self.eng._frozen16
        first = True
        for sl, idx, kind in _VGG_TO_RELU3_4:
            if kind == "pool":
                a = ops.maxpool2(a)
            elif first:
                a = ops.cast_bf16(eng.conv(a, f"vgg.slice{sl}.{idx}@rgb", ACT_RELU, precision="fp32"))
                first = False
            else:
                key = f"vgg.slice{sl}.{idx}"
                if (key + "@t16") not in e16.pc:
                    w, b = eng.sd[key + ".weight"], eng.sd[key + ".bias"]
                    pc = pack_conv(w, b, self.dev)
                    pc.wb = pack_conv_bf16(w, self.dev)
                    e16.pc[key + "@t16"] = pc
                a = e16.conv(a, key + "@t16", ACT_RELU)
        return ops.cast_f32(a)

    def _gray3(self, x: Act, taped: bool) -> Act:
        x3 = ops.gray_normalize3(x, VGG_MEAN, VGG_STD)
        if taped and x.requires_grad:
            x3.mark_grad()
            self.eng.tape.append(lambda: ops.gray_normalize3_bwd(x3.grad(), VGG_STD, x.grad()))
        return x3

    def _contextual(self, sr: Act, ref_frames: Act, t: int, scale: float):
        """CX(VGG relu3_4(SR repeated t times), VGG relu3_4(ref frames)) and, on the tape, its gradient w.r.t. SR scaled by
        ``scale`` (= ref_loss_factor).  model/contextual.py:8-52,115-138,216-233."""
        eng = self.eng
        fx = self._vgg_relu3_4(self._gray3(sr, True))                   # [b, h/4, w/4, 256], recorded
        tape, eng.tape = eng.tape, None
        if eng._frozen16 is not None and os.environ.get("GPEMSR_CX_TARGET16", "1") != "0":
            fy = self._vgg_relu3_4_target16(self._gray3(ref_frames, False))     # constant TARGET features on the bf16 data path (5/6 of the
        else:                                                                     # loss network's forward work: t frames per SR image)
            fy = self._vgg_relu3_4(self._gray3(ref_frames, False))      # [b*t, ...], constant
        eng.tape = tape
        return self.contextual_features(fx, fy, t, scale)

    def contextual_features(self, fx: Act, fy: Act, t: int, scale: float):
        """contextual_loss(x, y) (model/contextual.py:8-52) for x = every image of ``fx`` repeated t times, y = ``fy``
        ([b*t] images); appends the gradient w.r.t. ``fx`` (times ``scale``) to the tape."""
        return contextual_loss_taped(fx, fy, t, [float(scale)], self.band_width, self.eng.tape, self.dev)[0]

    # -- one optimisation step -----------------------------------------------------------------------------------------
    def forward_backward(self, LR: torch.Tensor, GT: torch.Tensor, forced_code_idx: Optional[torch.Tensor] = None,
                         forced_flow: Optional[torch.Tensor] = None):
        """Forward, both losses and the backward pass: fills ``flat_g``; returns (rec_loss, ref_loss) device scalars.
        ``forced_code_idx`` teacher-forces the frozen prior's code indices (parity tests; argmax is discontinuous)."""
        if not (LR.is_cuda and GT.is_cuda):
            raise RuntimeError("gpemsr_amd.train: LR/GT must live on a cuda/HIP device (there is no CPU path)")
        eng = self.eng
        self.flat_g.zero_()
        eng.tape = []
        out, ref_img = eng.forward_train(LR, forced_code_idx, forced_flow)
        self.last_sr = out.buf
        B, t = LR.shape[0], LR.shape[1]
        s = eng.scale
        gt = GT.to(torch.float32).contiguous()
        assert gt.numel() == out.buf.numel(), "GT must be [B,1,sH,sW]"
        rec = ops.l1_loss(out.buf, gt, float(self.opt.get("rec_loss_factor", 1.0)), out.grad().buf)
        ref_factor = float(self.opt.get("ref_loss_factor", 0.0))
        if ref_factor != 0.0:
            ref_act = Act(ref_img, B * t, LR.shape[3] * s, LR.shape[4] * s, 1, 1, 0)
            ref = self._contextual(out, ref_act, t, ref_factor)
        else:
            ref = torch.zeros(1, dtype=torch.float32, device=self.dev)
        for fn in reversed(eng.tape):
            fn()
        eng.tape = None
        return rec, ref

    def step(self, LR: torch.Tensor, GT: torch.Tensor, forced_code_idx: Optional[torch.Tensor] = None,
             forced_flow: Optional[torch.Tensor] = None):
        rec, ref = self.forward_backward(LR, GT, forced_code_idx, forced_flow)
        average_gradients(self.flat_g, self.world)                      # DistributedDataParallel: mean over the replicas
        self.step_count += 1
        o = self.opt
        b1, b2, eps, wd = self.adam_hparams()
        ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, b1, b2, eps, wd, self.step_count)
        self.lr = self.sched.step()
        self.eng.refresh_weights()
        self.model.mark_weights_written(self._param_keys)     # the inference engine's packs are stale now (validation!)
        return {"rec_loss": rec, "ref_loss": ref, "lr": self.lr}
