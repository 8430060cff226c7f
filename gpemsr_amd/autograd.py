"""torch.autograd bridge: the reference's OWN training script (train_stage3.py:343-366 -- ``SR, ref = model(LR)``,
``CLoss = ContextualLoss(model.vgg)``, ``loss_total.backward()``, ``optimizer_G.step()`` with torch's Adam, optionally under
DistributedDataParallel) runs unchanged on the HIP kernels.  Three ``torch.autograd.Function``s hand autograd the tape of
gpemsr_amd/train.py:

  * ``GPEMSR.forward`` with grad enabled      -> ``SRForward``:   forward_train; backward = the recorded tape, parameter
                                                 gradients are returned to autograd (they land in ``p.grad`` / DDP buckets);
  * ``model.vgg(x)`` on a tensor needing grad -> ``VGGForward``:  the five taps, backward = data gradients of the frozen convs;
  * ``ContextualLoss`` on features needing grad -> ``CXForward``: loss value, backward = gpemsr_cx_backward + GEMM.

``Stage3Trainer`` (train.py) remains the fast path: one flat-buffer Adam launch, SR features computed once instead of t
times, no NCHW<->NHWC copies at the Function boundaries.  This module is the drop-in path.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops
from .ops import ACT_RELU, Act
from .train import TrainEngine, contextual_loss_taped


class TrainState:
    """Per-model training state for the autograd path: a TrainEngine over the live Parameters (their storage is shared, so
    optimizer updates are seen; packed copies are refreshed when a Parameter's version counter moved) and one flat buffer
    the gradient kernels accumulate into."""

    def __init__(self, model, device):
        self.named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in self.named]
        self.flat_g = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
        gw: Dict[str, torch.Tensor] = {}
        gb: Dict[str, torch.Tensor] = {}
        self.views: List[torch.Tensor] = []
        names, off = set(), 0
        for (k, p), sz in zip(self.named, sizes):
            g = self.flat_g[off:off + p.numel()].view(p.shape)
            self.views.append(g)
            base, leaf = k.rsplit(".", 1)
            names.add(base)
            if leaf == "weight":
                gw[base] = g.view(g.shape[0], g.shape[1]) if g.dim() == 5 else g
            else:
                gb[base] = g
            off += sz
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        self.eng = TrainEngine(sd, device, model.scale, model.nframes, model.groups, model.nf, model._dec_nrb, names, gw, gb,
                               precision=model.precision)
        self.versions = self._versions()

    def _versions(self):
        return [p._version for _, p in self.named]

    def sync_weights(self):
        v = self._versions()
        if v != self.versions:
            self.eng.refresh_weights()
            self.versions = v

    def params(self):
        return [p for _, p in self.named]


class SRForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state: TrainState, x: torch.Tensor, *params):
        eng = state.eng
        state.sync_weights()
        eng.tape = []
        try:
            out_act, ref_img = eng.forward_train(x)
            ctx.tape = eng.tape
        finally:
            eng.tape = None
        ctx.state, ctx.out_act = state, out_act
        B, N = x.shape[0], x.shape[1]
        ref = ref_img.view(B, N, 1, ref_img.shape[-2], ref_img.shape[-1])
        ctx.mark_non_differentiable(ref)
        return out_act.buf, ref

    @staticmethod
    def backward(ctx, g_out, g_ref):
        state = ctx.state
        state.flat_g.zero_()
        g = ctx.out_act.grad()
        g.buf.copy_(g_out.contiguous().view(-1))
        for fn in reversed(ctx.tape):
            fn()
        ctx.tape = None
        return (None, None) + tuple(v.clone() for v in state.views)


_VGG_LAYERS = (
    (1, 0, "conv"), (1, 2, "conv"),
    (2, 4, "pool"), (2, 5, "conv"), (2, 7, "conv"),
    (3, 9, "pool"), (3, 10, "conv"), (3, 12, "conv"), (3, 14, "conv"), (3, 16, "conv"),
    (4, 18, "pool"), (4, 19, "conv"), (4, 21, "conv"), (4, 23, "conv"), (4, 25, "conv"),
    (5, 27, "pool"), (5, 28, "conv"), (5, 30, "conv"), (5, 32, "conv"), (5, 34, "conv"),
)


class VGGForward(torch.autograd.Function):
    """``VGG19.forward`` (model/VGG.py:34-52) -> (relu1_2, relu2_2, relu3_4, relu4_4, relu5_4), differentiable w.r.t. x."""

    @staticmethod
    def forward(ctx, state: TrainState, x: torch.Tensor):
        from .packing import pack_conv
        eng = state.eng
        a = ops.from_nchw(x.detach().to(torch.float32)).mark_grad()
        ctx.x_act = a
        eng.tape = []
        taps: List[Act] = []
        try:
            cur = 1
            for sl, idx, kind in _VGG_LAYERS:
                if sl != cur:
                    taps.append(a); cur = sl
                if kind == "pool":
                    a = eng.o.maxpool2(a)
                else:
                    name = f"vgg.slice{sl}.{idx}@rgb"
                    if name not in eng.pc:
                        key = f"vgg.slice{sl}.{idx}"
                        eng.pc[name] = pack_conv(eng.sd[key + ".weight"], eng.sd[key + ".bias"], eng.dev)
                    a = eng.conv(a, name, ACT_RELU)
            taps.append(a)
            ctx.tape = eng.tape
        finally:
            eng.tape = None
        ctx.taps = taps
        return tuple(t.nchw() for t in taps)

    @staticmethod
    def backward(ctx, *gtaps):
        for t, g in zip(ctx.taps, gtaps):
            if g is not None:
                ops.axpy(ops.from_nchw(g.to(torch.float32)), t.grad())
        for fn in reversed(ctx.tape):
            fn()
        ctx.tape = None
        return None, ctx.x_act.grad().nchw()


class Normalize3(torch.autograd.Function):
    """(x - mean_c) / std_c on a 3-channel NCHW batch (ContextualLoss.forward, model/contextual.py:222-224)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, mean3, std3):
        ctx.std3 = tuple(float(v) for v in std3)
        return ops.normalize3(ops.from_nchw(x.detach().to(torch.float32)), mean3, std3).nchw()

    @staticmethod
    def backward(ctx, g):
        return ops.normalize3(ops.from_nchw(g.to(torch.float32)), (0.0, 0.0, 0.0), ctx.std3).nchw(), None, None


class CXForward(torch.autograd.Function):
    """``contextual_loss(x, y, band_width, 'cosine')`` (model/contextual.py:8-52) -> (cx_loss, c), differentiable w.r.t. x."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, y: torch.Tensor, band_width: float):
        fx = ops.from_nchw(x.detach().to(torch.float32)).mark_grad()
        fy = ops.from_nchw(y.detach().to(torch.float32))
        ctx.scale, ctx.tape, ctx.fx = [1.0], [], fx
        loss, cw = contextual_loss_taped(fx, fy, 1, ctx.scale, float(band_width), ctx.tape, x.device)
        c = cw.view(x.shape[0], 1, y.shape[2], y.shape[3])
        ctx.mark_non_differentiable(c)
        return loss[0], c

    @staticmethod
    def backward(ctx, g_loss, g_c):
        ctx.scale[0] = float(g_loss)
        for fn in reversed(ctx.tape):
            fn()
        ctx.tape = None
        return ctx.fx.grad().nchw(), None, None
