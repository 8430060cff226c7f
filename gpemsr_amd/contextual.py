"""Contextual loss, forward (the `ref_loss` of the stage-3 training step, train_stage3.py:352-359) on the HIP kernels.

Mirror of the reference's ``model/contextual.py``: ``ContextualLoss(vgg, band_width=0.5, loss_type='cosine',
is_CoBi=False, use_vgg=True, vgg_layer='relu3_4')`` with ``forward(x, y) -> (cx_loss, c)`` (model/contextual.py:175-233),
and the functional ``contextual_loss(x, y, band_width, loss_type)`` (:8-52).  Only what the training step uses is built:
the cosine distance (compute_cosine_distance :115-138); 'L1'/'L2' distances and the bilateral variant (is_CoBi) raise.
Under ``torch.no_grad()`` (validation / logging) it returns the value only; with autograd on and ``x.requires_grad`` it is
differentiable w.r.t. ``x`` through gpemsr_amd/autograd.py (what train_stage3.py:359-364 needs).
"""
from __future__ import annotations

import torch

from . import ops

LOSS_TYPES = ['cosine', 'L1', 'L2']
VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def _features_nhwc(t: torch.Tensor) -> ops.Act:
    if not t.is_cuda:
        raise RuntimeError("gpemsr_amd.contextual: inputs must live on a cuda/HIP device (there is no CPU path)")
    return ops.from_nchw(t.to(torch.float32))


def contextual_loss_nhwc(x: ops.Act, y: ops.Act, band_width: float = 0.5):
    """x, y: NHWC feature maps [N, H, W, C] -> (cx_loss [scalar tensor], c [N, 1, Hy, Wy])."""
    assert x.n == y.n and x.c == y.c, "x and y need the same batch and channel counts"
    n, c = x.n, x.c
    px, py = x.h * x.w, y.h * y.w
    if px % 32 or py % 4 or c % 8:
        raise RuntimeError(f"gpemsr_amd.contextual: needs Hx*Wx % 32 == 0, Hy*Wy % 4 == 0, C % 8 == 0 (got {px}, {py}, {c})")
    xn, yn = ops.cx_normalized_pair(x, y)
    # S[n, i, j] = <x^[n, i, :], y^[n, j, :]>: a 1x1 "convolution" of the x^ rows with per-image weights y^ (rows j)
    ck = 32 if c % 32 == 0 else 8
    xa = xn.reshape_hw(px // 32, 32)
    sim = ops.conv2d([xa], ops.PackedConv(yn.buf, None, 1, py, (c,), ck), ops.ACT_NONE, weight_image_stride=py * c, tag="cx.sim")
    loss, cw, _ = ops.cx_from_similarity(sim.buf.view(n, px, py), band_width)
    return loss[0], cw.view(n, 1, y.h, y.w)


def contextual_loss(x: torch.Tensor, y: torch.Tensor, band_width: float = 0.5, loss_type: str = 'cosine'):
    """model/contextual.py:8-52 for NCHW feature tensors on the device."""
    assert loss_type in LOSS_TYPES, f'select a loss type from {LOSS_TYPES}.'
    if loss_type != 'cosine':
        raise NotImplementedError("gpemsr_amd.contextual: only loss_type='cosine' (the training step's) is built")
    with torch.no_grad():
        return contextual_loss_nhwc(_features_nhwc(x), _features_nhwc(y), band_width)


class ContextualLoss(torch.nn.Module):
    """Same constructor and call as the reference class; ``vgg`` is ``GPEMSR.vgg`` (gpemsr_amd.model._VGGFeatures)."""

    def __init__(self, vgg, band_width=0.5, loss_type='cosine', is_CoBi=False, use_vgg=True, vgg_layer='relu3_4'):
        super().__init__()
        assert loss_type in LOSS_TYPES, f'select a loss type from {LOSS_TYPES}.'
        if is_CoBi or loss_type != 'cosine':
            raise NotImplementedError("gpemsr_amd.contextual: only the cosine, non-bilateral loss of train_stage3.py is built")
        self.band_width, self.loss_type, self.is_CoBi = band_width, loss_type, is_CoBi
        if use_vgg:
            self.vgg_model = vgg
            self.vgg_layer = vgg_layer
            self.register_buffer('vgg_mean', torch.tensor([[[0.485]], [[0.456]], [[0.406]]], requires_grad=False))
            self.register_buffer('vgg_std', torch.tensor([[[0.229]], [[0.224]], [[0.225]]], requires_grad=False))

    def forward(self, x, y):
        if torch.is_grad_enabled() and x.requires_grad:
            # training (train_stage3.py:359, loss_total.backward()): same arithmetic behind torch.autograd Functions
            from .autograd import CXForward, Normalize3
            if hasattr(self, 'vgg_model'):
                assert x.shape[1] == 3 and y.shape[1] == 3, 'VGG model takes 3 channel images.'
                mean = [float(v) for v in self.vgg_mean.flatten()]
                std = [float(v) for v in self.vgg_std.flatten()]
                fx = getattr(self.vgg_model(Normalize3.apply(x, mean, std)), self.vgg_layer)
                with torch.no_grad():
                    fy = self.vgg_model.features_nhwc(ops.normalize3(_features_nhwc(y), mean, std), self.vgg_layer).nchw()
            else:
                fx, fy = x, y.detach()
            return CXForward.apply(fx, fy, self.band_width)
        with torch.no_grad():
            if hasattr(self, 'vgg_model'):
                assert x.shape[1] == 3 and y.shape[1] == 3, 'VGG model takes 3 channel images.'
                mean = [float(v) for v in self.vgg_mean.flatten()]
                std = [float(v) for v in self.vgg_std.flatten()]
                fx = self.vgg_model.features_nhwc(ops.normalize3(_features_nhwc(x), mean, std), self.vgg_layer)
                fy = self.vgg_model.features_nhwc(ops.normalize3(_features_nhwc(y), mean, std), self.vgg_layer)
            else:
                fx, fy = _features_nhwc(x), _features_nhwc(y)
            return contextual_loss_nhwc(fx, fy, self.band_width)
