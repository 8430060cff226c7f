"""PatchGAN discriminator of stage-1 (VQGAN) training on the HIP kernels -- drop-in for R:model/discriminator.py:9-32
(``Discriminator(args)``: same constructor argument dict, same state-dict keys ``model.{0,2,5,8,11}.weight`` / ``model.{0,11}.bias``,
``forward(x[B,1,H,W]) -> [B,1,h',w']``) plus the pieces of the adversarial training step (R:train_stage1.py:300-345) that involve it:

  * forward with saved activations, backward to the input image (generator's GAN loss) and to the weights (discriminator loss);
  * the R1 penalty (R:train_stage1.py:360-372): the gradient of  c * mean_b |d sum(D(x)) / dx|^2  with respect to D's weights, i.e. a
    gradient of a gradient.  D is  conv -> [InstanceNorm] -> LeakyReLU(0.2)  five times over; its backward pass is itself a chain of
    linear maps (col2im, the weight GEMM, the LeakyReLU mask) and the InstanceNorm backward operator, so the second-order pass is:
      (a) a sweep over that backward chain in REVERSE (first layer to last) carrying G = dL/d(backward result): the adjoint of col2im is
          im2col, of  dcol = dz W  it is  G_dz = G_dcol W^T  (+ the weight term dz^T G_dcol: a 1x1 wgrad), of the mask the mask, of the
          InstanceNorm backward operator itself again (it is symmetric) -- plus, at every InstanceNorm, the gradient with respect to the
          layer's FORWARD input (gpemsr_instnorm_bwd_bwd), because that operator is built from the forward activations;
      (b) an ordinary backward pass that starts from those forward-input gradients.

Every 4x4 convolution is im2col + the 1x1 form of gpemsr_conv2d (fp32 MFMA) / gpemsr_conv2d_wgrad (csrc/stage1_adv.hip).  torch holds the
parameters and permutes the tiny weight tensors between the reference's OIHW layout and the GEMM layout; there is no CPU path."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, Act
from .packing import pack_conv

SLOPE = 0.2


def layer_specs(args: dict):
    """[(state-dict index, cin, cout, stride, bias, instnorm, lrelu)] of R:model/discriminator.py:13-30."""
    ic, nf, nl = int(args['im_channel']), int(args['num_filters_last']), int(args['n_layers'])
    specs = [(0, ic, nf, 2, True, False, True)]
    mult, idx = 1, 2
    for i in range(1, nl + 1):
        last, mult = mult, min(2 ** i, 8)
        specs.append((idx, nf * last, nf * mult, 2 if i < nl else 1, False, True, True))
        idx += 3
    specs.append((idx, nf * mult, 1, 1, True, False, False))
    return specs


class Discriminator(nn.Module):
    def __init__(self, args, init_seed: int = 0):
        super().__init__()
        self.args = dict(args)
        self.specs = layer_specs(args)
        self.model = nn.Module()
        g = torch.Generator().manual_seed(1234 + init_seed)
        for idx, cin, cout, stride, bias, inorm, lrelu in self.specs:
            m = nn.Module()
            bound = 1.0 / (cin * 16) ** 0.5                                # nn.Conv2d's default init range
            m.register_parameter("weight", nn.Parameter((torch.rand(cout, cin, 4, 4, generator=g) * 2 - 1) * bound))
            if bias:
                m.register_parameter("bias", nn.Parameter((torch.rand(cout, generator=g) * 2 - 1) * bound))
            self.model.add_module(str(idx), m)
        self._engine = None

    def engine(self, device) -> "DiscEngine":
        if self._engine is None or self._engine.dev != device:
            self._engine = DiscEngine({k: v for k, v in self.named_parameters()}, self.specs, device)
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("gpemsr_amd.Discriminator: input must live on a cuda/HIP device (there is no CPU path)")
        with torch.no_grad():
            eng = self.engine(x.device)
            eng.repack()
            b, c, h, w = x.shape
            xa = ops.from_nchw(x.to(torch.float32))
            out, _ = eng.forward(xa, save=False)
            return out.nchw()[:, :1].contiguous()


class DiscEngine:
    """Forward / backward / second-order pass of the discriminator over packed GEMM weights.  ``params``: name -> tensor (reference layout,
    e.g. views of a trainer's flat buffer); ``repack()`` after they change."""

    def __init__(self, params: Dict[str, torch.Tensor], specs, device):
        self.p, self.specs, self.dev = params, specs, device
        self.pc_f: List[ops.PackedConv] = []
        self.pc_d: List[ops.PackedConv] = []
        self.repack()

    # -- weights ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _kp(cin: int) -> int:
        return (16 * cin + 31) // 32 * 32

    def _w2(self, idx: int, cin: int, cout: int) -> torch.Tensor:
        """OIHW [cout][cin][4][4] -> GEMM rows [cout_pad4][kp], k = (ky*4 + kx)*cin + ci."""
        w = self.p[f"model.{idx}.weight"].detach().to(torch.float32)
        w2 = torch.zeros((cout + 3) // 4 * 4, self._kp(cin), dtype=torch.float32, device=w.device)
        w2[:cout, :16 * cin] = w.permute(0, 2, 3, 1).reshape(cout, 16 * cin)
        return w2

    def repack(self):
        self.pc_f, self.pc_d = [], []
        for idx, cin, cout, stride, bias, inorm, lrelu in self.specs:
            w2 = self._w2(idx, cin, cout)
            cp = w2.shape[0]
            b = None
            if bias:
                b = torch.zeros(cp, dtype=torch.float32, device=w2.device)
                b[:cout] = self.p[f"model.{idx}.bias"].detach().to(torch.float32)
            self.pc_f.append(pack_conv(w2.view(cp, -1, 1, 1), b, self.dev))
            self.pc_d.append(pack_conv(w2.t().contiguous().view(-1, cp, 1, 1), None, self.dev))

    # -- forward ---------------------------------------------------------------------------------------------------------
    def forward(self, x: Act, save: bool = True):
        """x [B,H,W,1] -> (D(x) as Act [B,h',w',4] (channel 0 is the prediction, 1-3 are zero), saved activations)."""
        saved = []
        h = x
        for li, (idx, cin, cout, stride, bias, inorm, lrelu) in enumerate(self.specs):
            col = ops.im2col4(h, stride, self._kp(cin))
            z = ops.conv2d([col], self.pc_f[li], ACT_NONE, tag=f"disc.{idx}", precision="fp32")
            rec = {"x": h, "col": col if save else None, "z": z, "mr": None, "n": None, "y": None}
            t = z
            if inorm:
                t, rec["mr"] = ops.instnorm(z)
                rec["n"] = t
            if lrelu:
                t = ops.lrelu_slope(t, SLOPE)
                rec["y"] = t
            saved.append(rec)
            h = t
        return h, saved

    # -- backward --------------------------------------------------------------------------------------------------------
    def backward(self, saved, d_out: Optional[Act], want_dx: bool, gw: Optional[Dict[str, torch.Tensor]], extra_dz: Optional[dict] = None,
                 keep: Optional[list] = None) -> Optional[Act]:
        """Backward from d_out (gradient of the loss w.r.t. D's output Act, 4 channels; None = zero) down to the input image.
        gw: name -> gradient tensor in the reference layout, ACCUMULATED into (None: no weight gradients).  extra_dz: layer -> gradient
        injected at that layer's convolution output (the second-order pass).  keep: filled with every layer's intermediate gradients."""
        dy = d_out
        dx = None
        for li in range(len(self.specs) - 1, -1, -1):
            idx, cin, cout, stride, bias, inorm, lrelu = self.specs[li]
            rec = saved[li]
            k = {"dy": dy, "dn": None, "dz": None}
            dz = dy
            if dz is not None and lrelu:
                dz = ops.lrelu_slope_bwd(dz, rec["y"], SLOPE)
            k["dn"] = dz
            if dz is not None and inorm:
                dz = ops.instnorm_bwd(rec["z"], rec["mr"], dz)
            if extra_dz is not None and li in extra_dz:
                if dz is None:
                    dz = extra_dz[li]
                else:
                    ops.axpy(extra_dz[li], dz)
            k["dz"] = dz
            if keep is not None:
                keep.insert(0, k)
            if dz is None:
                dy = None
                continue
            if gw is not None:
                self._wgrad(li, rec["col"], dz, gw)
            if li == 0 and not want_dx:
                break
            dcol = ops.conv2d([dz], self.pc_d[li], ACT_NONE, tag=f"disc.{idx}.dgrad", precision="fp32")
            xin = rec["x"]
            dprev = ops.new_act(xin.n, xin.h, xin.w, xin.c, device=self.dev)
            ops.col2im4(dcol, xin.h, xin.w, cin, stride, dprev, accumulate=False)
            dy = dprev
            dx = dprev
        return dx if want_dx else None

    def _wgrad(self, li: int, col: Act, dz: Act, gw: Dict[str, torch.Tensor]):
        idx, cin, cout, stride, bias, inorm, lrelu = self.specs[li]
        cp, kp = dz.c, col.c
        dw2 = torch.zeros(cp, kp, dtype=torch.float32, device=self.dev)
        ops.conv2d_wgrad(col, dz, 1, 1, dw2.view(cp, kp, 1, 1), kp, 0, tag=f"disc.{idx}.wgrad")
        gw[f"model.{idx}.weight"].add_(dw2[:cout, :16 * cin].reshape(cout, 4, 4, cin).permute(0, 3, 1, 2))
        if bias and f"model.{idx}.bias" in gw:
            db = torch.zeros(cp, dtype=torch.float32, device=self.dev)
            ops.bias_grad(dz, db)
            gw[f"model.{idx}.bias"].add_(db[:cout])

    # -- R1 penalty ------------------------------------------------------------------------------------------------------
    def r1_penalty(self, x: Act, scale: float, gw: Dict[str, torch.Tensor], fwd=None) -> torch.Tensor:
        """penalty = mean_b sum_pixels (d sum(D(x)) / dx)^2 (R:train_stage1.py:360-372); gw += d(scale * penalty)/d(weights).  Returns the
        penalty (device scalar).  ``fwd``: (output, saved activations) of a forward pass over the same x, to reuse."""
        B = x.n
        out, saved = fwd if fwd is not None else self.forward(x, save=True)
        ones = ops.new_act(out.n, out.h, out.w, out.c, device=self.dev, zero=True)
        ones.torch().view(-1, out.c)[:, 0] = 1.0                             # d sum(D) / d D: channel 0 only (1-3 are padding)
        keep: list = []
        g = self.backward(saved, ones, True, None, keep=keep)                 # g = d sum(D(x)) / dx
        penalty = ops.sum_scaled(g.torch(), 1.0 / B, square=True)
        # (a) reverse sweep over the backward chain: G = dL/d(backward result), L = scale * penalty
        G = ops.new_act(g.n, g.h, g.w, g.c, device=self.dev, zero=True)
        ops.axpy(g, G, 2.0 * scale / B)
        extra = {}
        for li, (idx, cin, cout, stride, bias, inorm, lrelu) in enumerate(self.specs):
            rec, k = saved[li], keep[li]
            gcol = ops.im2col4(G, stride, self._kp(cin))                      # adjoint of col2im
            # weight term of dcol = dz . W2:  dL/dW2 = dz^T . gcol
            self._wgrad(li, gcol, k["dz"], {f"model.{idx}.weight": gw[f"model.{idx}.weight"]})
            gdz = ops.conv2d([gcol], self._pc_nobias(li), ACT_NONE, tag=f"disc.{idx}.r1", precision="fp32")      # adjoint: gcol . W2^T
            gdn = gdz
            if inorm:
                ez = ops.new_act(rec["z"].n, rec["z"].h, rec["z"].w, rec["z"].c, device=self.dev)
                gdn = ops.instnorm_bwd_bwd(rec["z"], rec["mr"], k["dn"], gdz, ez, accumulate_gx=False)
                extra[li] = ez
            G = ops.lrelu_slope_bwd(gdn, rec["y"], SLOPE) if lrelu else gdn     # the mask is its own adjoint
        # (b) ordinary backward from the forward-input gradients of the InstanceNorm layers
        if extra:
            self._backward_from(saved, extra, gw)
        return penalty

    def _backward_from(self, saved, extra: dict, gw):
        """backward with zero output gradient and seeds at the convolution outputs of ``extra``'s layers."""
        dy = None
        for li in range(max(extra), -1, -1):
            idx, cin, cout, stride, bias, inorm, lrelu = self.specs[li]
            rec = saved[li]
            dz = dy
            if dz is not None and lrelu:
                dz = ops.lrelu_slope_bwd(dz, rec["y"], SLOPE)
            if dz is not None and inorm:
                dz = ops.instnorm_bwd(rec["z"], rec["mr"], dz)
            if li in extra:
                if dz is None:
                    dz = extra[li]
                else:
                    ops.axpy(extra[li], dz)
            if dz is None:
                continue
            wkeys = {f"model.{idx}.weight": gw[f"model.{idx}.weight"]}
            if bias and f"model.{idx}.bias" in gw:
                wkeys[f"model.{idx}.bias"] = gw[f"model.{idx}.bias"]
            self._wgrad(li, rec["col"], dz, wkeys)
            if li == 0:
                break
            dcol = ops.conv2d([dz], self.pc_d[li], ACT_NONE, tag=f"disc.{idx}.dgrad", precision="fp32")
            xin = rec["x"]
            dprev = ops.new_act(xin.n, xin.h, xin.w, xin.c, device=self.dev)
            ops.col2im4(dcol, xin.h, xin.w, cin, stride, dprev, accumulate=False)
            dy = dprev

    def _pc_nobias(self, li: int) -> ops.PackedConv:
        pc = self.pc_f[li]
        if pc.b is None:
            return pc
        if not hasattr(self, "_nb"):
            self._nb = {}
        if li not in self._nb or self._nb[li].w is not pc.w:
            q = ops.PackedConv(pc.w, None, pc.ksize, pc.cout, pc.splits, pc.ck)
            self._nb[li] = q
        return self._nb[li]
