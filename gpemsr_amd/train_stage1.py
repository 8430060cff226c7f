"""Stage-1 (VQGAN) training: ``train_vqgan_onestep`` of R:train_stage1.py:291-357.

GENERATOR PHASE, ``current_step <= gan_start`` (the first 40,000 steps with option/train_stage1.yml; lines 313-326):

    decoded, _, q_loss = generator(imgs)                              # model/vqgan.py:24-28: Encoder -> Codebook -> Decoder
    vq_loss = rec_loss_factor * L1(imgs, decoded) + codebook_loss_factor * q_loss
    vq_loss.backward();  optimizer_G.step();  scheduler_G.step()

ADVERSARIAL PHASE, ``current_step > gan_start`` (lines 300-312, 330-357; needs ``Stage1Trainer(..., discriminator=...)``):

    vq_loss += gan_loss_factor * mean(-D(decoded))                    # generator step every generator_update_rate steps, D frozen
    d_loss = 0.5 * (mean(-D(imgs)) + mean(D(decoded.detach())));  d_loss.backward()
    every net_d_reg_every steps:  (r1_reg_weight / 2 * net_d_reg_every * r1_penalty(D(imgs), imgs)).backward()     # lines 339-345, 360-372
    optimizer_D.step();  scheduler_D.step()

D = the PatchGAN discriminator of R:model/discriminator.py:9-32 on the HIP kernels (gpemsr_amd/discriminator.py: im2col + fp32 MFMA GEMM,
InstanceNorm, LeakyReLU(0.2), their backward passes and the second-order pass of the R1 penalty).

on the HIP kernels: the recorded GroupNorm / residual block / non-local block / down-block of the stage-2 engine
(gpemsr_amd/train_stage2.py) plus the transposed-convolution backward of the stage-3 engine cover Encoder and Decoder;
``gpemsr_vq_codebook_loss`` is the codebook's loss, its two gradients and the straight-through estimator's value
(model/codebook.py:20-31); Adam and the cosine schedule are the trainers' common ones.

The generator lives inside the stage-3 model as ``refmodel.{encoder,codebook,decoder}`` (same blocks, same keys behind the prefix:
``Generator.load_state_dict`` takes them with strict=True, oracle/gen_golden_stage1.py does exactly that), so the trainer works on a
``gpemsr_amd.GPEMSR`` whose prior it trains; ``generator_state_dict()`` returns the reference's ``generator.state_dict()`` layout.

"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import ops
from .dist import average_gradients
from .engine import _seq_len
from .ops import ACT_NONE, ACT_RELU, Act
from .packing import pack_conv
from .train import CosineAnnealingLRRestart, MultiStepLRRestart, _TrainerState, flatten_parameters
from .train_stage2 import Stage2Engine

_GEN_PREFIXES = ("refmodel.encoder.", "refmodel.codebook.", "refmodel.decoder.")


class Stage1Engine(Stage2Engine):
    """Encoder / Decoder with the tape on (every layer trainable) and the codebook in between."""

    def vq_layer_train(self, x: Act, p: str) -> Act:
        if (p + ".upblock") in self.pc:
            return self.conv(x, p + ".upblock")                 # ConvTranspose2d(k3 s2 p1 op1): recorded by TrainEngine.conv
        return super().vq_layer_train(x, p)

    def encoder_train(self, img: Act) -> Act:
        """model/encoder.py:36-39 -> z [n, h/16, w/16, latent_dim]"""
        p = "refmodel.encoder"
        h = self.conv(img, p + ".input_layer.0", ACT_RELU)
        for i in range(_seq_len(self.sd, p + ".feat_extract")):
            h = self.vq_layer_train(h, f"{p}.feat_extract.{i}")
        for i in range(_seq_len(self.sd, p + ".output_layer")):
            h = self.vq_layer_train(h, f"{p}.output_layer.{i}")
        return h

    def nearest_codes(self, z: Act) -> torch.Tensor:
        """argmin_k |z - e_k|^2 = argmax_k (z . e_k - |e_k|^2 / 2) (model/codebook.py:20-25); the table changes every step, so the
        packed scoring weights are rebuilt from the live embedding."""
        E = self.par["refmodel.codebook.embedding.weight"]
        pc = pack_conv(E.view(E.shape[0], E.shape[1], 1, 1), -0.5 * (E * E).sum(dim=1), self.dev)
        tape, self.tape = self.tape, None
        try:
            return ops.argmax_rows(ops.conv2d([z], pc, ACT_NONE, tag="codebook.nearest"))
        finally:
            self.tape = tape

    def decoder_train(self, zq: Act) -> Act:
        """model/decoder.py forward -> decoded image [n, H, W, 1]"""
        p = "refmodel.decoder"
        x = zq
        for i in range(_seq_len(self.sd, p + ".input_layer")):
            x = self.vq_layer_train(x, f"{p}.input_layer.{i}")
        for i in range(_seq_len(self.sd, p + ".feat_extract")):
            x = self.vq_layer_train(x, f"{p}.feat_extract.{i}")
        return self.conv(x, p + ".output_layer")


class Stage1Trainer(_TrainerState):
    """One step of stage 1.  ``opt_train``: the ``train:`` block of option/train_stage1.yml (lr_G, lr_D, beta1, beta2, lr_scheme,
    T_period, restarts, restart_weights, eta_min, rec_loss_factor, codebook_loss_factor, gan_start, gan_loss_factor,
    generator_update_rate, r1_reg_weight, net_d_reg_every); ``beta``: the codebook's commitment weight
    (network.Generator.Codebook.beta); ``discriminator``: a ``gpemsr_amd.discriminator.Discriminator`` on the device (needed once
    ``current_step > gan_start``).

    ``step_count`` counts optimizer_G steps (Adam's bias correction and the scheduler position, as torch keeps them);
    ``current_step`` is the training loop's counter (R:train_stage1.py:268-271), which decides the phase; it advances by one per
    ``step()`` unless given."""

    def __init__(self, model, opt_train: dict, device, beta: float = 1.0, world: int = 1, discriminator=None):
        from . import _abi
        _abi.load()
        assert all(p.is_cuda for p in model.parameters()), "move the model to the device first (model.to(device))"
        assert model.precision == "fp32"
        self.model, self.dev, self.world, self.beta = model, device, world, float(beta)
        self.opt = dict(opt_train)
        named = [(k, p) for k, p in model.named_parameters() if k.startswith(_GEN_PREFIXES)]          # train_stage1.py:164-170
        self.flat_p, self.flat_g, self.flat_m, self.flat_v, gw, gb, names = flatten_parameters(named, device)
        self.n_params = sum(p.numel() for _, p in named)
        self._param_keys = [k for k, _ in named]
        model._engine = None
        model._train_state = None
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        self.eng = Stage1Engine(sd, device, model.scale, model.nframes, model.groups, model.nf, model._dec_nrb, names, gw, gb)
        self.gw, self.gb = gw, gb
        if os.environ.get("GPEMSR_FAST_REFRESH", "1") != "0":     # the per-step repack as ONE gather from the flat buffer (TrainEngine.enable_fast_refresh)
            self.eng.enable_fast_refresh(self.flat_p)
        self.step_count = 0
        self.current_step = 0
        o = self.opt

        def scheduler(lr0):                                                                          # train_stage1.py:171-190
            if o.get("lr_scheme", "CosineAnnealingLR_Restart") == "MultiStepLR":
                return MultiStepLRRestart(lr0, o["lr_steps"], o.get("restarts"), o.get("restart_weights"), o.get("lr_gamma", 0.1))
            return CosineAnnealingLRRestart(lr0, o.get("T_period", [1 << 30]), o.get("restarts"), o.get("restart_weights"), o.get("eta_min", 0.0))
        self.lr = float(o.get("lr_G", 4e-4))
        self.sched = scheduler(self.lr)
        self.disc = discriminator
        if discriminator is not None:
            from .discriminator import DiscEngine
            assert all(p.is_cuda for p in discriminator.parameters()), "move the discriminator to the device first"
            named_d = list(discriminator.named_parameters())                                         # train_stage1.py:160-163
            self.d_flat_p, self.d_flat_g, self.d_flat_m, self.d_flat_v, dgw, dgb, _ = flatten_parameters(named_d, device)
            self.d_gw = {**{k + ".weight": v for k, v in dgw.items()}, **{k + ".bias": v for k, v in dgb.items()}}
            self.deng = DiscEngine({k: p for k, p in named_d}, discriminator.specs, device)
            discriminator._engine = self.deng
            self.d_steps = 0
            self.lr_d = float(o.get("lr_D", 4e-4))
            self.sched_d = scheduler(self.lr_d)

    def generator_state_dict(self) -> dict:
        """``generator.state_dict()`` of the reference (model/vqgan.py:16-22): encoder.* / codebook.* / decoder.* keys."""
        return {k[len("refmodel."):]: v.detach().clone() for k, v in self.model.state_dict().items() if k.startswith(_GEN_PREFIXES)}

    def _adversarial(self, current_step: int) -> bool:
        adv = current_step > int(self.opt.get("gan_start", 1 << 60))
        if adv and self.disc is None:
            raise RuntimeError("gpemsr_amd.train_stage1: current_step > gan_start needs Stage1Trainer(..., discriminator=Discriminator(...))")
        return adv

    def _d_seed(self, out: Act, value: float) -> Act:
        """Gradient of  value * sum(D(x))  with respect to D's output tensor (channel 0 is the prediction; 1-3 are padding)."""
        d = ops.new_act(out.n, out.h, out.w, out.c, device=self.dev, zero=True)
        d.torch().view(-1, out.c)[:, 0] = value
        return d

    def forward_backward(self, imgs: torch.Tensor, forced_idx: Optional[torch.Tensor] = None, current_step: Optional[int] = None,
                         backward: bool = True):
        """-> (rec_loss, q_loss device scalars, code indices int32 [B*h*w]); fills ``flat_g`` with d(vq_loss)/d(parameters), the GAN
        term included when ``current_step > gan_start`` (``self.last_g_loss``).  ``forced_idx`` teacher-forces the arg-min (parity
        tests); ``backward=False``: forward only (the steps generator_update_rate skips, R:train_stage1.py:327-328)."""
        if not imgs.is_cuda:
            raise RuntimeError("gpemsr_amd.train_stage1: inputs must live on a cuda/HIP device (there is no CPU path)")
        step_no = self.current_step + 1 if current_step is None else int(current_step)
        adv = self._adversarial(step_no)
        eng, o = self.eng, self.opt
        self.flat_g.zero_()
        x = imgs.to(torch.float32).contiguous()
        assert x.dim() == 4 and x.shape[1] == 1, "expected [B,1,H,W] images"
        B, _, H, W = x.shape
        eng.tape = []
        z = eng.encoder_train(Act(x, B, H, W, 1, 1, 0))
        assert z.ld == z.c
        idx = eng.nearest_codes(z)
        self.last_idx = idx
        if forced_idx is not None:
            idx = forced_idx.to(device=self.dev, dtype=torch.int32).contiguous()
        assert idx.numel() == z.pixels
        E = eng.par["refmodel.codebook.embedding.weight"]
        zq = ops.new_act(z.n, z.h, z.w, z.c, device=self.dev)
        q_loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        ws = ops._workspace(1024, self.dev)
        ops._abi.check(ops._abi.load().gpemsr_vq_codebook_loss(
            z.ptr, z.ld, E.data_ptr(), idx.data_ptr(), z.pixels, z.c, self.beta, float(o.get("codebook_loss_factor", 1.0)),
            z.grad().ptr, z.ld, self.gw["refmodel.codebook.embedding"].data_ptr(), zq.ptr, zq.ld, ws.data_ptr(), ws.numel(),
            q_loss.data_ptr(), ops._stream()), "vq_codebook_loss")
        zq.mark_grad()
        eng.tape.append(lambda: ops.axpy(zq.grad(), z.grad()))          # straight-through: d/dz of z + (zq - z).detach()
        dec = eng.decoder_train(zq)
        self.last_decoded = dec
        rec = ops.l1_loss(dec.buf, x, float(o.get("rec_loss_factor", 1.0)), dec.grad().buf)
        self.last_g_loss = None
        if adv and backward:                                            # + gan_loss_factor * mean(-D(decoded)), D's weights frozen
            out, saved = self.deng.forward(dec, save=True)
            npred = out.n * out.h * out.w
            self.last_g_loss = ops.sum_scaled(out.torch(), -1.0 / npred)
            ddec = self.deng.backward(saved, self._d_seed(out, -float(o.get("gan_loss_factor", 1.0)) / npred), True, None)
            ops.axpy(ddec, dec.grad())
        if backward:
            for fn in reversed(eng.tape):
                fn()
        eng.tape = None
        return rec, q_loss, self.last_idx

    def discriminator_backward(self, imgs: torch.Tensor, decoded: Act, current_step: int) -> dict:
        """Fills ``d_flat_g`` with the gradient of  0.5 * (mean(-D(imgs)) + mean(D(decoded)))  [+ the scaled R1 penalty on the steps
        net_d_reg_every divides] with respect to D's parameters (R:train_stage1.py:330-345)."""
        o, deng = self.opt, self.deng
        self.d_flat_g.zero_()
        x = imgs.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        xa = Act(x, B, H, W, 1, 1, 0)
        out_r, saved_r = deng.forward(xa, save=True)
        npred = out_r.n * out_r.h * out_r.w
        res = {"d_loss_real": ops.sum_scaled(out_r.torch(), -1.0 / npred)}
        deng.backward(saved_r, self._d_seed(out_r, -0.5 / npred), False, self.d_gw)
        out_f, saved_f = deng.forward(decoded, save=True)
        res["d_loss_fake"] = ops.sum_scaled(out_f.torch(), 1.0 / npred)
        deng.backward(saved_f, self._d_seed(out_f, 0.5 / npred), False, self.d_gw)
        del saved_f
        reg_every = int(o.get("net_d_reg_every", 16))
        if current_step % reg_every == 0:
            scale = float(o.get("r1_reg_weight", 1.0)) / 2.0 * reg_every
            res["r1_penalty"] = deng.r1_penalty(xa, scale, self.d_gw, fwd=(out_r, saved_r))
            res["r1_loss"] = res["r1_penalty"] * scale
        return res

    def step(self, imgs: torch.Tensor, forced_idx: Optional[torch.Tensor] = None, current_step: Optional[int] = None):
        o = self.opt
        step_no = self.current_step + 1 if current_step is None else int(current_step)
        adv = self._adversarial(step_no)
        update_g = (not adv) or step_no % int(o.get("generator_update_rate", 1)) == 0
        rec, q_loss, _ = self.forward_backward(imgs, forced_idx, step_no, backward=update_g)
        res = {"rec_loss": rec, "q_loss": q_loss}
        b1, b2, eps, wd = self.adam_hparams()
        decoded = self.last_decoded                                     # D sees the images decoded BEFORE this step's generator update
        if update_g:
            average_gradients(self.flat_g, self.world)
            self.step_count += 1
            ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, b1, b2, eps, wd, self.step_count)
            self.lr = self.sched.step()
            self.eng.refresh_weights()
            self.model.mark_weights_written(self._param_keys)
        if adv:
            res["g_loss"] = self.last_g_loss
            res.update(self.discriminator_backward(imgs, decoded, step_no))
            average_gradients(self.d_flat_g, self.world)
            self.d_steps += 1
            ops.adam_step(self.d_flat_p, self.d_flat_g, self.d_flat_m, self.d_flat_v, self.lr_d, b1, b2, eps, float(o.get("weight_decay_D") or 0.0), self.d_steps)
            self.lr_d = self.sched_d.step()
            self.deng.repack()
            res["lr_d"] = self.lr_d
        self.current_step = step_no
        res["lr"] = self.lr
        return res

    def state_dict(self) -> dict:
        st = super().state_dict()
        st["current_step"] = self.current_step
        if self.disc is not None:
            st["disc"] = {"steps": self.d_steps, "lr": self.lr_d, "scheduler": dict(vars(self.sched_d)),
                          "exp_avg": self.d_flat_m.detach().clone(), "exp_avg_sq": self.d_flat_v.detach().clone()}
        return st

    def load_state_dict(self, st: dict):
        super().load_state_dict(st)
        self.current_step = int(st.get("current_step", self.step_count))
        if self.disc is not None and "disc" in st:
            d = st["disc"]
            self.d_steps, self.lr_d = int(d["steps"]), float(d["lr"])
            for k, v in d["scheduler"].items():
                setattr(self.sched_d, k, v)
            self.d_flat_m.copy_(d["exp_avg"].to(self.dev))
            self.d_flat_v.copy_(d["exp_avg_sq"].to(self.dev))
            self.deng.repack()
