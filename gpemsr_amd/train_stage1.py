"""Stage-1 (VQGAN) training, GENERATOR PHASE: ``train_vqgan_onestep`` of R:train_stage1.py:291-357 for ``current_step <= gan_start``
(the first 40,000 steps with option/train_stage1.yml; lines 313-326):

    decoded, _, q_loss = generator(imgs)                              # model/vqgan.py:24-28: Encoder -> Codebook -> Decoder
    vq_loss = rec_loss_factor * L1(imgs, decoded) + codebook_loss_factor * q_loss
    vq_loss.backward();  optimizer_G.step();  scheduler_G.step()

on the HIP kernels: the recorded GroupNorm / residual block / non-local block / down-block of the stage-2 engine
(gpemsr_amd/train_stage2.py) plus the transposed-convolution backward of the stage-3 engine cover Encoder and Decoder;
``gpemsr_vq_codebook_loss`` is the codebook's loss, its two gradients and the straight-through estimator's value
(model/codebook.py:20-31); Adam and the cosine schedule are the trainers' common ones.

The generator lives inside the stage-3 model as ``refmodel.{encoder,codebook,decoder}`` (same blocks, same keys behind the prefix:
``Generator.load_state_dict`` takes them with strict=True, oracle/gen_golden_stage1.py does exactly that), so the trainer works on a
``gpemsr_amd.GPEMSR`` whose prior it trains; ``generator_state_dict()`` returns the reference's ``generator.state_dict()`` layout.

NOT built: the adversarial phase (step > gan_start) -- PatchGAN discriminator (4x4 stride-2 convolutions without padding,
InstanceNorm2d; model/discriminator.py:9-32), hinge-style losses and the R1 penalty (a gradient of a gradient).  DESIGN.md section 7.
"""
from __future__ import annotations

from typing import Optional

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
    """Generator-phase step of stage 1.  ``opt_train``: the ``train:`` block of option/train_stage1.yml (lr_G, beta1, beta2, lr_scheme,
    T_period, restarts, restart_weights, eta_min, rec_loss_factor, codebook_loss_factor, gan_start); ``beta``: the codebook's
    commitment weight (network.Generator.Codebook.beta)."""

    def __init__(self, model, opt_train: dict, device, beta: float = 1.0, world: int = 1):
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
        self.step_count = 0
        o = self.opt
        self.lr = float(o.get("lr_G", 4e-4))
        if o.get("lr_scheme", "CosineAnnealingLR_Restart") == "MultiStepLR":
            self.sched = MultiStepLRRestart(self.lr, o["lr_steps"], o.get("restarts"), o.get("restart_weights"), o.get("lr_gamma", 0.1))
        else:
            self.sched = CosineAnnealingLRRestart(self.lr, o.get("T_period", [1 << 30]), o.get("restarts"), o.get("restart_weights"),
                                                  o.get("eta_min", 0.0))

    def generator_state_dict(self) -> dict:
        """``generator.state_dict()`` of the reference (model/vqgan.py:16-22): encoder.* / codebook.* / decoder.* keys."""
        return {k[len("refmodel."):]: v.detach().clone() for k, v in self.model.state_dict().items() if k.startswith(_GEN_PREFIXES)}

    def forward_backward(self, imgs: torch.Tensor, forced_idx: Optional[torch.Tensor] = None):
        """-> (rec_loss, q_loss device scalars, code indices int32 [B*h*w]); fills ``flat_g`` with d(vq_loss)/d(parameters).
        ``forced_idx`` teacher-forces the arg-min (parity tests)."""
        if not imgs.is_cuda:
            raise RuntimeError("gpemsr_amd.train_stage1: inputs must live on a cuda/HIP device (there is no CPU path)")
        step_no = self.step_count + 1
        if step_no > int(self.opt.get("gan_start", 1 << 60)):
            raise NotImplementedError("gpemsr_amd.train_stage1: the adversarial phase (current_step > gan_start) is not built "
                                      "(PatchGAN discriminator + R1 penalty; DESIGN.md section 7)")
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
        for fn in reversed(eng.tape):
            fn()
        eng.tape = None
        return rec, q_loss, self.last_idx

    def step(self, imgs: torch.Tensor, forced_idx: Optional[torch.Tensor] = None):
        rec, q_loss, _ = self.forward_backward(imgs, forced_idx)
        average_gradients(self.flat_g, self.world)
        self.step_count += 1
        o = self.opt
        b1, b2, eps, wd = self.adam_hparams()
        ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, b1, b2, eps, wd, self.step_count)
        self.lr = self.sched.step()
        self.eng.refresh_weights()
        self.model.mark_weights_written(self._param_keys)
        return {"rec_loss": rec, "q_loss": q_loss, "lr": self.lr}
