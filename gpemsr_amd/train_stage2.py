"""Stage-2 training step on the HIP kernels (first half): the reference's ``train_vqgan_onestep``
(/root/reference/GPEMSR-CREMI/GPEMSR/train_stage2.py:351-366)

    logits, gt_indices = lrgenerator(img_LR, img_GT)          # model/vqgan_indexer.py:77-84 (x8) / :35-41 (x16)
    loss = CrossEntropyLoss()(logits, gt_indices); loss.backward(); optimizer_G.step(); scheduler_G.step()

Only the indexer trains (encoder / codebook / decoder are frozen, train_stage2.py:152-170): the target indices come from
the frozen ``Encoder(GT)`` + nearest codebook vector (model/codebook.py:15-25), the logits from ``Indexer8/16(LR)``
(model/indexer.py:98-102).  It reuses the stage-3 machinery (gpemsr_amd/train.py: tape, recorded convolutions, wgrad, Adam,
schedulers, flat buffers, gradient all-reduce) and adds what the VQGAN-style blocks need: GroupNorm(+ReLU) backward, the
attention block's backward (softmax backward + four GEMMs on the conv kernel with per-image operands) and the
cross-entropy loss.  The GPEMSR host module carries the generator's weights under ``refmodel.*`` (model/GPEMSR.py:271-284),
so the trainer works on a ``gpemsr_amd.GPEMSR`` instance and a stage-2 checkpoint is ``{k[len('refmodel.'):]: v}`` of its
state dict.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import ops
from .dist import average_gradients
from .engine import _seq_len
from .ops import ACT_NONE, ACT_RELU, Act
from .train import CosineAnnealingLRRestart, MultiStepLRRestart, TrainEngine, _TrainerState, flatten_parameters


class Stage2Engine(TrainEngine):
    pack_encoder = True                       # the frozen Encoder is evaluated here (it is dead weight in stage 3)

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for name in self.trainable:           # attention q: forward weights are packed with C^-1/2 folded in (engine._pack_one)
            if name.endswith(".q") and ".feat_extract." in name:
                self.wscale[name] = float(int(self.sd[name + ".weight"].shape[0]) ** (-0.5))

    # -- recorded GroupNorm(32, 1e-6)(+ReLU)(+skip), not in place (the backward needs the layer input) -------------------
    def gn_train(self, x: Act, base: str, relu: bool, residual: Optional[Act] = None) -> Act:
        lib = ops._abi.load()
        gamma, beta = self.par[base + ".weight"], self.par[base + ".bias"]
        groups, eps = 32, 1e-6
        hw = x.h * x.w
        parts = max(1, min(64, hw // 64))
        ws = torch.empty(x.n * parts * x.c * 2, dtype=torch.float32, device=self.dev)
        mr = torch.empty(x.n * groups * 2, dtype=torch.float32, device=self.dev)
        ops._abi.check(lib.gpemsr_groupnorm_stats(x.ptr, x.n, hw, x.c, x.ld, groups, eps, ws.data_ptr(), parts, mr.data_ptr(), ops._stream()),
                       "groupnorm_stats")
        out = ops.new_act(x.n, x.h, x.w, x.c, device=self.dev)
        ops._abi.check(lib.gpemsr_groupnorm_apply(x.ptr, x.n, hw, x.c, x.ld, groups, mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), int(relu),
                                                  residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                                                  out.ptr, out.ld, ops._stream()), "groupnorm_apply")
        if self.tape is not None and (x.requires_grad or base in self.trainable):
            out.mark_grad()
            train_affine = base in self.trainable

            def _bwd():
                g = out.grad()
                if residual is not None and residual.requires_grad:
                    ops.axpy(g, residual.grad())
                need = x.n * parts * x.c * 2 + x.n * x.c * 2 + x.n * groups * 2
                w2 = ops._workspace(need, self.dev)
                dx = x.grad()
                ops._abi.check(lib.gpemsr_groupnorm_bwd(x.ptr, x.ld, g.ptr, g.ld, x.n, hw, x.c, groups, mr.data_ptr(), gamma.data_ptr(),
                                                        beta.data_ptr(), int(relu), w2.data_ptr(), w2.numel(), dx.ptr, dx.ld,
                                                        self.gw[base].data_ptr() if train_affine else None,
                                                        self.gb[base].data_ptr() if train_affine else None, ops._stream()), "groupnorm_bwd")
            self.tape.append(_bwd)
        return out

    def vq_resblock_train(self, x: Act, p: str) -> Act:
        """model/blocks.py:8-29 with every piece recorded."""
        t = self.gn_train(self.conv(x, p + ".block.0"), p + ".block.1", True)
        u = self.conv(t, p + ".block.3")
        skip = self.conv(x, p + ".channel_up") if (p + ".channel_up") in self.pc else x
        return self.gn_train(u, p + ".block.4", True, residual=skip)

    def nonlocal_train(self, x: Act, p: str) -> Act:
        """model/blocks.py:61-83.  Tokens are NHWC rows, so with S = q k^T, P = softmax(S), A = P v:
             dP = dA v^T,  dv = P^T dA,  dS = softmax'(P, dP),  dq = dS k,  dk = dS^T q
        are 1x1 'convolutions' with per-image operands (weight_image_stride); the transposed operands come from
        gpemsr_transpose_images."""
        n, h, w, c = x.n, x.h, x.w, x.c
        T = h * w
        if T % 32 != 0 or c % 32 != 0:
            raise RuntimeError(f"gpemsr_amd: non-local block needs latent tokens ({T}) and channels ({c}) to be multiples of 32")
        hn = self.gn_train(x, p + ".gn", False)
        q = self.conv(hn, p + ".q")                       # scaled by C^-1/2 (folded into the packed weights)
        k = self.conv(hn, p + ".k")
        v = self.conv(hn, p + ".v")
        gh, gw = T // 32, 32

        def gemm(src: Act, wt: Act, cout: int, cin: int, tag: str) -> Act:
            """out[n][rows][cout] = src[n][rows][cin] . wt[n][cout][cin]^T"""
            return ops.conv2d([src], ops.PackedConv(wt.buf, None, 1, cout, (cin,), 32), ACT_NONE, weight_image_stride=cout * cin, tag=tag)

        qa = q.reshape_hw(gh, gw)
        S = gemm(qa, k, T, c, p + ".qk")                                   # [n, T, T]
        ops.softmax_rows_(S.buf, n * T, T)                                 # P, in place
        vT = ops.transpose_images(v)                                       # [n][c][T]
        A = gemm(S, vT, c, T, p + ".pv")                                   # [n, T, c]
        A = A.reshape_hw(h, w)
        A.mark_grad()
        out = self.conv(A, p + ".proj_out", ACT_NONE, residual=x)

        def _bwd():
            dA = A.grad().reshape_hw(gh, gw)
            dP = gemm(dA, v, T, c, p + ".dP")                              # dP[i][j] = sum_c dA[i][c] v[j][c]
            PT = ops.transpose_images(Act(S.buf, n, T, 1, T, T, 0))        # [n][T(j)][T(i)]
            dAT = ops.transpose_images(A.grad())                           # [n][c][T]
            dv = gemm(Act(PT.buf, n, gh, gw, T, T, 0), dAT, c, T, p + ".dv")   # dv[j][c] = sum_i P[i][j] dA[i][c]
            ops.axpy(dv.reshape_hw(h, w), v.grad())
            ops._abi.check(ops._abi.load().gpemsr_softmax_bwd_rows(S.ptr, dP.ptr, n * T, T, ops._stream()), "softmax_bwd_rows")   # dP <- dS
            kT = ops.transpose_images(k)
            dq = gemm(dP, kT, c, T, p + ".dq")                             # dq[i][c] = sum_j dS[i][j] k[j][c]
            ops.axpy(dq.reshape_hw(h, w), q.grad())
            dST = ops.transpose_images(Act(dP.buf, n, T, 1, T, T, 0))
            qT = ops.transpose_images(q)
            dk = gemm(Act(dST.buf, n, gh, gw, T, T, 0), qT, c, T, p + ".dk")   # dk[j][c] = sum_i dS[i][j] q[i][c]
            ops.axpy(dk.reshape_hw(h, w), k.grad())
        # order on the tape: proj_out's record (appended by self.conv above) runs first in the backward, then this closure,
        # then the q/k/v convolutions and the GroupNorm recorded before it
        rec = self.tape.pop()
        self.tape.append(_bwd)
        self.tape.append(rec)
        return out

    def vq_layer_train(self, x: Act, p: str) -> Act:
        if (p + ".block.0") in self.pc:
            return self.vq_resblock_train(x, p)
        if (p + ".downblock") in self.pc:
            return self.conv(x, p + ".downblock", stride=2)
        if (p + ".q") in self.pc:
            return self.nonlocal_train(x, p)
        if p in self.pc:
            return self.conv(x, p)
        raise KeyError(p)

    def indexer_logits_train(self, xf: Act) -> Act:
        """Indexer8/16.forward (model/indexer.py:98-102 / :51-55) with the tape on -> logits [n, h, w, 1024]."""
        p = "refmodel.indexer"
        h = self.conv(xf, p + ".input_layer.0", ACT_RELU)
        for i in range(_seq_len(self.sd, p + ".feat_extract")):
            h = self.vq_layer_train(h, f"{p}.feat_extract.{i}")
        for i in range(_seq_len(self.sd, p + ".output_layer")):
            h = self.vq_layer_train(h, f"{p}.output_layer.{i}")
        return self.conv(h, p + ".embedding")

    # -- frozen target path ------------------------------------------------------------------------------------------------
    def encoder_indices(self, gt: Act) -> torch.Tensor:
        """Encoder(GT) (model/encoder.py:36-39) -> nearest codebook vector (model/codebook.py:15-25) -> int32 indices.
        argmin_k |z - e_k|^2 = argmax_k (z . e_k - |e_k|^2 / 2)."""
        p = "refmodel.encoder"
        tape, self.tape = self.tape, None
        try:
            h = self.conv(gt, p + ".input_layer.0", ACT_RELU)
            for i in range(_seq_len(self.sd, p + ".feat_extract")):
                h = self.vq_layer(h, f"{p}.feat_extract.{i}")
            for i in range(_seq_len(self.sd, p + ".output_layer")):
                h = self.vq_layer(h, f"{p}.output_layer.{i}")
            E = self.par["refmodel.codebook.embedding.weight"]
            if "codebook.nearest" not in self.pc:
                from .packing import pack_conv
                self.pc["codebook.nearest"] = pack_conv(E.view(E.shape[0], E.shape[1], 1, 1), -0.5 * (E * E).sum(dim=1), self.dev)
            score = ops.conv2d([h], self.pc["codebook.nearest"], ACT_NONE, tag="codebook.nearest")
            return ops.argmax_rows(score)
        finally:
            self.tape = tape


class Stage2Trainer(_TrainerState):
    """``train_vqgan_onestep`` (train_stage2.py:351-366).  ``opt_train``: the ``train:`` block of
    option/train_stage2_x{8,16}.yml (lr_G, beta1, beta2, lr_scheme, T_period, restarts, restart_weights, eta_min,
    weight_decay_G)."""

    def __init__(self, model, opt_train: dict, device, world: int = 1):
        from . import _abi
        _abi.load()
        assert all(p.is_cuda for p in model.parameters()), "move the model to the device first (model.to(device))"
        assert model.precision == "fp32"
        self.model, self.dev, self.world = model, device, world
        self.opt = dict(opt_train)
        named = [(k, p) for k, p in model.named_parameters() if k.startswith("refmodel.indexer.")]     # train_stage2.py:152-176
        self.flat_p, self.flat_g, self.flat_m, self.flat_v, gw, gb, names = flatten_parameters(named, device)
        self.n_params = sum(p.numel() for _, p in named)
        self._param_keys = [k for k, _ in named]
        model._engine = None
        model._train_state = None
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        self.eng = Stage2Engine(sd, device, model.scale, model.nframes, model.groups, model.nf, model._dec_nrb, names, gw, gb)
        self.gw, self.gb = gw, gb
        if os.environ.get("GPEMSR_FAST_REFRESH", "1") != "0":     # the per-step repack as ONE gather from the flat buffer (TrainEngine.enable_fast_refresh)
            self.eng.enable_fast_refresh(self.flat_p)
        self.step_count = 0
        o = self.opt
        self.lr = float(o.get("lr_G", 4e-4))
        if o.get("lr_scheme", "CosineAnnealingLR_Restart") == "MultiStepLR":
            self.sched = MultiStepLRRestart(self.lr, o["lr_steps"], o.get("restarts"), o.get("restart_weights"), o.get("lr_gamma", 0.1))
        else:
            self.sched = CosineAnnealingLRRestart(self.lr, o.get("T_period", [1 << 30]), o.get("restarts"), o.get("restart_weights"),
                                                  o.get("eta_min", 0.0))

    def forward_backward(self, img_LR: torch.Tensor, img_GT: torch.Tensor, forced_target: Optional[torch.Tensor] = None):
        """-> (loss [1] device tensor, target indices int32 [B*h*w]); fills ``flat_g``.  ``forced_target`` teacher-forces the
        encoder's code indices (parity tests: arg-min over 1024 codes is discontinuous)."""
        if not (img_LR.is_cuda and img_GT.is_cuda):
            raise RuntimeError("gpemsr_amd.train_stage2: inputs must live on a cuda/HIP device (there is no CPU path)")
        eng = self.eng
        self.flat_g.zero_()
        lr = img_LR.to(torch.float32).contiguous()
        gt = img_GT.to(torch.float32).contiguous()
        assert lr.dim() == 4 and lr.shape[1] == 1 and gt.dim() == 4 and gt.shape[1] == 1, "expected [B,1,h,w] LR and [B,1,H,W] GT"
        target = eng.encoder_indices(Act(gt, gt.shape[0], gt.shape[2], gt.shape[3], 1, 1, 0))
        self.last_target = target
        if forced_target is not None:
            target = forced_target.to(device=self.dev, dtype=torch.int32).contiguous()
        eng.tape = []
        logits = eng.indexer_logits_train(Act(lr, lr.shape[0], lr.shape[2], lr.shape[3], 1, 1, 0))
        rows = logits.pixels
        assert target.numel() == rows, f"encoder latent ({target.numel()} tokens) and indexer latent ({rows}) disagree"
        assert logits.ld == logits.c
        self.last_logits = logits
        row_loss = torch.empty(rows, dtype=torch.float32, device=self.dev)
        loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        g = logits.grad()
        ops._abi.check(ops._abi.load().gpemsr_cross_entropy(logits.ptr, target.data_ptr(), rows, logits.c, 1.0, row_loss.data_ptr(),
                                                            loss.data_ptr(), g.ptr, ops._stream()), "cross_entropy")
        for fn in reversed(eng.tape):
            fn()
        eng.tape = None
        return loss, target

    def step(self, img_LR: torch.Tensor, img_GT: torch.Tensor, forced_target: Optional[torch.Tensor] = None):
        loss, _ = self.forward_backward(img_LR, img_GT, forced_target)
        average_gradients(self.flat_g, self.world)
        self.step_count += 1
        o = self.opt
        b1, b2, eps, wd = self.adam_hparams()
        ops.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr, b1, b2, eps, wd, self.step_count)
        self.lr = self.sched.step()
        self.eng.refresh_weights()
        self.model.mark_weights_written(self._param_keys)     # the inference engine's packs are stale now
        return {"loss": loss, "lr": self.lr}
