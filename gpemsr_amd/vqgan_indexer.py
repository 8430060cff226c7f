"""Drop-in for the reference's ``model/vqgan_indexer.py::lrGenerator8 / lrGenerator16`` (the stage-2 model object of
train_stage2.py:120-141): same constructor argument (the ``lrGenerator8`` / ``lrGenerator16`` block of the option file),
same state-dict keys (``indexer.* decoder.* codebook.* encoder.*``), same methods --

    forward(lr, gt) -> (logits [B*h*w, 1024], gtcodebook_indices [B*h*w])       model/vqgan_indexer.py:77-84 / :35-41
    output_ref(imgs) -> decoded images                                            :69-74 / :27-32
    ref_extract(imgs) -> the five prior features                                  :87-91 / :44-48

-- on the HIP kernels.  In ``train()`` mode with autograd on, the logits are differentiable through
``torch.autograd.Function`` (the recorded tape of gpemsr_amd/train_stage2.py), so the reference's
``train_vqgan_onestep`` (train_stage2.py:351-366: torch CrossEntropyLoss, ``loss.backward()``, torch Adam) runs unchanged.
Only the indexer receives gradients -- as in the reference, where the encoder merely produces integer targets."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .arch import param_specs
from .ops import Act
from .synth import synth_tensor

_PFX = "refmodel."


class _State:
    def __init__(self, gen: "_LRGenerator", device):
        from .train_stage2 import Stage2Engine
        self.named = [(k, p) for k, p in gen.named_parameters() if p.requires_grad and k.startswith("indexer.")]
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in self.named]
        self.flat_g = torch.zeros(max(sum(sizes), 4), dtype=torch.float32, device=device)
        gw: Dict[str, torch.Tensor] = {}
        gb: Dict[str, torch.Tensor] = {}
        self.views: List[torch.Tensor] = []
        names, off = set(), 0
        for (k, p), sz in zip(self.named, sizes):
            g = self.flat_g[off:off + p.numel()].view(p.shape)
            self.views.append(g)
            base, leaf = (_PFX + k).rsplit(".", 1)
            names.add(base)
            (gw if leaf == "weight" else gb)[base] = g
            off += sz
        sd = {_PFX + k: v.detach() for k, v in gen.state_dict().items()}
        self.eng = Stage2Engine(sd, device, gen.scale, 5, 8, 64, gen._dec_nrb, names, gw, gb)
        self.versions = [p._version for _, p in self.named]

    def sync_weights(self):
        v = [p._version for _, p in self.named]
        if v != self.versions:
            self.eng.refresh_weights()
            self.versions = v

    def params(self):
        return [p for _, p in self.named]


class IndexerForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state: _State, lr: torch.Tensor, gt: torch.Tensor, *params):
        eng = state.eng
        state.sync_weights()
        lr = lr.detach().to(torch.float32).contiguous()
        gt = gt.detach().to(torch.float32).contiguous()
        target = eng.encoder_indices(Act(gt, gt.shape[0], gt.shape[2], gt.shape[3], 1, 1, 0))
        eng.tape = []
        try:
            logits = eng.indexer_logits_train(Act(lr, lr.shape[0], lr.shape[2], lr.shape[3], 1, 1, 0))
            ctx.tape = eng.tape
        finally:
            eng.tape = None
        ctx.state, ctx.logits = state, logits
        idx = target.to(torch.int64)
        ctx.mark_non_differentiable(idx)
        return logits.buf.view(logits.pixels, logits.c), idx

    @staticmethod
    def backward(ctx, g_logits, g_idx):
        state = ctx.state
        state.flat_g.zero_()
        ctx.logits.grad().buf.copy_(g_logits.contiguous().view(-1))
        for fn in reversed(ctx.tape):
            fn()
        ctx.tape = None
        return (None, None, None) + tuple(v.clone() for v in state.views)


class _LRGenerator(nn.Module):
    scale = 8

    def __init__(self, args, init_seed: int = 0):
        super().__init__()
        key = "Indexer8" if self.scale == 8 else "Indexer16"
        assert key in args and all(k in args for k in ("Decoder", "Codebook", "Encoder")), f"expected the {key}/Decoder/Codebook/Encoder blocks"
        self._dec_nrb = int(args["Decoder"]["num_resblock_per_scale"])
        specs = param_specs(argref=args, scale=self.scale, mode="8to1" if self.scale == 8 else "16to1")
        for name, spec in specs.items():
            if not name.startswith(_PFX):
                continue
            parts = name[len(_PFX):].split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, nn.Module())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(synth_tensor(name, spec, init_seed), requires_grad=True))
        self._state = None
        self._infer = None

    def _apply(self, fn, *a, **k):
        self._state = None
        self._infer = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._state = None
        self._infer = None
        return r

    def _engine(self, device, train: bool):
        if train:
            if self._state is None or self._state.eng.dev != device:
                self._state = _State(self, device)
            return self._state
        live = {_PFX + k: v for k, v in self.state_dict(keep_vars=True).items()}
        if self._infer is None or self._infer.dev != device:
            from .train_stage2 import Stage2Engine
            sd = {k: v.detach() for k, v in live.items()}
            self._infer = Stage2Engine(sd, device, self.scale, 5, 8, 64, self._dec_nrb, (), {}, {})
        else:
            self._infer.sync_weights(live)       # optimizer steps since the packs were made (validation between steps)
        return self._infer

    def forward(self, lr: torch.Tensor, gt: torch.Tensor):
        if not (lr.is_cuda and gt.is_cuda):
            raise RuntimeError("gpemsr_amd.vqgan_indexer: inputs must live on a cuda/HIP device (there is no CPU path)")
        if self.training and torch.is_grad_enabled():
            st = self._engine(lr.device, True)
            return IndexerForward.apply(st, lr, gt, *st.params())
        with torch.no_grad():
            eng = self._engine(lr.device, False)
            lr = lr.to(torch.float32).contiguous()
            gt = gt.to(torch.float32).contiguous()
            target = eng.encoder_indices(Act(gt, gt.shape[0], gt.shape[2], gt.shape[3], 1, 1, 0))
            logits = eng.indexer_logits(Act(lr, lr.shape[0], lr.shape[2], lr.shape[3], 1, 1, 0))
            return logits.buf.view(logits.pixels, logits.c), target.to(torch.int64)

    def ref_extract(self, imgs: torch.Tensor):
        """The five prior features (NCHW), model/vqgan_indexer.py:87-91."""
        with torch.no_grad():
            eng = self._engine(imgs.device, False)
            x = imgs.to(torch.float32).contiguous()
            feats = eng.ref_extract(Act(x, x.shape[0], x.shape[2], x.shape[3], 1, 1, 0), None, None)
            return [f.nchw() for f in feats]

    def output_ref(self, imgs: torch.Tensor):
        """Decoder(codebook(indexer(imgs))): the decoded prior image, model/vqgan_indexer.py:69-74 (== ref_extract's last item)."""
        return self.ref_extract(imgs)[-1]


class lrGenerator8(_LRGenerator):
    scale = 8


class lrGenerator16(_LRGenerator):
    scale = 16
