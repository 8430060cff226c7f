"""Drop-in host module for the reference's ``model.GPEMSR.GPEMSR``.

Same constructor arguments, same ``forward(x[B,N,1,H,W]) -> (out[B,1,sH,sW],
ref_img[B,N,1,sH,sW])``, same state-dict key layout / ``load_state_dict(strict=True)``
behaviour and ``requires_grad`` flags as
/root/reference/GPEMSR-CREMI/GPEMSR/model/GPEMSR.py:225-456 (callers:
output_GPEMSR.py:36-52, train_stage3.py:123-141) -- but the forward runs on the
hand-written HIP kernels of libgpemsr_hip.so (gpemsr_amd/engine.py), on a
cuda/HIP device only.  There is no CPU fallback: calling forward without a GPU or
without the built library raises.
"""
from __future__ import annotations

import os
import weakref
from collections import namedtuple
from typing import Optional

import torch
import torch.nn as nn

from .arch import param_specs
from .synth import synth_tensor


def _register(root: nn.Module, dotted: str, tensor: torch.Tensor, trainable: bool, is_buffer: bool):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            # `model.vgg` is callable like the reference's VGG19 (train_stage3.py:353-355 hands it to ContextualLoss)
            mod.add_module(p, _VGGFeatures(root) if (mod is root and p == "vgg") else nn.Module())
        mod = mod._modules[p]
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=trainable))


class GPEMSR(nn.Module):
    """See module docstring.  Extra (optional) keyword arguments, all defaulted so the
    reference call sites work unchanged:
      init_seed    seed of the deterministic synthetic initialisation used until a
                   checkpoint is loaded (the reference's files are Google-Drive only);
      frame_chunk / tile_chunk   batching granularity of the two halves of the forward;
      precision    "fp32" (default: exact fp32 MFMA), "bf16x3" (3x3 convs on the bf16 matrix pipe with split hi+lo
                   operands, fp32-grade: ~1e-5 per op) or "bf16" (plain bf16 operands, fp32 accumulate);
      indexer_precision   bf16 path: "bf16" | "bf16x3:N" | "fp32:N" | "fp32:all" (how much of the indexer runs above bf16);
      winograd     fp32 path: "f4x4" (default) | "decoder_f4x4" | "f2x2" | "off" -- which 3x3 layers take a Winograd form
                   (option key `winograd`).
    """

    def __init__(self, ref_path_G, ref_path_Indexer, argref, nf=64, nframes=5, groups=8, front_RBs=5, back_RBs=10,
                 w_ref=True, ref_fusion_feat_RBs=3, align_mode='POD', fusion_mode='ThreeDA', mode='16to1', scale=16,
                 init_seed: int = 0, frame_chunk: int = 80, tile_chunk: int = 16, precision: str = "fp32",
                 indexer_precision: str = "bf16", winograd: str = None):
        super().__init__()
        if not (w_ref and align_mode == 'POD' and fusion_mode == 'ThreeDA'):
            raise NotImplementedError("gpemsr_amd implements the shipped configuration: w_ref=True, POD, ThreeDA")
        if (scale, mode) not in ((8, '8to1'), (16, '16to1')):
            raise ValueError('scale is wrong!')                      # model/GPEMSR.py:286-287
        self.nf, self.center, self.scale = nf, nframes // 2, scale
        self.nframes, self.groups = nframes, groups
        self.w_ref, self.align_mode, self.fusion_mode, self.mode = w_ref, align_mode, fusion_mode, mode
        self._dec_nrb = int(argref["Decoder"]["num_resblock_per_scale"])
        self._chunks = (frame_chunk, tile_chunk)
        self.precision = precision
        self.indexer_precision = indexer_precision       # bf16 path: "bf16" | "bf16x3:N" | "fp32:N" (gpemsr_amd/engine.py)
        self.winograd = winograd                         # fp32 path: None (= "f4x4") | "f4x4" | "decoder_f4x4" | "f2x2" | "off" (gpemsr_amd/engine.py)
        self._specs = param_specs(argref=argref, nf=nf, nframes=nframes, groups=groups, front_RBs=front_RBs,
                                  back_RBs=back_RBs, w_ref=w_ref, ref_fusion_feat_RBs=ref_fusion_feat_RBs,
                                  align_mode=align_mode, fusion_mode=fusion_mode, mode=mode, scale=scale)
        # (`model.vgg` is created as a callable _VGGFeatures module when its first parameter is registered, so the
        # state-dict key order stays the reference's)
        for name, spec in self._specs.items():
            _register(self, name, synth_tensor(name, spec, init_seed), spec.trainable, spec.is_buffer)
        self._engine = None
        self._train_state = None
        self._hip_written = set()     # parameter names a HIP kernel updated in place (trainers): see _get_engine
        # the reference loads the frozen prior at construction and raises if a file is missing (model/GPEMSR.py:275-276,
        # 283-284); None (build_model(load_prior_files=False)) keeps the deterministic synthetic prior
        for path, what in ((ref_path_G, "ref_path_G"), (ref_path_Indexer, "ref_path_Indexer")):
            if path is not None and not os.path.exists(str(path)):
                raise FileNotFoundError(f"gpemsr_amd.GPEMSR: {what}={path!r} does not exist (pass None to keep the "
                                        "synthetic prior, e.g. config.build_model(load_prior_files=False))")
        if ref_path_G is not None:
            self.refmodel.load_state_dict(torch.load(ref_path_G, map_location="cpu"), strict=False)
        if ref_path_Indexer is not None:
            self.refmodel.indexer.load_state_dict(torch.load(ref_path_Indexer, map_location="cpu"), strict=True)

    # -- weight lifecycle: any change of the parameters invalidates the packed copies
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = dict(state_dict)
        if strict:   # tolerate checkpoints written by a basicsr without the spynet mean/std buffers
            for k in ("align_module.spynet.mean", "align_module.spynet.std"):
                if k not in sd and k in self._specs:
                    sd[k] = self.state_dict()[k]
        r = super().load_state_dict(sd, strict=strict, **kw)
        self._engine = None
        self._train_state = None
        return r

    def _apply(self, fn, *a, **k):
        self._engine = None
        self._train_state = None
        return super()._apply(fn, *a, **k)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_train_state") else copy.deepcopy(v, memo)
        new._hip_written = set()
        new._rebind()
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None                  # packed device copies are rebuilt on first use
        st["_train_state"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._rebind()

    def _rebind(self):
        if "vgg" in self._modules:
            object.__setattr__(self._modules["vgg"], "_owner_ref", weakref.ref(self))

    def train(self, mode: bool = True):      # train() == eval(): no BatchNorm/Dropout in the network
        return super().train(mode)

    def _get_engine(self, device):
        live = self.state_dict(keep_vars=True)
        if self._engine is None or self._engine.dev != device:
            from . import _abi
            from .engine import Engine
            _abi.load()                        # fail loudly if the HIP library is missing
            sd = {k: v.detach() for k, v in live.items()}
            self._engine = Engine(sd, device, self.scale, self.nframes, self.groups, self.nf, self._dec_nrb,
                                  frame_chunk=self._chunks[0], tile_chunk=self._chunks[1], precision=self.precision,
                                  indexer_precision=self.indexer_precision, winograd=self.winograd)
        else:
            # validation between optimizer steps (R:train_stage3.py:197-312), torch optimizers on the autograd path,
            # model.refmodel.indexer.load_state_dict(...): repack whatever changed since the packs were made
            self._engine.sync_weights(live, force=self._hip_written)
        self._hip_written = set()
        return self._engine

    def mark_weights_written(self, names):
        """Called by the trainers after a HIP kernel (Adam on the flat buffer) wrote these parameters in place."""
        self._hip_written = set(getattr(self, "_hip_written", ())) | set(names)

    def invalidate_packed_weights(self, names=None):
        """Tell the module that parameters were written in a way torch's (data_ptr, _version) counters do not show -- ``p.data.copy_()``,
        ``p.data.mul_()`` (EMA updates, weight clipping), ``torch._foreach_*`` on ``.data`` views, raw-pointer writes: ``.data`` has its
        own version counter, so ``sync_weights`` cannot see those.  ``names`` = state-dict keys to repack on the next forward; None =
        all of them (drops the engine; the next forward repacks everything)."""
        if names is None:
            self._engine = None
            self._train_state = None
        else:
            self.mark_weights_written(names)

    def _get_train_state(self, device):
        """State of the torch.autograd path (gpemsr_amd/autograd.py): built on the first differentiable forward."""
        if getattr(self, "_train_state", None) is None or self._train_state.eng.dev != device:
            from . import _abi
            from .autograd import TrainState
            _abi.load()
            self._train_state = TrainState(self, device)
        return self._train_state

    def forward(self, x, forced_code_idx: Optional[torch.Tensor] = None, trace: Optional[dict] = None, want_u8: bool = False):
        """The reference's ``SR, ref_img = model(LQ)``.  Additive keyword: ``want_u8=True`` (inference) returns a third value, the 8-bit
        image [B, sH, sW] that util.tensor2img would make of SR, written by the network's last kernel."""
        if not x.is_cuda:
            raise RuntimeError("gpemsr_amd.GPEMSR.forward: input must live on a cuda/HIP device "
                               "(the MI355X kernel path is the only path)")
        if self.training and torch.is_grad_enabled() and forced_code_idx is None and trace is None:
            # train_stage3.py:343-349: model.train(); SR, ref_img = model(LR) with autograd on -> differentiable outputs whose
            # backward is the recorded HIP tape (parameter gradients reach p.grad / DistributedDataParallel as usual)
            from .autograd import SRForward
            st = self._get_train_state(x.device)
            return SRForward.apply(st, x, *st.params())
        with torch.no_grad():
            eng = self._get_engine(x.device)
            if want_u8 and x.shape[0] > 0:
                return eng.with_u8(lambda: eng.forward(x, forced_code_idx, trace), x.shape[0], x.shape[3], x.shape[4])
            return eng.forward(x, forced_code_idx, trace)

    def forward_volume(self, frames, windows, forced_code_idx: Optional[torch.Tensor] = None, want_u8: bool = False):
        """Volume mode: ``frames`` [T,1,H,W] are the distinct LR slices, ``windows`` [Wn,nframes] the slice numbers of
        every sliding window (edge windows repeat slices, output_GPEMSR.py:54-84,98-128).  The per-slice half of the
        network (VQGAN prior, VGG mask, prior fusion, pyramid) runs once per slice instead of once per window; results
        equal ``forward`` on the stacked windows bit for bit.  Returns (out [Wn,1,sH,sW], ref_img [T,1,sH,sW])."""
        if not frames.is_cuda:
            raise RuntimeError("gpemsr_amd.GPEMSR.forward_volume: input must live on a cuda/HIP device")
        with torch.no_grad():
            eng = self._get_engine(frames.device)
            if want_u8:
                return eng.with_u8(lambda: eng.forward_volume(frames, windows, forced_code_idx), len(windows), frames.shape[2], frames.shape[3])
            return eng.forward_volume(frames, windows, forced_code_idx)

    @property
    def vgg_features(self):
        return self.vgg


VggOutputs = namedtuple("VggOutputs", ['relu1_2', 'relu2_2', 'relu3_4', 'relu4_4', 'relu5_4'])

# torchvision vgg19 `features` (cfg E, no BN) as sliced by model/VGG.py:17-29: (slice, index, kind)
_VGG_LAYERS = (
    (1, 0, "conv"), (1, 2, "conv"),
    (2, 4, "pool"), (2, 5, "conv"), (2, 7, "conv"),
    (3, 9, "pool"), (3, 10, "conv"), (3, 12, "conv"), (3, 14, "conv"), (3, 16, "conv"),
    (4, 18, "pool"), (4, 19, "conv"), (4, 21, "conv"), (4, 23, "conv"), (4, 25, "conv"),
    (5, 27, "pool"), (5, 28, "conv"), (5, 30, "conv"), (5, 32, "conv"), (5, 34, "conv"),
)
_VGG_TAPS = {'relu1_2': 1, 'relu2_2': 2, 'relu3_4': 3, 'relu4_4': 4, 'relu5_4': 5}


class _VGGFeatures(nn.Module):
    """``model.vgg`` lookalike (model/VGG.py:34-52): calling it with a 3-channel NCHW batch returns the namedtuple of
    relu1_2 ... relu5_4, every conv (+ReLU) on the HIP conv kernels and the four 2x2 max-pools on gpemsr_maxpool2.
    The weights of slices 2-5 (frozen, only needed by the stage-3 contextual loss, train_stage3.py:352-359) are packed on
    first use."""

    def __init__(self, owner: "GPEMSR"):
        super().__init__()
        object.__setattr__(self, "_owner_ref", weakref.ref(owner))     # not a submodule: the owner contains us

    @property
    def owner(self) -> "GPEMSR":
        return self._owner_ref()

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_owner_ref", None)            # weak references do not pickle; GPEMSR.__setstate__ rebinds
        return st

    def _pc(self, eng, sl: int, idx: int):
        from .packing import pack_conv
        name = f"vgg.slice{sl}.{idx}@rgb" if (sl, idx) == (1, 0) else f"vgg.slice{sl}.{idx}"
        if name not in eng.pc:           # slice1.0 is packed for 1-channel input in the SR forward; here it sees RGB
            key = f"vgg.slice{sl}.{idx}"
            eng.pc[name] = pack_conv(eng.sd[key + ".weight"], eng.sd[key + ".bias"], eng.dev)
        return eng.pc[name]

    def features_nhwc(self, a, upto: str = 'relu5_4'):
        """NHWC Act in -> the requested tap as an NHWC Act (stops there)."""
        from . import ops
        eng = self.owner._get_engine(a.buf.device)
        last = _VGG_TAPS[upto]
        for sl, idx, kind in _VGG_LAYERS:
            if sl > last:
                break
            if kind == "pool":
                a = ops.maxpool2(a)
            else:
                a = ops.conv2d([a], self._pc(eng, sl, idx), ops.ACT_RELU, tag=f"vgg.slice{sl}.{idx}", precision=eng.precision)
        return a

    def forward(self, x: torch.Tensor) -> VggOutputs:
        from . import ops
        if not x.is_cuda:
            raise RuntimeError("gpemsr_amd: model.vgg needs a cuda/HIP tensor (there is no CPU path)")
        assert x.dim() == 4 and x.shape[1] == 3, "VGG model takes 3 channel images."
        if torch.is_grad_enabled() and x.requires_grad:
            # the training script hands model.vgg to ContextualLoss (train_stage3.py:352-355): differentiable w.r.t. x
            from .autograd import VGGForward
            return VggOutputs(*VGGForward.apply(self.owner._get_train_state(x.device), x))
        with torch.no_grad():
            eng = self.owner._get_engine(x.device)
            a = ops.from_nchw(x.to(torch.float32))
            taps, cur = [], 1
            for sl, idx, kind in _VGG_LAYERS:
                if sl != cur:
                    taps.append(a.nchw()); cur = sl
                if kind == "pool":
                    a = ops.maxpool2(a)
                else:
                    a = ops.conv2d([a], self._pc(eng, sl, idx), ops.ACT_RELU, tag=f"vgg.slice{sl}.{idx}", precision=eng.precision)
            taps.append(a.nchw())
            return VggOutputs(*taps)

    def relu1_2(self, x1: torch.Tensor) -> torch.Tensor:
        """relu1_2 of a 1-channel image expanded to 3 identical channels (the SR forward's use, model/GPEMSR.py:386,390)."""
        from . import ops
        eng = self.owner._get_engine(x1.device)
        a = ops.from_nchw(x1.to(torch.float32))
        f = eng.conv(eng.conv(a, "vgg.slice1.0", ops.ACT_RELU), "vgg.slice1.2", ops.ACT_RELU)
        return f.nchw()
