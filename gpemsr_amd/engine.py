"""Device-side forward of the stage-3 GPEMSR network on the HIP kernel library.

This is the host logic above the C ABI: it owns the packed weights and issues
the kernel sequence for ``GPEMSR.forward`` (reference:
/root/reference/GPEMSR-CREMI/GPEMSR/model/GPEMSR.py:323-456 and the modules it
calls).  The structure mirrors the reference stage by stage; what changes is how
each stage is executed:

  * activations are NHWC float32 and channel concatenations are never
    materialised (multi-source convolutions);
  * the 5 slices of every tile are batched through the per-frame front half
    (features, VQGAN prior, VGG mask, MPF) in frame chunks, and the 5 POD
    alignments of every tile are batched as 5*B "pairs" instead of a Python loop;
  * SpyNet is evaluated once per pair (the reference evaluates it twice with
    identical arguments, :99-100) and only VGG slice1 is evaluated (slices 2-5
    are dead in the forward, model/VGG.py:37-44);
  * PixelShuffle, residual adds, LeakyReLU/ReLU/sigmoid, the mask multiply and
    the bilinear base add are epilogues of the producing convolution.

torch is used for allocation and stream plumbing only; there is no CPU path.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional

import torch

from . import ops
from .ops import ACT_LRELU, ACT_LRELU_SIGMOID, ACT_NONE, ACT_RELU, Act
from .packing import (pack_conv, pack_conv_bf16, pack_dcn_rows_bf16, pack_conv_split, pack_convT, pack_convT_bf16, pack_convT_split, pack_dcn, pack_linear, pack_linear_bf16x3, pack_rowpair7,
                      pack_vgg_first, pack_cout1_taps, pack_winograd, pack_winograd4, pack_winograd7, pack_winograd77, pack_conv7_c32_cout16, pack_conv7_c8_cout32, pack_upconv_out, pack_rowsum7, pack_cout1_taps_f32, pack_upconv_out_f32, pack_rowsum7_f32)

_SPY_MEAN = (0.485, 0.456, 0.406)
_SPY_STD = (0.229, 0.224, 0.225)


def _seq_len(sd, prefix: str) -> int:
    idx, plen = set(), len(prefix) + 1
    for k in sd:
        if k.startswith(prefix + "."):
            head = k[plen:].split(".", 1)[0]
            if head.isdigit():
                idx.add(int(head))
    return (max(idx) + 1) if idx else 0


class Engine:
    BF16_SOFTMAX_MAX_COLS = 16384      # gpemsr_softmax_rows_bf16: 256 threads x 8 x 8 values (= the fp32 row softmax's limit)

    def __init__(self, sd: Dict[str, torch.Tensor], device, scale: int, nframes: int = 5, groups: int = 8,
                 nf: int = 64, dec_num_res_blocks: int = 1, frame_chunk: int = 80, tile_chunk: int = 16,
                 precision: str = "fp32", indexer_precision: str = "bf16", winograd: str = None):
        assert scale in (8, 16)
        assert nf == 64, "kernels are specialised for nf=64 (every shipped option file)"
        self.sd = sd
        self.dev = device
        self.scale, self.N, self.groups, self.nf = scale, nframes, groups, nf
        self.center = nframes // 2
        self.dec_nrb = dec_num_res_blocks
        self.frame_chunk, self.tile_chunk = frame_chunk, tile_chunk
        assert precision in ("fp32", "bf16x3", "bf16", "bf16op"), precision
        self.vgg_cos_epi = precision == "fp32" and getattr(self, "fuse_tail_f32", True) and os.environ.get("GPEMSR_VGG_COS32", "1") != "0"
        self.gn_epi = precision == "bf16" or (precision == "fp32" and getattr(self, "fuse_tail_f32", True) and os.environ.get("GPEMSR_GN_EPI32", "1") != "0")
        # fp32: exact fp32 MFMA everywhere (default).
        # bf16: bf16 NHWC activations in HBM + bf16 MFMA (BASELINE configs[2]); 1-channel images, flows, deformable offsets
        #       and the indexer's logits + argmax stay at fp32 precision: the logits GEMM takes the fp32 output of
        #       the last convolution and runs as three bf16 products of hi + lo operands (GPEMSR_LOGITS_X3=0: the fp32 MFMA kernel).
        # bf16x3 / bf16op: fp32 activations; convolutions whose sources are multiples of 16 channels run on the bf16 matrix
        #       pipe with operands converted in LDS (split hi+lo = fp32-grade, or plain bf16 operands).
        self.precision = precision
        self.bf16 = precision == "bf16"
        # bf16 path only: "<mode>:<N>" runs the LAST N units of the indexer (R:model/indexer.py:89-96: output_layer's three ResidualBlocks
        # and its 1x1 convolution, counted from the end) on fp32 activations with `mode` = bf16x3 (split hi + lo products, fp32-grade) or
        # fp32 (exact kernel): the argmax over 1024 logits is discontinuous, so the code indices a free-running bf16 forward picks differ
        # from the reference's wherever the top-2 margin is below the logit error; the tail's precision trades time for agreement.
        # "bf16" (default): everything on the bf16 data path, logits GEMM as three products (always).
        self.indexer_precision = indexer_precision or "bf16"
        self._hp_mode, self._hp_n = None, 0
        # "fp32:all" / "bf16x3:all": the WHOLE indexer (R:model/indexer.py:63-102) on the exact-fp32 / split-bf16 kernels -- a second, indexer-only Engine over the same
        # weights picks the code indices, the rest of the path stays bf16 (the setting with ~100 % code agreement; cost: the fp32 indexer)
        self._idx_engine = None
        if self.bf16 and self.indexer_precision in ("fp32:all", "bf16x3:all"):
            # ("bf16x3:all": the same second engine with its convolutions as three split hi + lo bf16 products on fp32 activations -- fp32-grade
            #  logits from the bf16 matrix pipe; measured in profiles/r06_indexer_precision_sweep.log)
            sub = {k: v for k, v in sd.items() if k.startswith("refmodel.indexer.")}
            self._idx_engine = Engine(sub, device, scale, nframes, groups, nf, dec_num_res_blocks, frame_chunk, tile_chunk,
                                      precision=self.indexer_precision.split(":")[0], winograd=winograd)
        elif self.bf16 and self.indexer_precision != "bf16":
            mode, _, cnt = self.indexer_precision.partition(":")
            assert mode in ("bf16x3", "fp32") and (cnt == "" or cnt.isdigit()), f"indexer_precision {indexer_precision!r}: bf16 | bf16x3:N | fp32:N | bf16x3:all | fp32:all"
            self._hp_mode, self._hp_n = mode, int(cnt or 1)
        self.fold_gn = os.environ.get("GPEMSR_FOLD_GN", "1") != "0"     # bf16 path: first GroupNorm apply of a VQGAN block inside the consuming conv
        self.fold_gn32 = os.environ.get("GPEMSR_FOLD_GN32", "1") != "0"  # fp32 path: the same, in the input transform of the F(4x4) Winograd form
        self.flash_attn = os.environ.get("GPEMSR_FLASH_ATTN", "1") != "0"   # bf16 path: NonLocalBlock products + softmax as one kernel (C = 512, T % 128 == 0)
        # exact-fp32 path, option key `winograd` (`None` = the default "f4x4"):
        #   "f4x4"          3x3 stride-1 layers with >= 64 input channels (% 8) and cout % 64 == 0 (or >= 128 couts through zero-padded weights:
        #                   the DCN packs' 216-channel offset convolutions) in the F(4x4,3x3) form (36 instead of 144 multiplies per 4x4 outputs,
        #                   ~2e-5 of the result; csrc/conv_wino4.hip; 64-channel layers included: 7.2 -> 4.9 ms on a 1024^2 map), the other 3x3
        #                   stride-1 layers in F(2x2,3x3) (16 instead of 36, ~5e-6; csrc/conv_wino.hip), SpyNet's 8 -> 32 and 32 <-> 64 7x7 layers in
        #                   the 2-D form F(2x2,7x7) (64 of 196 multiplies, ~5e-6; csrc/conv7_wino2d.hip);
        #   "decoder_f4x4"  the same, but the indexer's layers stay on F(2x2): its arg-max decides codebook entries, this keeps the tighter
        #                   rounding there (+17 ms per 16-window step);
        #   "f2x2"          F(2x2) for every 3x3 layer, the 1-D row form F(2,7) (8 of 14, csrc/conv7_wino.hip) for SpyNet's 32 <-> 64 layers;
        #   "off"           the direct form everywhere (3e-6 per layer).
        # The training engines use the Winograd forms for their FROZEN layers only (`_wino_layer`: the transformed weights of a trainable layer
        # are not a permutation of its master weights, so the one-gather repack cannot refresh them).
        # Development switches (A/B scripts; the option key wins): GPEMSR_WINOGRAD=0, GPEMSR_WINOGRAD4=0|decoder|all, GPEMSR_WINOGRAD7=0.
        if winograd is None:
            w2 = os.environ.get("GPEMSR_WINOGRAD", "1") != "0"
            w4 = os.environ.get("GPEMSR_WINOGRAD4", "all")
            assert w4 in ("0", "decoder", "all"), f"GPEMSR_WINOGRAD4={w4!r}: 0 | decoder | all"
            winograd = "off" if not w2 else {"0": "f2x2", "decoder": "decoder_f4x4", "all": "f4x4"}[w4]
            w7 = os.environ.get("GPEMSR_WINOGRAD7", "1") != "0"
        else:
            w7 = True
        assert winograd in ("f4x4", "decoder_f4x4", "f2x2", "off"), f"winograd: {winograd!r} (f4x4 | decoder_f4x4 | f2x2 | off)"
        self.winograd_form = winograd
        self.winograd = precision == "fp32" and winograd != "off"
        self.winograd7 = self.winograd and w7
        self.winograd77 = self.winograd7 and winograd != "f2x2" and os.environ.get("GPEMSR_WINOGRAD77", "1") != "0"   # (2-D form of the same 7x7 layers; A/B switch)
        self.winograd4 = {"f4x4": "all", "decoder_f4x4": "decoder"}.get(winograd, "0") if self.winograd else "0"
        self.winograd4_min_cin = int(os.environ.get("GPEMSR_WINOGRAD4_MIN_CIN", "64"))
        self.fuse_argmax = os.environ.get("GPEMSR_FUSE_ARGMAX", "1") != "0"   # bf16 path: codebook arg-max inside the logits GEMM (no logits tensor)
        self.fuse_vgg = True            # bf16 path: gpemsr_vgg_mask_bf16 (tests switch it off to compare with the layer-by-layer form)
        self.fuse_dcn = os.environ.get("GPEMSR_FUSE_DCN", "1") != "0"     # bf16 path: DCN sampling + contraction in one kernel (gpemsr_dcn_conv_bf16)
        self.split = precision in ("bf16x3", "bf16op")
        self._forced_flow = None
        self.o = ops            # operator namespace; the training engine swaps in a recording proxy (gpemsr_amd/train.py)
        self.pc: Dict[str, ops.PackedConv] = {}
        self.par: Dict[str, torch.Tensor] = {}
        self._pack_all()
        self._sig = self._signature(sd)

    # ------------------------------------------------------------------ packing
    def _pack_all(self):
        sd, dev, nf = self.sd, self.dev, self.nf
        self._ffc = 64 if self.bf16 else 48      # channels of the flow/frame concat buffer (34 used; bf16: 32-channel chunks)
        splits = {
            "reffusionconv1": (nf, 64), "reffusionconv2": (nf, 128, nf), "down_fea_conv2": (nf, nf),
            "reffusionconv3": (nf, 256, 2 * nf), "down_fea_conv3": (nf, 2 * nf), "reffusionconv4": (nf, 512, 3 * nf),
            "reduce_dim_conv": (nf, 3 * nf, nf) if self.scale == 16 else (nf, 2 * nf, nf),
            # [nbr_fea, ref_fea, flow1|flow2|nbr_frame|ref_frame]: the 16+16 flow features and the 2 frames share ONE
            # 48-channel buffer (14 zero channels, zero weight columns), so every source is a multiple of 16 channels
            "align_module.L3_offset_conv1": (nf, nf, self._ffc), "align_module.L2_offset_conv1": (nf, nf, self._ffc),
            "align_module.L1_offset_conv1": (nf, nf, self._ffc), "align_module.L2_offset_conv2": (nf, nf),
            "align_module.L1_offset_conv2": (nf, nf), "align_module.L2_fea_conv": (nf, nf),
            "align_module.L1_fea_conv": (nf, nf), "align_module.cas_offset_conv1": (nf, nf),
        }
        self._splits = splits
        for k, w in sd.items():
            self._pack_one(k, w)
        m, s = sd.get("align_module.spynet.mean"), sd.get("align_module.spynet.std")
        self.spy_mean = tuple(float(v) for v in m.flatten()) if m is not None else _SPY_MEAN
        self.spy_std = tuple(float(v) for v in s.flatten()) if s is not None else _SPY_STD

    # ------------------------------------------------------------------ weight lifecycle
    @staticmethod
    def _signature(tensors: Dict[str, torch.Tensor]):
        """(storage address, torch version counter) per entry: moves whenever torch writes a tensor in place or rebinds it."""
        return {k: (v.data_ptr(), v._version) for k, v in tensors.items()}

    def sync_weights(self, live: Dict[str, torch.Tensor], force=()):
        """Bring the packed copies up to date with the live parameters: every entry whose storage or version counter moved
        since it was packed (an optimizer step, ``p.data = ...``, a sub-module ``load_state_dict``) and every name in
        ``force`` (weights written by a HIP kernel through the raw pointer, which torch's counters cannot see) is repacked.
        The reference re-reads its Parameters on every forward (R:train_stage3.py:197-312 validates between optimizer
        steps); a packed copy must therefore never outlive the weights it was made from."""
        if getattr(self, "_idx_engine", None) is not None:
            self._idx_engine.sync_weights({k: v for k, v in live.items() if k.startswith("refmodel.indexer.")}, [k for k in force if k.startswith("refmodel.indexer.")])
        sig = self._signature(live)
        changed = {k for k in sig if sig[k] != self._sig.get(k)} | {k for k in force if k in live}
        if not changed:
            return 0
        bases = set()
        for k in changed:
            self.sd[k] = live[k].detach()
            bases.add(k.rsplit(".", 1)[0])
        if any(b.startswith("refmodel.decoder.") for b in bases):
            self.par.pop("@upout.frag", None); self.par.pop("@upout.consts", None)
        if any(b.startswith("vgg.slice1.0") for b in bases):
            self.par.pop("vgg.w1", None); self.par.pop("vgg.b1", None)
        for b in bases:
            for nm in [n for n in self.pc if n == b or n.startswith(b + "@")]:
                del self.pc[nm]                               # incl. lazily packed variants (model.vgg's "@rgb" entry)
            if (b + ".weight") in self.sd:
                self._pack_one(b + ".weight", self.sd[b + ".weight"])
        if any(k.startswith("align_module.spynet.") and k.endswith((".mean", ".std")) for k in changed):
            m, sdv = self.sd.get("align_module.spynet.mean"), self.sd.get("align_module.spynet.std")
            self.spy_mean = tuple(float(v) for v in m.flatten()) if m is not None else _SPY_MEAN
            self.spy_std = tuple(float(v) for v in sdv.flatten()) if sdv is not None else _SPY_STD
        self._sig = sig
        return len(changed)

    def _pack_one(self, k: str, w: torch.Tensor):
        """Repack one state-dict entry into its kernel-native form (called for all at load, and again for the trainable
        convolutions after every optimizer step by gpemsr_amd/train.py)."""
        sd, dev, nf, splits = self.sd, self.dev, self.nf, self._splits
        ps = {"upconv1", "upconv2", "upconv3", "upconv4"}
        if not k.endswith(".weight"):
            return
        name = k[:-7]
        if (name.startswith("refmodel.encoder.") and not getattr(self, "pack_encoder", False)) or \
                name.startswith("vgg.slice") and not name.startswith("vgg.slice1."):
            return                                   # never evaluated in the stage-3 forward (stage 2 packs the encoder)
        b = sd.get(name + ".bias")
        if w.dim() == 4 and name.endswith("dcnpack"):
            self.pc[name] = pack_dcn(w, b, dev)
            if self.bf16:       # 1x1 over the tap-major column tensor [9*cin]
                self.pc[name].wb = pack_conv_bf16(w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1, 1, 1), dev)
                if tuple(w.shape) == (64, 64, 3, 3):
                    self.pc[name].wrows = pack_dcn_rows_bf16(w, dev)       # sampling + contraction in one kernel (csrc/dcn_bf16.hip)
        elif w.dim() == 4 and (name.startswith("reffea_L") or name.endswith(".upblock")):
            self.pc[name] = pack_convT(w, b, dev)
            if self.split and w.shape[0] % 16 == 0:
                self.pc[name].w16 = pack_convT_split(self.pc[name], dev)
            if self.bf16:
                self.pc[name].wb = pack_convT_bf16(w, dev)
        elif name == "vgg.slice1.0":
            self.pc[name] = pack_vgg_first(w, b, dev)
        elif w.dim() == 4 and name.endswith((".q",)) and ".feat_extract." in name:
            c = w.shape[0]
            sc = float(int(c) ** (-0.5))
            self.pc[name] = pack_conv(w, b, dev, scale=sc)                        # fold C^-1/2 (blocks.py:76)
            if self.split and c % 32 == 0:
                self.pc[name].w16 = pack_conv_split(self.pc[name], w.detach().to(torch.float32) * sc, dev)
            if self.bf16:
                self.pc[name].wb = pack_conv_bf16(w, dev, scale=sc)
        elif w.dim() == 4:
            if name.endswith("_offset_conv1") and w.shape[1] == 2 * nf + 34:
                w = torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, self._ffc - 34))   # 162 -> 176 (bf16: 192) input channels
            w7c8 = None
            if self.precision != "fp32" and w.shape[1] == 8 and w.shape[2] == 7 and ".spynet." in name:
                if self.bf16 and tuple(w.shape) == (32, 8, 7, 7) and os.environ.get("GPEMSR_CONV7_C8", "1") != "0":
                    w7c8 = pack_conv7_c8_cout32(w, dev)     # four taps per 16x16x32 MFMA (csrc/conv7_bf16.hip)
                # SpyNet stems (8 -> 32, 7x7): zero-pad cin to 16 so they run on the split-bf16 kernel too
                w = torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, 8))
            self.pc[name] = pack_conv(w, b, dev, splits.get(name), pixel_shuffle=name in ps)
            self.pc[name].w7c8 = w7c8
            kk = w.shape[2]
            if (self.split or self._is_hp_layer(name)) and (kk in (3, 7) and all(c % 16 == 0 for c in self.pc[name].splits)
                                                            or kk == 1 and all(c % 32 == 0 for c in self.pc[name].splits)):
                self.pc[name].w16 = pack_conv_split(self.pc[name], w, dev, pixel_shuffle=name in ps)
            if self.bf16 and all(c % 16 == 0 for c in self.pc[name].splits):
                self.pc[name].wb = pack_conv_bf16(w, dev, self.pc[name].splits, pixel_shuffle=name in ps)
            if (self.winograd and self.winograd4 != "0" and self._wino_layer(name) and kk == 3 and w.shape[0] % 32 != 0 and w.shape[0] >= 128 and name not in ps
                    and w.shape[1] >= self.winograd4_min_cin and all(c % 8 == 0 for c in self.pc[name].splits)):
                # cout % 32 != 0 (the DCN packs' 216-channel offset convolutions): no F(2x2) form, but the F(4x4) one through zero-padded weights
                self.pc[name].wino4 = pack_winograd4(w, dev)
            if (self.winograd and self._wino_layer(name) and kk == 3 and w.shape[0] % 32 == 0 and (name not in ps or w.shape[0] % 64 == 0)
                    and all(c % 8 == 0 for c in self.pc[name].splits)):
                self.pc[name].wino = pack_winograd(w, dev, pixel_shuffle=name in ps)      # Winograd form of the 3x3 stride-1 layers (fp32 path)
                if (self.winograd4 != "0" and w.shape[0] % (256 if name in ps else 64) == 0 and w.shape[1] >= self.winograd4_min_cin
                        and (self.winograd4 == "all" or not name.startswith("refmodel.indexer."))):
                    self.pc[name].wino4 = pack_winograd4(w, dev, pixel_shuffle=name in ps)     # F(4x4,3x3) form
            if self.winograd7 and self._wino_layer(name) and kk == 7 and w.shape[0] % 16 == 0 and w.shape[1] % 8 == 0 and len(self.pc[name].splits) == 1:
                if w.shape[0] % 32 == 0 and w.shape[1] >= int(os.environ.get("GPEMSR_WINOGRAD7_MIN_CIN", "32")):
                    self.pc[name].wino7 = pack_winograd7(w, dev)      # 1-D Winograd F(2, 7) form of SpyNet's 32 <-> 64 7x7 layers (fp32 path)
                if (self.winograd77 and w.shape[1] >= int(os.environ.get("GPEMSR_WINOGRAD77_MIN_CIN", "8"))
                        and (w.shape[0] % 32 == 0 or os.environ.get("GPEMSR_WINOGRAD77_C16", "1") != "0")):
                    # the 2-D form F(2x2, 7x7): 64 instead of 112 (1-D) / 196 (direct) multiplies per 2x2 outputs; also the one-chunk 8 -> 32 stems
                    # (447.2 -> 445.4 ms per step, profiles/r06_ab_winograd77.log)
                    self.pc[name].wino77 = pack_winograd77(w, dev)
            if self.bf16 and tuple(w.shape) == (1, 64, 3, 3):
                self.pc[name].wtap = pack_cout1_taps(w, dev)            # 64 -> 1 on the matrix cores (csrc/tap_sum.hip)
            if not self.bf16 and tuple(w.shape) == (1, 64, 3, 3) and getattr(self, "fuse_tail_f32", True):
                self.pc[name].wtap32 = pack_cout1_taps_f32(w, dev)      # the same on the fp32 matrix pipe
            if self.precision == "fp32" and tuple(w.shape) == (2, 16, 7, 7) and getattr(self, "fuse_tail_f32", True):
                self.pc[name].wrow7_32 = pack_rowsum7_f32(w, dev)       # SpyNet flow update as row sums, fp32 matrix pipe
            if self.precision == "fp32" and w.shape[0] == 16 and tuple(w.shape[2:]) == (7, 7) and getattr(self, "fuse_tail_f32", True) \
                    and os.environ.get("GPEMSR_ROWPAIR7", "1") != "0":
                self.pc[name].wpair7 = pack_rowpair7(w, dev)            # SpyNet 32 -> 16: row-pair form (16 couts fill half a matrix tile)
            if self.bf16 and tuple(w.shape) == (16, 32, 7, 7) and os.environ.get("GPEMSR_CONV7_C16", "1") != "0":
                self.pc[name].w7c16 = pack_conv7_c32_cout16(w, dev)     # SpyNet 32 -> 16 on the 16x16x32 MFMA shape (csrc/conv7_bf16.hip)
            if self.bf16 and tuple(w.shape) == (2, 16, 7, 7):
                self.pc[name].wrow7 = pack_rowsum7(w, dev)              # SpyNet flow update as row sums (csrc/tap_sum.hip)
        elif w.dim() == 2 and name.endswith("indexer.embedding"):
            self.pc[name] = pack_linear(w, b, dev)
            if self.bf16 and w.shape[1] % 16 == 0 and os.environ.get("GPEMSR_LOGITS_X3", "1") != "0":
                self.pc[name + "@x3"] = pack_linear_bf16x3(w, b, dev)
        elif w.dim() == 2 and name.endswith("codebook.embedding"):
            self.par[k] = w.detach().to(torch.float32).contiguous().to(dev)
        elif w.dim() == 5:                             # ThreeDA.conv3D_{1,2}: [t,t,1,1,1]
            self.par[k] = w.detach().to(torch.float32).reshape(w.shape[0], w.shape[1]).contiguous().to(dev)
            self.par[name + ".bias"] = b.detach().to(torch.float32).contiguous().to(dev)
        elif w.dim() == 1:                             # GroupNorm affine
            self.par[k] = w.detach().to(torch.float32).contiguous().to(dev)
            self.par[name + ".bias"] = b.detach().to(torch.float32).contiguous().to(dev)

    def _hp_first_unit(self) -> int:
        """Index of the first indexer output_layer unit that runs at the higher precision (n_out: none)."""
        n_out = _seq_len(self.sd, "refmodel.indexer.output_layer")
        return n_out if self._hp_mode is None else max(0, n_out - self._hp_n)

    def _is_hp_layer(self, name: str) -> bool:
        if self._hp_mode != "bf16x3" or not name.startswith("refmodel.indexer.output_layer."):
            return False
        return int(name.split(".")[3]) >= self._hp_first_unit()

    # ------------------------------------------------------------------ helpers
    def _wino_layer(self, name: str) -> bool:
        """Layers that may keep a Winograd weight form (every eligible one; the training engines exclude their trainable layers)."""
        return True

    @staticmethod
    def wino_geometry_ok(x: Act, cout: int) -> bool:
        """The Winograd kernels cut the image into 8 x 32 (cout % 64 == 0) or 16 x 32 pixel tiles: a map that fills less than 2/3 of its
        tiles (the 16 x 16 levels of the training crops) is faster on the direct kernel."""
        th = 8 if cout % 64 == 0 else 16
        return 3 * x.h * x.w >= 2 * (-(-x.h // th) * th) * (-(-x.w // 32) * 32)

    def _wino4_runs(self, x: Act, pc) -> bool:
        """Will `conv(x, <layer of pc>)` take the F(4x4) Winograd form?  (the decisions of `conv` below)"""
        return (self.winograd and pc.wino is not None and pc.wino4 is not None and self.wino_geometry_ok(x, pc.cout)
                and 3 * x.h * x.w >= 2 * (-(-x.h // 16) * 16) * (-(-x.w // 32) * 32))

    def conv(self, srcs, name, act=ACT_NONE, **kw) -> Act:
        kw.setdefault("precision", self.precision)
        pc = self.pc[name]
        if self.winograd and kw["precision"] == "fp32" and (pc.wino is not None or pc.wino4 is not None) \
                and self.wino_geometry_ok(srcs if isinstance(srcs, Act) else srcs[0], pc.cout):
            kw.setdefault("winograd", True)
            x0 = srcs if isinstance(srcs, Act) else srcs[0]
            if pc.wino4 is not None and 3 * x0.h * x0.w < 2 * (-(-x0.h // 16) * 16) * (-(-x0.w // 32) * 32):
                kw.setdefault("winograd4", False)            # a map that fills < 2/3 of the F(4x4) kernel's 16 x 32 tiles stays on F(2x2) (8 x 32 tiles)
        return self.o.conv2d(srcs, pc, act, tag=name, **kw)

    def resblocks_nobn(self, x: Act, prefix: str, pixmul: Optional[Act] = None) -> Act:
        """basicsr ResidualBlockNoBN chain; ``pixmul`` multiplies the output of the LAST block
        (the MPF mask, model/GPEMSR.py:403)."""
        n = _seq_len(self.sd, prefix)
        for i in range(n):
            t = self.conv(x, f"{prefix}.{i}.conv1", ACT_RELU)
            x = self.conv(t, f"{prefix}.{i}.conv2", ACT_NONE, residual=x, pixmul=pixmul if i == n - 1 else None)
        return x

    # ------------------------------------------------------------------ VQGAN prior
    def vq_resblock(self, x: Act, p: str, precision: Optional[str] = None) -> Act:
        if precision is not None and precision != self.precision:
            # a block at another precision than the engine's (the indexer's tail, `indexer_precision`): fp32 activations, plain sequence
            t = self.conv(x, p + ".block.0", precision=precision)
            self.o.groupnorm_relu(t, self.par[p + ".block.1.weight"], self.par[p + ".block.1.bias"], True, out=t)
            u = self.conv(t, p + ".block.3", precision=precision)
            skip = self.conv(x, p + ".channel_up", precision=precision) if (p + ".channel_up") in self.pc else x
            return self.o.groupnorm_relu(u, self.par[p + ".block.4.weight"], self.par[p + ".block.4.bias"], True, residual=skip, out=u)
        # the conv epilogue leaves the GroupNorm partial sums (no statistics pass over the tensor): bf16 path and exact-fp32 path
        epi = self.gn_epi
        if self.bf16 and epi and int(os.environ.get("GPEMSR_GN_EPI_MIN_C", "100")) > self.pc[p + ".block.0"].cout:
            # 64-channel blocks (K = 576: little matrix work per output to hide the epilogue's lane reductions behind): one statistics
            # pass over the stored tensor (0.55 ms per 2.7 GB) is cheaper than the sums in the epilogue (+0.9 ms); A/B -1.1 ms per step
            epi = False
        t = self.conv(x, p + ".block.0", gn_stats=epi)
        pc3 = self.pc[p + ".block.3"]
        if (not self.bf16 and self.precision == "fp32" and self.fold_gn32 and self._wino4_runs(t, pc3) and self.o.conv_affine_source_ok32(t, pc3)):
            # exact-fp32 path: the same fold in the input transform of the F(4x4) Winograd form (csrc/conv_wino4.hip, W4_AFF)
            sc, sh = self.o.groupnorm_scale_shift(t, self.par[p + ".block.1.weight"], self.par[p + ".block.1.bias"])
            u = self.conv(t, p + ".block.3", gn_stats=epi, a_affine=(sc, sh, True))
        elif self.bf16 and self.fold_gn and self.o.conv_affine_source_ok(t, self.pc[p + ".block.3"]):
            # the first Normalize + ReLU of the block is applied by the second convolution while it stages its source: the normalised
            # tensor never exists in HBM (one read + one write of the block's intermediate less)
            sc, sh = self.o.groupnorm_scale_shift(t, self.par[p + ".block.1.weight"], self.par[p + ".block.1.bias"])
            u = self.conv(t, p + ".block.3", gn_stats=epi, a_affine=(sc, sh, True))
        else:
            self.o.groupnorm_relu(t, self.par[p + ".block.1.weight"], self.par[p + ".block.1.bias"], True, out=t)
            u = self.conv(t, p + ".block.3", gn_stats=epi)
        skip = self.conv(x, p + ".channel_up") if (p + ".channel_up") in self.pc else x
        return self.o.groupnorm_relu(u, self.par[p + ".block.4.weight"], self.par[p + ".block.4.bias"], True, residual=skip, out=u)

    def nonlocal_block(self, x: Act, p: str) -> Act:
        """model/blocks.py:61-83 with the score matrix materialised per frame chunk."""
        n, h, w, c = x.n, x.h, x.w, x.c
        T = h * w
        if c % 32 != 0 or T % 4 != 0:
            raise RuntimeError(f"gpemsr_amd: non-local block needs channels ({c}) % 32 == 0 and latent tokens ({T}) % 4 == 0")
        if self.bf16:
            # the flash kernel has no row-length limit (C = 512, T % 128 == 0); the layered form's bf16 softmax holds <= 16384 columns
            if T % 16 == 0 and (T <= self.BF16_SOFTMAX_MAX_COLS or (self.flash_attn and self.o.flash_attention_ok(T, c))):
                return self._nonlocal_bf16(x, p)
            # token counts the bf16 matrix-product tiles cannot take (e.g. CREMI's 156 x 156 LR slices -> 78 x 78 = 6084 tokens) or
            # rows longer than the bf16 softmax kernel holds in registers: this one block runs on the exact-fp32 kernels (zero-padded
            # score rows; they take what the fp32 path takes), the rest of the path stays bf16
            return self.o.cast_bf16(self._nonlocal_ragged(self.o.cast_f32(x), p, precision="fp32"))
        if T % 32 != 0:
            return self._nonlocal_ragged(x, p)
        hn = self.o.groupnorm_relu(x, self.par[p + ".gn.weight"], self.par[p + ".gn.bias"], relu=False)
        q = self.conv(hn, p + ".q")                       # already scaled by C^-1/2
        k = self.conv(hn, p + ".k")
        gh, gw = T // 32, 32                               # GEMM rows as 32-wide "images" (the conv tile is 4x32 pixels)
        # V^T[c][j] = sum_c' Wv[c][c'] hn[j][c']  (bias folded into the PV product: softmax rows sum to 1)
        wv = self.pc[p + ".v"]
        vT = self._vt(wv.w, hn, n, c, T)
        out = self.o.new_act(n, h, w, c, device=self.dev)
        fc = max(1, min(n, (1 << 30) // (T * T * 4)))      # frames per score-matrix chunk (<= 1 GiB)
        for f0 in range(0, n, fc):
            m = min(fc, n - f0)
            qa = q.images(f0, m).reshape_hw(gh, gw)
            kf, vf = k.images(f0, m), vT.images(f0, m)
            # bf16x3 / bf16: the B operands (k, v^T) are activations, so they are split + re-ordered on the device
            k16 = self.o.split_pack_rows(kf) if self.split else None
            S = self.o.conv2d([qa], self.o.PackedConv(kf.buf, None, 1, T, (c,), 32, w16=k16), ACT_NONE,
                           weight_image_stride=T * c, tag=p + ".qk", precision=self.precision)
            del k16
            self.o.softmax_rows_(S.buf, m * T, T)
            v16 = self.o.split_pack_rows(vf) if self.split else None
            self.o.conv2d([S], self.o.PackedConv(vf.buf, wv.b, 1, c, (T,), 32, w16=v16), ACT_NONE,
                       weight_image_stride=c * T, out=out.images(f0, m).reshape_hw(gh, gw), tag=p + ".pv",
                       precision=self.precision)
            del v16
        return self.conv(out, p + ".proj_out", ACT_NONE, residual=x)

    def _nonlocal_bf16(self, x: Act, p: str) -> Act:
        """model/blocks.py:61-83 on the bf16 path.  q, k and v^T are 1x1 products; k and v^T leave their epilogues already in
        the B-operand layout of the next product ("kpack"), the score matrix and P are bf16, softmax arithmetic is fp32."""
        n, h, w, c = x.n, x.h, x.w, x.c
        T = h * w
        if T % 16 != 0:
            raise RuntimeError(f"gpemsr_amd: the bf16 non-local block needs latent tokens ({T}) % 16 == 0")
        o = self.o
        hn = o.groupnorm_relu(x, self.par[p + ".gn.weight"], self.par[p + ".gn.bias"], relu=False)
        q = self.conv(hn, p + ".q")                                    # [n][T][C], C^-1/2 folded in
        kp = self.conv(hn, p + ".k", kpack=True)                        # [n][C/8][T][8]
        wv = self.pc[p + ".v"]
        if not hasattr(wv, "_wa"):                                      # W_v as the A operand of v^T = W_v . hn^T: [C rows][C]
            wv._wa = self.sd[p + ".v.weight"].detach().reshape(c, c).to(torch.bfloat16).contiguous().to(self.dev)
        flash = self.flash_attn and o.flash_attention_ok(T, c)
        hnp = o.pack_rows_bf16(hn, perm16=flash)                        # hn as B operand: [n][C/8][T][8]
        wa = Act(wv._wa, n, 1, c, c, c, 0)                              # n "images" that alias the one weight matrix
        vtp = o.conv2d([wa], o.PackedConv(None, None, 1, T, (c,), 32, wb=hnp), ACT_NONE, weight_image_stride=T * c,
                       src_image_stride=[0], kpack=True, tag="nonlocal.vT", precision="bf16")     # v^T: [n][T/8][C][8]
        del hnp, hn
        if flash:
            # one kernel: q.k^T, online softmax, P.v -- the T x T score matrix stays on chip (csrc/attn_bf16.hip)
            out = o.flash_attention_bf16(q, kp, vtp, wv.b, tag=p + ".flash")
            del kp, vtp, q
            return self.conv(out, p + ".proj_out", ACT_NONE, residual=x)
        out = o.new_act(n, h, w, c, device=self.dev, bf16=True)
        fc = max(1, min(n, (1 << 30) // (T * T)))                        # frames per score-matrix chunk (<= 2 GiB of bf16)
        for f0 in range(0, n, fc):
            m = min(fc, n - f0)
            S = o.conv2d([q.images(f0, m)], o.PackedConv(None, None, 1, T, (c,), 32, wb=kp[f0:f0 + m]), ACT_NONE,
                         weight_image_stride=T * c, tag=p + ".qk", precision="bf16")
            P = o.softmax_rows_bf16(S)
            o.conv2d([P], o.PackedConv(None, wv.b, 1, c, (T,), 32, wb=vtp[f0:f0 + m]), ACT_NONE, weight_image_stride=c * T,
                     out=out.images(f0, m), tag=p + ".pv", precision="bf16")
            del S, P
        return self.conv(out, p + ".proj_out", ACT_NONE, residual=x)

    def _nonlocal_ragged(self, x: Act, p: str, precision: Optional[str] = None) -> Act:
        """The same block when the token count is not a multiple of 32 (e.g. 24x40 LR tiles -> 12x20 latents): the score rows
        and v^T rows are padded with zeros to the GEMM's 32-column granule (row stride Tp), the row softmax runs over the T
        real columns; the products stay on the exact f32 kernel in every precision mode."""
        n, h, w, c = x.n, x.h, x.w, x.c
        T = h * w
        Tp = (T + 31) // 32 * 32
        hn = self.o.groupnorm_relu(x, self.par[p + ".gn.weight"], self.par[p + ".gn.bias"], relu=False)
        kw = {} if precision is None else {"precision": precision}
        q = self.conv(hn, p + ".q", **kw)
        k = self.conv(hn, p + ".k", **kw)
        wv = self.pc[p + ".v"]
        vT = Act(torch.zeros(n * c * Tp, dtype=torch.float32, device=self.dev), n, c // 32, 32, T, Tp, 0)
        a = Act(wv.w, n, c // 32, 32, c, c, 0)
        self.o.conv2d([a], self.o.PackedConv(hn.buf, None, 1, T, (c,), 32), ACT_NONE, weight_image_stride=T * c,
                      src_image_stride=[0], out=vT, tag="nonlocal.vT")
        out = self.o.new_act(n, h, w, c, device=self.dev)
        fc = max(1, min(n, (1 << 30) // (T * Tp * 4)))
        for f0 in range(0, n, fc):
            m = min(fc, n - f0)
            S = Act(torch.zeros(m * T * Tp, dtype=torch.float32, device=self.dev), m, h, w, T, Tp, 0)
            self.o.conv2d([q.images(f0, m)], self.o.PackedConv(k.images(f0, m).buf, None, 1, T, (c,), 32), ACT_NONE,
                          weight_image_stride=T * c, out=S, tag=p + ".qk")
            self.o.softmax_rows_(S.buf, m * T, T, Tp)
            Sp = Act(S.buf, m, h, w, Tp, Tp, 0)               # padded columns are zeros and meet zero rows of v^T
            self.o.conv2d([Sp], self.o.PackedConv(vT.images(f0, m).buf, wv.b, 1, c, (Tp,), 32), ACT_NONE,
                          weight_image_stride=c * Tp, out=out.images(f0, m), tag=p + ".pv")
        return self.conv(out, p + ".proj_out", ACT_NONE, residual=x, **kw)

    def _vt(self, wv_packed: torch.Tensor, hn: Act, n: int, c: int, T: int) -> Act:
        vT = self.o.new_act(n, c // 32, 32, T, device=self.dev)
        a = Act(wv_packed, n, c // 32, 32, c, c, 0)        # n "images" that all alias the one [C][C] weight matrix
        self.o.conv2d([a], self.o.PackedConv(hn.buf, None, 1, T, (c,), 32), ACT_NONE, weight_image_stride=T * c,
                   src_image_stride=[0], out=vT, tag="nonlocal.vT")
        return vT

    def vq_layer(self, x: Act, p: str) -> Act:
        if (p + ".block.0") in self.pc:
            return self.vq_resblock(x, p)
        if (p + ".downblock") in self.pc:
            return self.conv(x, p + ".downblock", stride=2)
        if (p + ".upblock") in self.pc:
            return self.conv(x, p + ".upblock")
        if (p + ".q") in self.pc:
            return self.nonlocal_block(x, p)
        if p in self.pc:
            return self.conv(x, p)
        raise KeyError(p)

    def indexer_logits(self, xf: Act, argmax: bool = False):
        """Indexer8/16.forward (R:model/indexer.py:98-102).  argmax=True (bf16 path, three-product logits GEMM): returns the int32 code
        indices instead -- the arg-max rides in the GEMM's epilogue and the [cells][1024] logits never reach memory."""
        p = "refmodel.indexer"
        h = self.conv(xf, p + ".input_layer.0", ACT_RELU)
        for i in range(_seq_len(self.sd, p + ".feat_extract")):
            h = self.vq_layer(h, f"{p}.feat_extract.{i}")
        n_out = _seq_len(self.sd, p + ".output_layer")
        hp0 = self._hp_first_unit()
        for i in range(n_out):
            name = f"{p}.output_layer.{i}"
            if i >= hp0:                                   # the tail at `indexer_precision`: fp32 activations from here on
                if h.bf16:
                    h = self.o.cast_f32(h)
                h = self.vq_resblock(h, name, precision=self._hp_mode) if (name + ".block.0") in self.pc else self.conv(h, name, precision=self._hp_mode)
            elif self.bf16 and i == n_out - 1 and name in self.pc:
                h = self.conv(h, name, out_f32=True)       # the logits GEMM + argmax stay fp32
            else:
                h = self.vq_layer(h, name)
        if (p + ".embedding@x3") in self.pc and not h.bf16:
            # nn.Linear (indexer.py:100) at fp32 precision on the bf16 matrix pipe: operands split hi + lo, three products, fp32
            # accumulation (packing.pack_linear_bf16x3; ~2^-16 of the logit scale, the bf16 activations upstream move logits by ~1e-2)
            hi, lo = self.o.split_hi_lo_bf16(h)
            del h
            if argmax:
                return self.o.conv2d([hi, lo, hi], self.pc[p + ".embedding@x3"], ACT_NONE, tag=p + ".embedding+argmax", precision="bf16", argmax=True)
            return self.o.conv2d([hi, lo, hi], self.pc[p + ".embedding@x3"], ACT_NONE, tag=p + ".embedding", precision="bf16", out_f32=True)
        if h.bf16:
            h = self.o.cast_f32(h)
        return self.conv(h, p + ".embedding", precision="fp32")    # nn.Linear on NHWC == 1x1 conv (indexer.py:100)

    def ref_extract(self, xf: Act, forced_idx: Optional[torch.Tensor], trace: Optional[dict]) -> List[Act]:
        s = self.scale
        ln, lh_, lw_ = xf.n, (xf.h // 2 if s == 8 else xf.h), (xf.w // 2 if s == 8 else xf.w)     # latent grid (R:model/indexer.py:78-79: x8 halves once)
        fused = (self.bf16 and self.fuse_argmax and forced_idx is None and trace is None and ("refmodel.indexer.embedding@x3") in self.pc
                 and self.o is ops)
        if self._idx_engine is not None and forced_idx is None:
            logits = self._idx_engine.indexer_logits(xf)           # the whole indexer at exact fp32 (`indexer_precision: fp32:all`)
            idx = self.o.argmax_rows(logits)
            if trace is not None:
                trace.setdefault("logits", []).append(logits.torch().clone())
                trace.setdefault("code_idx", []).append(idx.clone())
            del logits
        elif fused:
            idx = self.indexer_logits(xf, argmax=True)             # arg-max in the logits GEMM's epilogue: no [cells][1024] tensor
        else:
            logits = self.indexer_logits(xf)
            assert (logits.n, logits.h, logits.w) == (ln, lh_, lw_)
            idx = self.o.argmax_rows(logits) if forced_idx is None else forced_idx.to(torch.int32).contiguous()
            if trace is not None:
                trace.setdefault("logits", []).append(logits.torch().clone())
                trace.setdefault("code_idx", []).append(idx.clone())
            del logits
        gather = self.o.gather_rows_bf16 if self.bf16 else self.o.gather_rows
        x = gather(self.par["refmodel.codebook.embedding.weight"], idx, ln, lh_, lw_)
        p = "refmodel.decoder"
        for i in range(_seq_len(self.sd, p + ".input_layer")):
            x = self.vq_layer(x, f"{p}.input_layer.{i}")
        n_fe = _seq_len(self.sd, p + ".feat_extract")
        feats = []
        x = self.vq_layer(x, f"{p}.feat_extract.0")
        nrb = self.dec_nrb
        # bf16: the last up-block (64 -> 64) and the output layer (64 -> 1) have nothing in between and the 64-channel tensor at
        # the full resolution has no other reader: they run as one composed operator (csrc/tap_sum.hip)
        last = f"{p}.feat_extract.{n_fe - 1}.upblock"
        fuse_tail = (getattr(self, "fuse_tail_f32", True) or self.bf16) and (n_fe >= 2 and last in self.pc and tuple(self.sd[last + ".weight"].shape) == (64, 64, 3, 3)
                     and tuple(self.sd[p + ".output_layer.weight"].shape) == (1, 64, 3, 3) and (n_fe - 2 - nrb + 1) % (nrb + 1) != 0)
        for i in range(n_fe - 1 - int(fuse_tail)):
            x = self.vq_layer(x, f"{p}.feat_extract.{i + 1}")
            if (i - nrb + 1) % (nrb + 1) == 0:
                feats.append(x)
        if fuse_tail:
            if "@upout.frag" not in self.par:
                packer = pack_upconv_out if self.bf16 else pack_upconv_out_f32
                self.par["@upout.frag"], self.par["@upout.consts"] = packer(
                    self.sd[last + ".weight"], self.sd.get(last + ".bias"), self.sd[p + ".output_layer.weight"],
                    self.sd.get(p + ".output_layer.bias"), self.dev)
            tail = self.o.upconv_out_bf16 if self.bf16 else self.o.upconv_out_f32
            feats.append(tail(x, self.par["@upout.frag"], self.par["@upout.consts"], tag=last + "+output_layer"))
        else:
            feats.append(self.conv(x, p + ".output_layer"))
        return feats

    # ------------------------------------------------------------------ mask
    def vgg_mask(self, ref_img: Act, up_lr: Act) -> Act:
        """model/GPEMSR.py:386-395 for a chunk of frames -> cosine map [n,sH/16,sW/16,1]."""
        n = ref_img.n
        out = self.o.new_act(n, ref_img.h // 16, ref_img.w // 16, 1, device=self.dev)
        per = max(1, (1 << 30) // (ref_img.h * ref_img.w * 64 * 4))
        for i0 in range(0, n, per):
            m = min(per, n - i0)
            fa = self.conv(self.conv(ref_img.images(i0, m), "vgg.slice1.0", ACT_RELU), "vgg.slice1.2", ACT_RELU)
            tb = self.conv(up_lr.images(i0, m), "vgg.slice1.0", ACT_RELU)
            if self.vgg_cos_epi and self.o.conv_cosine_ok(tb, self.pc["vgg.slice1.2"]) and fa.ld % 4 == 0:
                # the second relu1_2 map never reaches memory: its patch sums against the first come from the convolution's epilogue
                o = self.conv(tb, "vgg.slice1.2", ACT_RELU, cos_with=fa)
            else:
                fb = self.conv(tb, "vgg.slice1.2", ACT_RELU)
                o = self.o.patch_cosine(fa, fb)
                del fb
            del tb
            self.o.copy_channels(o, out.images(i0, m))
        return out

    def vgg_mask_fused(self, ref_img: Act, xf: Act) -> Act:
        """The same from the LR slices themselves, in one kernel (bf16 path): no up-sampled image, no feature maps in HBM."""
        if "vgg.w1" not in self.par:
            w = self.sd["vgg.slice1.0.weight"].detach().to(torch.float32)
            self.par["vgg.w1"] = w.sum(dim=1).reshape(w.shape[0], 9).contiguous().to(self.dev)      # 3 identical input channels
            self.par["vgg.b1"] = self.sd["vgg.slice1.0.bias"].detach().to(torch.float32).contiguous().to(self.dev)
        pc2 = self.pc["vgg.slice1.2"]
        if os.environ.get("GPEMSR_VGG_UPLR", "1") != "0":
            # the LR slice is up-sampled ONCE (gpemsr_bilinear, fp32 1-channel: 4 MB per slice) and the fused kernel reads both images the
            # same way; resampling it on the fly inside the kernel's producer waves cost 4 ms per step (round 3, in-kernel stamps)
            up = self.o.bilinear(xf, self.scale * xf.h, self.scale * xf.w)
            return self.o.vgg_mask_bf16(ref_img, up, 1, self.par["vgg.w1"], self.par["vgg.b1"], pc2.wb, pc2.b)
        return self.o.vgg_mask_bf16(ref_img, xf, self.scale, self.par["vgg.w1"], self.par["vgg.b1"], pc2.wb, pc2.b)

    # ------------------------------------------------------------------ per-frame front half
    def front(self, xf: Act, forced_idx, trace) -> Dict[str, Act]:
        s, H, W = self.scale, xf.h, xf.w
        L1 = self.conv(xf, "conv_first", ACT_LRELU)
        L1 = self.resblocks_nobn(L1, "feature_extraction")
        if trace is not None:
            trace.setdefault("L1_fea", []).append(L1.nchw())
        Lr2 = self.conv(L1, "reffea_L2_conv1", ACT_LRELU)
        Lr3 = self.conv(Lr2, "reffea_L3_conv1", ACT_LRELU)
        Lr4 = self.conv(Lr3, "reffea_L4_conv1", ACT_LRELU) if s == 16 else None
        ref_x16, ref_x8, ref_x4, ref_x2, ref_img = self.ref_extract(xf, forced_idx, trace)
        if (self.bf16 or getattr(self, "_frozen16", None) is not None) and (s * H) % 16 == 0 and (s * W) % 16 == 0 and self.fuse_vgg:
            mask = self.vgg_mask_fused(ref_img, xf)
        else:
            up_lr = self.o.bilinear(xf, s * H, s * W)
            mask = self.vgg_mask(ref_img, up_lr)
            del up_lr
        if trace is not None:
            trace.setdefault("mask_cos", []).append(mask.nchw())
        mask = self.conv(mask, "refmaskconv1", ACT_LRELU)
        mask = self.conv(mask, "refmaskconv2", ACT_LRELU)
        mask = self.conv(mask, "refmaskconv3", ACT_LRELU_SIGMOID)
        mh, mw = mask.h, mask.w
        if s == 16:
            fine, mid, coarse = Lr4, Lr3, Lr2
        else:
            fine, mid, coarse = Lr3, Lr2, L1
        r2 = self.conv([fine, ref_x2], "reffusionconv1")
        r2 = self.resblocks_nobn(r2, "fusion_fea_block1", pixmul=self.o.bilinear(mask, mh * 8, mw * 8))
        r2 = self.conv(r2, "down_fea_conv1", stride=2)
        r4 = self.conv([mid, ref_x4, r2], "reffusionconv2")
        r4 = self.resblocks_nobn(r4, "fusion_fea_block2", pixmul=self.o.bilinear(mask, mh * 4, mw * 4))
        r4 = self.conv([r4, r2], "down_fea_conv2", stride=2)
        r8 = self.conv([coarse, ref_x8, r4], "reffusionconv3")
        r8 = self.resblocks_nobn(r8, "fusion_fea_block3", pixmul=self.o.bilinear(mask, mh * 2, mw * 2))
        if s == 16:
            r8 = self.conv([r8, r4], "down_fea_conv3", stride=2)
            r16 = self.conv([L1, ref_x16, r8], "reffusionconv4")
            r16 = self.resblocks_nobn(r16, "fusion_fea_block4", pixmul=mask)
            L1 = self.conv([r16, r8, L1], "reduce_dim_conv")
        else:
            L1 = self.conv([r8, r4, L1], "reduce_dim_conv")
        L2 = self.conv(self.conv(L1, "fea_L2_conv1", ACT_LRELU, stride=2), "fea_L2_conv2", ACT_LRELU)
        L3 = self.conv(self.conv(L2, "fea_L3_conv1", ACT_LRELU, stride=2), "fea_L3_conv2", ACT_LRELU)
        return {"L1": L1, "L2": L2, "L3": L3, "ref_img": ref_img}

    # ------------------------------------------------------------------ SpyNet + POD
    def spynet(self, ref: Act, supp: Act) -> Act:
        """basicsr SpyNet.forward(ref, supp) on 1-channel frames -> flow [n,h,w,2] (x,y)."""
        h, w = ref.h, ref.w
        hf, wf = ((h + 31) // 32) * 32, ((w + 31) // 32) * 32
        if (hf, wf) != (h, w):
            ref, supp = self.o.bilinear(ref, hf, wf), self.o.bilinear(supp, hf, wf)
        rp, sp = [ref], [supp]
        for _ in range(5):
            rp.insert(0, self.o.avgpool2(rp[0])); sp.insert(0, self.o.avgpool2(sp[0]))
        flow = None
        p = "align_module.spynet.basic_module"
        for lvl in range(6):
            if self.bf16:
                up, inp = self.o.spynet_prep_bf16(rp[lvl], sp[lvl], flow, self.spy_mean, self.spy_std)
            else:
                up, inp = self.o.spynet_prep(rp[lvl], sp[lvl], flow, self.spy_mean, self.spy_std, pad16=self.split)
            t = self.conv(inp, f"{p}.{lvl}.basic_module.0", ACT_RELU)
            t = self.conv(t, f"{p}.{lvl}.basic_module.2", ACT_RELU)
            t = self.conv(t, f"{p}.{lvl}.basic_module.4", ACT_RELU)
            t = self.conv(t, f"{p}.{lvl}.basic_module.6", ACT_RELU)
            flow = self.conv(t, f"{p}.{lvl}.basic_module.8", ACT_NONE, residual=up, out_f32=self.bf16)   # flows stay fp32
        if (hf, wf) != (h, w):
            out = self.o.new_act(flow.n, h, w, 2, device=self.dev)
            self.o.bilinear(flow.slice(0, 1), h, w, mul=float(w) / float(wf), out=out.slice(0, 1))
            self.o.bilinear(flow.slice(1, 1), h, w, mul=float(h) / float(hf), out=out.slice(1, 1))
            flow = out
        return flow

    def dcn(self, x: Act, feat: Act, name: str, act: int) -> Act:
        om = self.conv(feat, name + ".conv_offset", force_mfma=True, out_f32=self.bf16)    # sampling coordinates stay fp32
        if self.bf16 and self.fuse_dcn and self.o is ops and ops.dcn_conv_ok(x, om, self.pc[name], self.groups):
            # deformable sampling and the 64 x 576 contraction in one kernel: the column tensor stays in LDS (csrc/dcn_bf16.hip)
            return ops.dcn_conv_bf16(x, om, self.pc[name], act, tag=name)
        col = self.o.dcn_columns(x, om, self.groups)
        return self.conv(col, name, act)

    def pod(self, nbr: List[Act], ref: List[Act], nbr_frame: Act, ref_frame: Act, trace) -> Act:
        """POD.forward (model/GPEMSR.py:98-140) for P = 5*B' (neighbour, centre) pairs at once."""
        p = "align_module"
        P, H, W = nbr_frame.n, nbr_frame.h, nbr_frame.w
        if self._forced_flow is not None:      # teacher-forced SpyNet output [P,4H,4W,2] (gradient parity tests)
            flow = self._forced_flow
            assert (flow.n, flow.h, flow.w, flow.c) == (P, 4 * H, 4 * W, 2)
        else:
            flow = self.spynet(self.o.bilinear(nbr_frame, 4 * H, 4 * W), self.o.bilinear(ref_frame, 4 * H, 4 * W))
        if trace is not None:
            trace.setdefault("flow", []).append(flow.nchw())
        ffc = self._ffc

        def flow_frames(h, w):      # [flow1 16 | flow2 16 | nbr_frame | ref_frame | zeros up to ffc]
            return self.o.new_act(P, h, w, ffc, device=self.dev, bf16=self.bf16, zero=True)
        fl1 = flow_frames(H, W)
        self.conv(flow, p + ".flowdsconv0_1", stride=4, out=fl1.slice(0, 16))
        self.conv(flow, p + ".flowdsconv0_2", stride=4, out=fl1.slice(16, 16))
        fl2 = flow_frames(H // 2, W // 2)
        self.conv(fl1.slice(0, 16), p + ".flowdsconv1_1", stride=2, out=fl2.slice(0, 16))
        self.conv(fl1.slice(16, 16), p + ".flowdsconv1_2", stride=2, out=fl2.slice(16, 16))
        fl3 = flow_frames(H // 4, W // 4)
        self.conv(fl2.slice(0, 16), p + ".flowdsconv2_1", stride=2, out=fl3.slice(0, 16))
        self.conv(fl2.slice(16, 16), p + ".flowdsconv2_2", stride=2, out=fl3.slice(16, 16))
        if self.bf16:
            # the two frames are 1-channel fp32 images: resize them in fp32 (channel by channel == the 2-channel resize of
            # model/GPEMSR.py:107-110), then drop them into the bf16 concat buffers
            for fl, (hh, ww) in ((fl1, (H, W)), (fl2, (H // 2, W // 2)), (fl3, (H // 4, W // 4))):
                if (hh, ww) != (H, W):
                    nbr_frame, ref_frame = self.o.bilinear(nbr_frame, hh, ww), self.o.bilinear(ref_frame, hh, ww)
                self.o.copy_channels_f32_bf16(nbr_frame, fl.slice(32, 1)); self.o.copy_channels_f32_bf16(ref_frame, fl.slice(33, 1))
        else:
            fr1 = fl1.slice(32, 2)
            self.o.copy_channels(nbr_frame, fr1.slice(0, 1)); self.o.copy_channels(ref_frame, fr1.slice(1, 1))
            self.o.bilinear(fr1, H // 2, W // 2, out=fl2.slice(32, 2))
            self.o.bilinear(fl2.slice(32, 2), H // 4, W // 4, out=fl3.slice(32, 2))

        o3 = self.conv([nbr[2], ref[2], fl3], p + ".L3_offset_conv1", ACT_LRELU)
        o3 = self.conv(o3, p + ".L3_offset_conv2", ACT_LRELU)
        f3 = self.dcn(nbr[2], o3, p + ".L3_dcnpack", ACT_LRELU)

        o2 = self.conv([nbr[1], ref[1], fl2], p + ".L2_offset_conv1", ACT_LRELU)
        o3u = self.o.bilinear(o3, o3.h * 2, o3.w * 2, mul=2.0)
        o2 = self.conv([o2, o3u], p + ".L2_offset_conv2", ACT_LRELU)
        o2 = self.conv(o2, p + ".L2_offset_conv3", ACT_LRELU)
        f2 = self.dcn(nbr[1], o2, p + ".L2_dcnpack", ACT_NONE)
        f3u = self.o.bilinear(f3, f3.h * 2, f3.w * 2)
        f2 = self.conv([f2, f3u], p + ".L2_fea_conv", ACT_LRELU)

        o1 = self.conv([nbr[0], ref[0], fl1], p + ".L1_offset_conv1", ACT_LRELU)
        o2u = self.o.bilinear(o2, o2.h * 2, o2.w * 2, mul=2.0)
        o1 = self.conv([o1, o2u], p + ".L1_offset_conv2", ACT_LRELU)
        o1 = self.conv(o1, p + ".L1_offset_conv3", ACT_LRELU)
        f1 = self.dcn(nbr[0], o1, p + ".L1_dcnpack", ACT_NONE)
        f2u = self.o.bilinear(f2, f2.h * 2, f2.w * 2)
        f1 = self.conv([f1, f2u], p + ".L1_fea_conv", ACT_NONE)

        off = self.conv([f1, ref[0]], p + ".cas_offset_conv1", ACT_LRELU)
        off = self.conv(off, p + ".cas_offset_conv2", ACT_LRELU)
        return self.dcn(f1, off, p + ".cas_dcnpack", ACT_LRELU)

    # ------------------------------------------------------------------ ThreeDA
    def three_da(self, aligned: Act, B: int) -> Act:
        """ThreeDA.forward (model/GPEMSR.py:172-222); aligned is [B*N,h,w,c], frame-major per tile."""
        p, N = "ThreeDA", self.N
        centre = self.o.copy_images(aligned, B, 1, N, self.center)
        emb_ref = self.conv(centre, p + ".temporal_attn1")
        emb = self.conv(aligned, p + ".temporal_attn2")
        af = self.o.temporal_gate(aligned, emb, emb_ref, B, N)
        m1 = self.o.frame_mix_lrelu(af, N, self.par[p + ".conv3D_1.weight"], self.par[p + ".conv3D_1.bias"])
        f1 = self.conv(m1, p + ".conv3D_fusion_1", ACT_LRELU)
        m2 = self.o.frame_mix_lrelu(af, N, self.par[p + ".conv3D_2.weight"], self.par[p + ".conv3D_2.bias"])
        f2 = self.conv(m2, p + ".conv3D_fusion_2", ACT_LRELU)
        feat = self.conv(af, p + ".feat_fusion", ACT_LRELU, residual=f1)
        f3 = self.conv(feat, p + ".conv2D_fusion_3")
        attn = self.conv(af, p + ".spatial_attn1", ACT_LRELU)
        attn = self.conv(self.o.pool3s2_maxavg(attn), p + ".spatial_attn2", ACT_LRELU)
        lvl = self.conv(attn, p + ".spatial_attn_l1", ACT_LRELU)
        lvl = self.conv(self.o.pool3s2_maxavg(lvl), p + ".spatial_attn_l2", ACT_LRELU)
        lvl = self.conv(lvl, p + ".spatial_attn_l3", ACT_LRELU)
        lvl = self.o.bilinear(lvl, lvl.h * 2, lvl.w * 2)
        attn = self.conv(attn, p + ".spatial_attn3", ACT_LRELU, residual=lvl)
        attn = self.conv(attn, p + ".spatial_attn4", ACT_LRELU)
        attn = self.o.bilinear(attn, attn.h * 2, attn.w * 2)
        attn = self.conv(attn, p + ".spatial_attn5")
        add = self.conv(self.conv(attn, p + ".spatial_attn_add1", ACT_LRELU), p + ".spatial_attn_add2")
        return self.o.threeda_combine(feat, attn, add, f2, f3)

    # ------------------------------------------------------------------ whole forward
    def _check_lr(self, H: int, W: int):
        # the L2 / L3 pyramid and the x8 latent grid (H/2 x W/2) halve twice (model/GPEMSR.py:395,424-426); CREMI's 156 x 156 x8 LR
        # slices are a multiple of 4 but not of 8
        if H % 4 != 0 or W % 4 != 0 or H < 4 or W < 4:
            raise RuntimeError(f"gpemsr_amd: LR height/width must be multiples of 4 (got {H} x {W})")

    def _front_all(self, xa: Act, forced_idx, trace):
        """Per-frame half (everything up to the L1/L2/L3 pyramid, model/GPEMSR.py:325-426) for all frames of ``xa``."""
        nfr, H, W, s = xa.n, xa.h, xa.w, self.scale
        L1 = self.o.new_act(nfr, H, W, 64, device=self.dev, bf16=self.bf16)
        L2 = self.o.new_act(nfr, H // 2, W // 2, 64, device=self.dev, bf16=self.bf16)
        L3 = self.o.new_act(nfr, H // 4, W // 4, 64, device=self.dev, bf16=self.bf16)
        ref_img = torch.empty(nfr, 1, H * s, W * s, dtype=torch.float32, device=self.dev)
        ref_act = Act(ref_img, nfr, H * s, W * s, 1, 1, 0)
        lat = (H // 2) * (W // 2) if s == 8 else H * W
        for f0 in range(0, nfr, self.frame_chunk):
            m = min(self.frame_chunk, nfr - f0)
            fi = None if forced_idx is None else forced_idx.reshape(-1)[f0 * lat:(f0 + m) * lat]
            r = self.front(xa.images(f0, m), fi, trace)
            self.o.copy_channels(r["L1"], L1.images(f0, m)); self.o.copy_channels(r["L2"], L2.images(f0, m))
            self.o.copy_channels(r["L3"], L3.images(f0, m)); self.o.copy_channels(r["ref_img"], ref_act.images(f0, m))
            del r
        if trace is not None:
            trace["L1_fused"] = L1.nchw()
        return (L1, L2, L3), ref_img

    def _back(self, xa: Act, pyr, windows: torch.Tensor, trace) -> torch.Tensor:
        """Per-window half (POD alignment, ThreeDA, reconstruction, upsampler; model/GPEMSR.py:427-455).
        ``windows`` [Wn, N] int32 (device): frame numbers of every window; the centre is column N // 2."""
        N, s, H, W = self.N, self.scale, xa.h, xa.w
        Wn = windows.shape[0]
        out = torch.empty(Wn, 1, H * s, W * s, dtype=torch.float32, device=self.dev)
        out_act = Act(out, Wn, H * s, W * s, 1, 1, 0)
        self._last_out_act = out_act
        for b0 in range(0, Wn, self.tile_chunk):
            bm = min(self.tile_chunk, Wn - b0)
            win = windows[b0:b0 + bm]
            nbr_idx = win.reshape(-1).contiguous()
            cen_idx = win[:, self.center].contiguous()
            ref_idx = cen_idx.repeat_interleave(N).contiguous()
            nbr = [self.o.gather_images(t, nbr_idx) for t in pyr]
            ref = [self.o.gather_images(t, ref_idx) for t in pyr]
            nbr_frame = self.o.gather_images(xa, nbr_idx)
            ref_frame = self.o.gather_images(xa, ref_idx)
            aligned = self.pod(nbr, ref, nbr_frame, ref_frame, trace)
            if trace is not None:
                trace.setdefault("aligned", []).append(aligned.nchw().view(bm, N, 64, H, W))
            fea = self.three_da(aligned, bm)
            if trace is not None:
                trace.setdefault("fused", []).append(fea.nchw())
            o = self.resblocks_nobn(fea, "recon_trunk")
            if trace is not None:
                trace.setdefault("recon", []).append(o.nchw())
            o = self.conv(o, "upconv1", ACT_LRELU)
            if trace is not None:
                trace.setdefault("up1", []).append(o.nchw())
            o = self.conv(o, "upconv2", ACT_LRELU)
            o = self.conv(o, "upconv3", ACT_LRELU)
            if s == 16:
                o = self.conv(o, "upconv4", ACT_LRELU)
            if trace is not None:
                trace.setdefault("up_last", []).append(o.nchw())
            o = self.conv(o, "HRconv", ACT_LRELU)
            if trace is not None:
                trace.setdefault("hr", []).append(o.nchw())
            base = self.o.bilinear(self.o.gather_images(xa, cen_idx), H * s, W * s)
            u8 = getattr(self, "_u8_out", None)          # uint8 image straight from the last kernel (conv_last + base + tensor2img)
            self.conv(o, "conv_last", ACT_NONE, residual=base, out=out_act.images(b0, bm), out_u8=None if u8 is None else u8[b0:b0 + bm])
        return out

    def with_u8(self, fn, n_out: int, H: int, W: int):
        """Run ``fn`` (a forward) with the uint8 image of the SR output produced by the network's last kernel; returns fn's results
        plus that image [n_out, sH, sW] (R:util/util.py:145-163 semantics; what output_GPEMSR.py writes)."""
        self._u8_out = torch.empty(n_out, H * self.scale, W * self.scale, dtype=torch.uint8, device=self.dev)
        try:
            res = fn()
            return tuple(res) + (self._u8_out,)
        finally:
            self._u8_out = None

    def forward(self, x: torch.Tensor, forced_idx: Optional[torch.Tensor] = None, trace: Optional[dict] = None):
        """GPEMSR.forward (model/GPEMSR.py:323-456): independent windows x[B, N, 1, H, W] -> (out, ref_img)."""
        if not x.is_cuda:
            raise RuntimeError("gpemsr_amd: forward needs a device (cuda/HIP) tensor; there is no CPU path")
        B, N, C, H, W = x.shape
        assert N == self.N and C == 1, "expected [B, nframes, 1, H, W]"
        self._check_lr(H, W)
        if B == 0:                                         # empty batch: nothing to launch
            sc = self.scale
            return (torch.empty(0, 1, H * sc, W * sc, dtype=torch.float32, device=self.dev),
                    torch.empty(0, N, C, H * sc, W * sc, dtype=torch.float32, device=self.dev))
        x = x.to(torch.float32).contiguous()
        xa = Act(x, B * N, H, W, 1, 1, 0)
        pyr, ref_img = self._front_all(xa, forced_idx, trace)
        windows = torch.arange(B * N, dtype=torch.int32, device=self.dev).view(B, N)
        out = self._back(xa, pyr, windows, trace)
        return out, ref_img.view(B, N, C, H * self.scale, W * self.scale)

    def forward_volume(self, frames: torch.Tensor, windows: torch.Tensor, forced_idx: Optional[torch.Tensor] = None):
        """Volume mode: the sliding windows of output_GPEMSR.py:54-128 share 4 of their 5 slices,
        and everything up to the L1/L2/L3 pyramid depends on one slice only, so the per-frame half runs ONCE per slice and
        each window gathers its frames' cached features.  Results are bit-identical to ``forward`` on the stacked windows
        (every kernel treats the images of a batch independently).
        frames [T, 1, H, W]; windows [Wn, N] integer frame numbers (edge windows repeat frames) -> (out [Wn, 1, sH, sW],
        ref_img [T, 1, sH, sW])."""
        if not frames.is_cuda:
            raise RuntimeError("gpemsr_amd: forward_volume needs a device (cuda/HIP) tensor; there is no CPU path")
        T, C, H, W = frames.shape
        assert C == 1, "expected [T, 1, H, W]"
        self._check_lr(H, W)
        windows = torch.as_tensor(windows)
        assert windows.dim() == 2 and windows.shape[1] == self.N, "windows must be [Wn, nframes]"
        assert int(windows.min()) >= 0 and int(windows.max()) < T, "window frame number out of range"
        frames = frames.to(torch.float32).contiguous()
        xa = Act(frames, T, H, W, 1, 1, 0)
        pyr, ref_img = self._front_all(xa, forced_idx, None)
        out = self._back(xa, pyr, windows.to(device=self.dev, dtype=torch.int32).contiguous(), None)
        return out, ref_img
