"""Parameter inventory of the stage-3 GPEMSR network, derived from the
constructor arguments exactly as the reference derives its modules.

Reference (paths relative to /root/reference/GPEMSR-CREMI/GPEMSR/):
  model/GPEMSR.py:226-321 (GPEMSR.__init__), :64-96 (POD), :143-170 (ThreeDA),
  model/vqgan_indexer.py:60-67 (lrGenerator8) / :20-26 (lrGenerator16),
  model/indexer.py:6-48,58-96, model/decoder.py:7-33, model/encoder.py:5-36,
  model/codebook.py:6-13, model/blocks.py:5-60, model/VGG.py:13-28.

``param_specs(...)`` returns an ordered ``{state_dict key: ParamSpec}`` map.
It is the single source of truth for (a) the ``GPEMSR`` host module's
parameters, (b) the synthetic-weight generator and (c) the weight repacker.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class ParamSpec:
    shape: Tuple[int, ...]
    kind: str            # conv_w | convT_w | bias | gn_w | gn_b | linear_w | emb | conv3d_w | buf_mean | buf_std
    trainable: bool
    is_buffer: bool = False


_VGG_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]
_VGG_SLICES = ((0, 4), (4, 9), (9, 18), (18, 27), (27, 36))     # model/VGG.py:18-27


class _Spec(OrderedDict):
    def conv(self, name, cin, cout, k, trainable=True):
        self[name + ".weight"] = ParamSpec((cout, cin, k, k), "conv_w", trainable)
        self[name + ".bias"] = ParamSpec((cout,), "bias", trainable)

    def convT(self, name, cin, cout, trainable=True):
        self[name + ".weight"] = ParamSpec((cin, cout, 3, 3), "convT_w", trainable)
        self[name + ".bias"] = ParamSpec((cout,), "bias", trainable)

    def gn(self, name, c, trainable=False):
        self[name + ".weight"] = ParamSpec((c,), "gn_w", trainable)
        self[name + ".bias"] = ParamSpec((c,), "gn_b", trainable)

    def resblock_nobn(self, name, nf):                      # basicsr ResidualBlockNoBN
        self.conv(name + ".conv1", nf, nf, 3)
        self.conv(name + ".conv2", nf, nf, 3)

    def vq_resblock(self, name, cin, cout):                 # model/blocks.py:8-29
        self.conv(name + ".block.0", cin, cout, 3, False)
        self.gn(name + ".block.1", cout)
        self.conv(name + ".block.3", cout, cout, 3, False)
        self.gn(name + ".block.4", cout)
        if cin != cout:
            self.conv(name + ".channel_up", cin, cout, 1, False)

    def nonlocal_block(self, name, c):                      # model/blocks.py:50-60
        self.gn(name + ".gn", c)
        for n in ("q", "k", "v", "proj_out"):
            self.conv(f"{name}.{n}", c, c, 1, False)


def _indexer(spec: _Spec, p: str, a: dict, scale: int):
    """model/indexer.py: Indexer8 (:58-96) / Indexer16 (:6-48)."""
    cl = list(a["channel_list"])
    spec.conv(p + ".input_layer.0", a["im_channel"], cl[0], 3, False)
    nrb = a["num_resblock_per_scale"]
    li = 0
    down_at = 3 if scale == 8 else 4                         # :27 vs :78 ; i==4 never fires for 5 entries
    for i in range(len(cl) - 1):
        cin, cout = cl[i], cl[i + 1]
        for _ in range(nrb - 1):
            spec.vq_resblock(f"{p}.feat_extract.{li}", cin, cin); li += 1
        if i == down_at:
            spec.conv(f"{p}.feat_extract.{li}.downblock", cin, cout, 3, False); li += 1
        else:
            spec.vq_resblock(f"{p}.feat_extract.{li}", cin, cout); li += 1
    if scale == 16 and len(cl) == 4:                         # :31-34 (dead for the shipped YAML)
        for _ in range(nrb - 1):
            spec.vq_resblock(f"{p}.feat_extract.{li}", cl[-1], cl[-1]); li += 1
        spec.convT(f"{p}.feat_extract.{li}.upblock", cl[-1], cl[-1], False); li += 1
    if a["use_non_local"]:
        spec.nonlocal_block(f"{p}.feat_extract.{li}", cl[-1]); li += 1
    oi = 0
    for _ in range(a["num_output_resblck"]):
        spec.vq_resblock(f"{p}.output_layer.{oi}", cl[-1], cl[-1]); oi += 1
    spec.conv(f"{p}.output_layer.{oi}", cl[-1], a["latent_dim"], 1, False)
    spec[p + ".embedding.weight"] = ParamSpec((1024, a["latent_dim"]), "linear_w", False)
    spec[p + ".embedding.bias"] = ParamSpec((1024,), "bias", False)


def _decoder(spec: _Spec, p: str, a: dict):
    """model/decoder.py:7-33."""
    cl = list(a["channel_list"])
    spec.conv(p + ".input_layer.0", a["latent_dim"], cl[0], 1, False)
    for i in range(a["num_input_resblck"]):
        spec.vq_resblock(f"{p}.input_layer.{i + 1}", cl[0], cl[0])
    li = 0
    if a["use_non_local"]:
        spec.nonlocal_block(f"{p}.feat_extract.{li}", cl[0]); li += 1
    for i in range(len(cl) - 1):
        for _ in range(a["num_resblock_per_scale"]):
            spec.vq_resblock(f"{p}.feat_extract.{li}", cl[i], cl[i]); li += 1
        spec.convT(f"{p}.feat_extract.{li}.upblock", cl[i], cl[i + 1], False); li += 1
    spec.conv(p + ".output_layer", cl[-1], a["im_channel"], 3, False)


def _encoder(spec: _Spec, p: str, a: dict):
    """model/encoder.py:5-36 -- weights only; never evaluated in stage 3."""
    cl = list(a["channel_list"])
    spec.conv(p + ".input_layer.0", a["im_channel"], cl[0], 3, False)
    li = 0
    for i in range(len(cl) - 1):
        for _ in range(a["num_resblock_per_scale"]):
            spec.vq_resblock(f"{p}.feat_extract.{li}", cl[i], cl[i]); li += 1
        spec.conv(f"{p}.feat_extract.{li}.downblock", cl[i], cl[i + 1], 3, False); li += 1
    if a["use_non_local"]:
        spec.nonlocal_block(f"{p}.feat_extract.{li}", cl[-1]); li += 1
    oi = 0
    for _ in range(a["num_output_resblck"]):
        spec.vq_resblock(f"{p}.output_layer.{oi}", cl[-1], cl[-1]); oi += 1
    spec.conv(f"{p}.output_layer.{oi}", cl[-1], a["latent_dim"], 1, False)


def param_specs(argref: dict, nf: int = 64, nframes: int = 5, groups: int = 8, front_RBs: int = 5,
                back_RBs: int = 10, w_ref: bool = True, ref_fusion_feat_RBs: int = 3,
                align_mode: str = "POD", fusion_mode: str = "ThreeDA", mode: str = "16to1",
                scale: int = 16, **_ignored) -> "OrderedDict[str, ParamSpec]":
    if scale not in (8, 16):
        raise ValueError("scale is wrong!")                  # model/GPEMSR.py:286-287
    s = _Spec()
    s.conv("conv_first", 1, nf, 3)
    for i in range(front_RBs):
        s.resblock_nobn(f"feature_extraction.{i}", nf)
    if w_ref:
        # VGG19 (frozen): model/VGG.py
        layer_idx, cin, convs = 0, 3, {}
        for v in _VGG_CFG_E:
            if v == "M":
                layer_idx += 1
            else:
                convs[layer_idx] = (cin, v); cin = v; layer_idx += 2
        for si, (lo, hi) in enumerate(_VGG_SLICES):
            for idx in range(lo, hi):
                if idx in convs:
                    s.conv(f"vgg.slice{si + 1}.{idx}", convs[idx][0], convs[idx][1], 3, False)
        s.conv("refmaskconv1", 1, nf, 3)
        s.conv("refmaskconv2", nf, nf, 3)
        s.conv("refmaskconv3", nf, 1, 3)
        for l in (2, 3, 4):
            s.convT(f"reffea_L{l}_conv1", nf, nf)
        s.conv("reffusionconv1", nf + 64, nf, 3)
        for i in range(ref_fusion_feat_RBs):
            s.resblock_nobn(f"fusion_fea_block1.{i}", nf)
        s.conv("down_fea_conv1", nf, nf, 3)
        s.conv("reffusionconv2", 2 * nf + 128, nf, 3)
        for i in range(ref_fusion_feat_RBs):
            s.resblock_nobn(f"fusion_fea_block2.{i}", nf)
        s.conv("down_fea_conv2", 2 * nf, 2 * nf, 3)
        s.conv("reffusionconv3", 3 * nf + 256, nf, 3)
        for i in range(ref_fusion_feat_RBs):
            s.resblock_nobn(f"fusion_fea_block3.{i}", nf)
        s.conv("down_fea_conv3", 3 * nf, 3 * nf, 3)
        s.conv("reffusionconv4", 4 * nf + 512, nf, 3)
        for i in range(ref_fusion_feat_RBs):
            s.resblock_nobn(f"fusion_fea_block4.{i}", nf)
        s.conv("reduce_dim_conv", (5 if scale == 16 else 4) * nf, nf, 1)
        ik = "Indexer16" if scale == 16 else "Indexer8"
        _indexer(s, "refmodel.indexer", argref[ik], scale)
        _decoder(s, "refmodel.decoder", argref["Decoder"])
        cb = argref["Codebook"]
        s["refmodel.codebook.embedding.weight"] = ParamSpec((cb["num_codebook_vectors"], cb["latent_dim"]), "emb", False)
        _encoder(s, "refmodel.encoder", argref["Encoder"])
    if align_mode == "POD":
        s.conv("fea_L2_conv1", nf, nf, 3); s.conv("fea_L2_conv2", nf, nf, 3)
        s.conv("fea_L3_conv1", nf, nf, 3); s.conv("fea_L3_conv2", nf, nf, 3)
        p = "align_module"
        # state_dict order: a module's own buffers precede its children
        s[p + ".spynet.mean"] = ParamSpec((1, 3, 1, 1), "buf_mean", False, True)
        s[p + ".spynet.std"] = ParamSpec((1, 3, 1, 1), "buf_std", False, True)
        for lvl in range(6):
            for idx, (ci, co) in zip((0, 2, 4, 6, 8), ((8, 32), (32, 64), (64, 32), (32, 16), (16, 2))):
                s.conv(f"{p}.spynet.basic_module.{lvl}.basic_module.{idx}", ci, co, 7, False)
        s.conv(p + ".flowdsconv0_1", 2, 16, 3); s.conv(p + ".flowdsconv0_2", 2, 16, 3)
        for n in ("flowdsconv1_1", "flowdsconv1_2", "flowdsconv2_1", "flowdsconv2_2"):
            s.conv(f"{p}.{n}", 16, 16, 3)

        def dcn(name):
            s[name + ".weight"] = ParamSpec((nf, nf, 3, 3), "conv_w", True)
            s[name + ".bias"] = ParamSpec((nf,), "bias", True)
            s.conv(name + ".conv_offset", nf, groups * 27, 3)
        s.conv(p + ".L3_offset_conv1", 2 * nf + 34, nf, 3); s.conv(p + ".L3_offset_conv2", nf, nf, 3); dcn(p + ".L3_dcnpack")
        s.conv(p + ".L2_offset_conv1", 2 * nf + 34, nf, 3); s.conv(p + ".L2_offset_conv2", 2 * nf, nf, 3)
        s.conv(p + ".L2_offset_conv3", nf, nf, 3); dcn(p + ".L2_dcnpack"); s.conv(p + ".L2_fea_conv", 2 * nf, nf, 3)
        s.conv(p + ".L1_offset_conv1", 2 * nf + 34, nf, 3); s.conv(p + ".L1_offset_conv2", 2 * nf, nf, 3)
        s.conv(p + ".L1_offset_conv3", nf, nf, 3); dcn(p + ".L1_dcnpack"); s.conv(p + ".L1_fea_conv", 2 * nf, nf, 3)
        s.conv(p + ".cas_offset_conv1", 2 * nf, nf, 3); s.conv(p + ".cas_offset_conv2", nf, nf, 3); dcn(p + ".cas_dcnpack")
    if fusion_mode == "ThreeDA":
        p, t = "ThreeDA", nframes
        s.conv(p + ".temporal_attn1", nf, nf, 3); s.conv(p + ".temporal_attn2", nf, nf, 3)
        s.conv(p + ".feat_fusion", t * nf, nf, 1)
        for n in ("conv3D_1", "conv3D_2"):
            s[f"{p}.{n}.weight"] = ParamSpec((t, t, 1, 1, 1), "conv3d_w", True)
            s[f"{p}.{n}.bias"] = ParamSpec((t,), "bias", True)
        s.conv(p + ".conv3D_fusion_1", t * nf, nf, 1); s.conv(p + ".conv3D_fusion_2", t * nf, nf, 1)
        s.conv(p + ".conv2D_fusion_3", nf, nf, 1)
        s.conv(p + ".spatial_attn1", t * nf, nf, 1); s.conv(p + ".spatial_attn2", 2 * nf, nf, 1)
        s.conv(p + ".spatial_attn3", nf, nf, 3); s.conv(p + ".spatial_attn4", nf, nf, 1)
        s.conv(p + ".spatial_attn5", nf, nf, 3); s.conv(p + ".spatial_attn_l1", nf, nf, 1)
        s.conv(p + ".spatial_attn_l2", 2 * nf, nf, 3); s.conv(p + ".spatial_attn_l3", nf, nf, 3)
        s.conv(p + ".spatial_attn_add1", nf, nf, 1); s.conv(p + ".spatial_attn_add2", nf, nf, 1)
    for i in range(back_RBs):
        s.resblock_nobn(f"recon_trunk.{i}", nf)
    s.conv("upconv1", nf, nf * 4, 3); s.conv("upconv2", nf, 64 * 4, 3); s.conv("upconv3", 64, 64 * 4, 3)
    if mode == "16to1":
        s.conv("upconv4", 64, 64 * 4, 3)
    s.conv("HRconv", 64, 64, 3)
    s.conv("conv_last", 64, 1, 3)
    return s
