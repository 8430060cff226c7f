"""Option-file handling (the reference's YAML contract, util/util.py:23-56 and
output_GPEMSR.py:19-43): ordered YAML load, NoneDict semantics, model construction."""
from __future__ import annotations

from collections import OrderedDict

import yaml


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def load_options(path: str) -> dict:
    class _Loader(yaml.SafeLoader):
        pass
    _Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                            lambda loader, node: OrderedDict(loader.construct_pairs(node)))
    with open(path, mode="r", encoding="utf-8") as f:
        return yaml.load(f, Loader=_Loader)


def build_model(opt: dict, load_prior_files: bool = True, **extra):
    """GPEMSR(...) exactly as output_GPEMSR.py:36-43 constructs it."""
    from .model import GPEMSR
    net = opt["network"]
    # additive option key (not in the reference's YAMLs): `precision: fp32 | bf16x3 | bf16` under `network:` or at the top level
    prec = net.get("precision") or opt.get("precision")
    if prec and "precision" not in extra:
        extra["precision"] = str(prec)
    # additive: `indexer_precision: bf16 | bf16x3:N | fp32:N` (bf16 path: the indexer's last N units at a higher precision)
    ip = net.get("indexer_precision") or opt.get("indexer_precision")
    if ip and "indexer_precision" not in extra:
        extra["indexer_precision"] = str(ip)
    # additive: `winograd: f4x4 | decoder_f4x4 | f2x2 | off` (fp32 path: which 3x3 stride-1 layers run in a Winograd form; default f4x4)
    wg = net.get("winograd") if net.get("winograd") is not None else opt.get("winograd")
    if wg is False:                     # YAML reads a bare `off` as a boolean
        wg = "off"
    if wg and "winograd" not in extra:
        extra["winograd"] = str(wg)
    return GPEMSR(ref_path_G=net["ref_path_G"] if load_prior_files else None,
                  ref_path_Indexer=net["ref_path_Indexer"] if load_prior_files else None,
                  argref=net["argref"], nf=net["nf"], nframes=net["nframes"], groups=net["groups"],
                  front_RBs=net["front_RBs"], back_RBs=net["back_RBs"], w_ref=net["w_ref"],
                  ref_fusion_feat_RBs=net["ref_fusion_feat_RBs"], align_mode=net["align_mode"],
                  fusion_mode=net["fusion_mode"], mode=net["mode"], scale=opt["scale"], **extra)
