"""Deterministic synthetic weights in the reference's state-dict layout.

The reference ships no checkpoints (Google-Drive only, /root/reference/README.md:29-46),
so benchmarks, smoke tests and parity tests use random-init weights of the exact
architecture.  Every tensor is a pure function of (key name, shape, seed): a
numpy PCG64 stream keyed by crc32(name), so the GPU box regenerates bit-identical
weights without shipping 358 MB.  Scales are chosen so activations stay O(1)
through ~100 layers (He-style fan-in scaling, damped residual branches).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict

import numpy as np
import torch

from .arch import ParamSpec, param_specs

_DAMPED = ("feature_extraction.", "recon_trunk.", "fusion_fea_block")


def _gain(name: str, spec: ParamSpec) -> float:
    if any(t in name for t in _DAMPED):
        return 1.4 if ".conv1." in name else 0.25           # x + conv2(relu(conv1 x)): keep branch small
    if "spynet" in name:
        return 0.25 if name.endswith("basic_module.8.weight") else 1.4
    if name.endswith("conv_offset.weight"):
        return 0.7
    if name.startswith("conv_last") or name.startswith("refmodel.decoder.output_layer"):
        return 0.25
    if ".block.0." in name or ".block.3." in name:          # conv feeding a GroupNorm: scale-free
        return 1.0
    return 1.0


def synth_tensor(name: str, spec: ParamSpec, seed: int = 0) -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))
    shape, kind = spec.shape, spec.kind
    if kind == "buf_mean":
        return torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(shape)
    if kind == "buf_std":
        return torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(shape)
    z = rng.standard_normal(size=shape, dtype=np.float32)
    if kind == "conv_w":
        fan_in = shape[1] * shape[2] * shape[3]
        z *= _gain(name, spec) / np.sqrt(fan_in)
    elif kind == "convT_w":                                  # each output pixel sees 9/4 taps on average
        z *= 1.0 / np.sqrt(shape[0] * 9 / 4.0)
    elif kind == "linear_w":
        z *= 1.0 / np.sqrt(shape[1])
    elif kind == "conv3d_w":
        z *= 1.0 / np.sqrt(shape[1])
    elif kind == "emb":
        z *= 0.5
    elif kind == "gn_w":
        z = 1.0 + 0.1 * z
    elif kind == "gn_b":
        z *= 0.1
    elif kind == "bias":
        z *= 0.05
        if name == "refmodel.decoder.output_layer.bias":
            z += 0.5
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))


def synth_state_dict(specs: "OrderedDict[str, ParamSpec]", seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, synth_tensor(k, v, seed)) for k, v in specs.items())


def synth_for_network(network_opt: dict, scale: int, seed: int = 0):
    """Convenience: state dict for the ``network`` block of an option YAML."""
    kw = {k: v for k, v in network_opt.items() if k not in ("ref_path_G", "ref_path_Indexer")}
    return synth_state_dict(param_specs(scale=scale, **kw), seed)


def synth_lr_tiles(batch: int, nframes: int, h: int, w: int, seed: int = 0, kind: str = "uniform") -> torch.Tensor:
    """Synthetic LR input ``[B,N,1,H,W]`` float32 in [0,1] (the range of
    data/util.py:75-88 ``read_img``).  ``uniform`` = U[0,1) noise;
    ``smooth`` = low-pass-filtered noise with slow drift across the N slices,
    closer to EM imagery so flow/deformable offsets are representative."""
    rng = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, 0x5EED, batch, nframes, h, w]))
    if kind == "uniform":
        x = rng.random(size=(batch, nframes, 1, h, w), dtype=np.float32)
    elif kind == "smooth":
        base = rng.random(size=(batch, 1, 1, h + 8, w + 8), dtype=np.float32)
        drift = rng.random(size=(batch, nframes, 1, h + 8, w + 8), dtype=np.float32)
        z = 0.7 * base + 0.3 * drift
        for _ in range(3):                                   # separable 3-tap box blur, 3 passes
            z = (z[..., :-2, :] + z[..., 1:-1, :] + z[..., 2:, :]) / 3.0
            z = (z[..., :, :-2] + z[..., :, 1:-1] + z[..., :, 2:]) / 3.0
        z = z[..., 1:1 + h, 1:1 + w]
        z = (z - z.min()) / max(float(z.max() - z.min()), 1e-6)
        x = (0.85 * z + 0.15 * rng.random(size=z.shape, dtype=np.float32)).astype(np.float32)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.ascontiguousarray(x))
