"""Shim of torchvision.ops.deform_conv2d (modulated, k3): sampling is done with
F.grid_sample(zeros padding, align_corners=True), which has the same
out-of-image rule as torchvision's bilinear_interpolate.  Deliberately a
different formulation from oracle/gpemsr_oracle.py::deform_conv2d_v2 so the two
restatements cross-check each other."""
import torch
from torch.nn import functional as F


def deform_conv2d(input, offset, weight, bias=None, stride=1, padding=1, dilation=1, mask=None):
    B, C, H, W = input.shape
    Co, _, kh, kw = weight.shape
    K = kh * kw
    G = offset.shape[1] // (2 * K)
    cg = C // G
    pad = padding if isinstance(padding, int) else padding[0]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=input.dtype), torch.arange(W, dtype=input.dtype), indexing='ij')
    cols = []
    for g in range(G):
        xg = input[:, g * cg:(g + 1) * cg]
        per_k = []
        for k in range(K):
            ky, kx = k // kw, k % kw
            py = ys + (ky - pad) + offset[:, g * 2 * K + 2 * k]
            px = xs + (kx - pad) + offset[:, g * 2 * K + 2 * k + 1]
            grid = torch.stack((2.0 * px / max(W - 1, 1) - 1.0, 2.0 * py / max(H - 1, 1) - 1.0), dim=3)
            v = F.grid_sample(xg, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
            if mask is not None:
                v = v * mask[:, g * K + k].unsqueeze(1)
            per_k.append(v)
        cols.append(torch.stack(per_k, dim=2))            # B,cg,K,H,W
    cols = torch.cat(cols, dim=1)                         # B,C,K,H,W
    out = torch.einsum('bckhw,ock->bohw', cols, weight.reshape(Co, C, K))
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out.contiguous()          # torchvision returns a dense NCHW tensor (callers .view() it, model/GPEMSR.py:176)
