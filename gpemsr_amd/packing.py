"""Weight repacking: reference state-dict layouts -> the kernel-native layout of
gpemsr_conv2d ([tap][cout][cin_pad], cin fastest, cin padded per concat source to
the kernel's channel chunk: 8 for k>=3, 32 for 1x1).  Done once at load time.

Reference layouts (model/GPEMSR.py, model/blocks.py): Conv2d OIHW; ConvTranspose2d
[Cin,Cout,3,3]; Linear [out,in]; DCNv2Pack.weight [Cout,Cin,3,3]."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .ops import PackedConv


def _pad_split(w_tco: torch.Tensor, splits: Sequence[int], ck: int) -> torch.Tensor:
    """w_tco: [tap][cout][cin] -> cin padded per split to a multiple of ck."""
    pieces, off = [], 0
    for c in splits:
        piece = w_tco[:, :, off:off + c]
        pad = (-c) % ck
        if pad:
            piece = torch.cat([piece, piece.new_zeros(piece.shape[0], piece.shape[1], pad)], dim=2)
        pieces.append(piece)
        off += c
    assert off == w_tco.shape[2], (off, w_tco.shape)
    return torch.cat(pieces, dim=2).contiguous()


def pack_conv(w: torch.Tensor, b: Optional[torch.Tensor], device, splits: Optional[Sequence[int]] = None,
              pixel_shuffle: bool = False, scale: float = 1.0) -> PackedConv:
    cout, cin, kh, kw = w.shape
    assert kh == kw
    ck = 32 if kh == 1 else 8
    splits = tuple(splits) if splits is not None else (cin,)
    wt = w.detach().to(torch.float32).permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
    bb = None if b is None else b.detach().to(torch.float32).clone()
    if scale != 1.0:
        wt = wt * scale
        bb = None if bb is None else bb * scale
    if pixel_shuffle:
        # PixelShuffle(2): out[c', 2y+i, 2x+j] = in[4c'+2i+j, y, x]; regroup rows as (2i+j)*C/4 + c'
        cq = cout // 4
        perm = torch.tensor([4 * c + q for q in range(4) for c in range(cq)], dtype=torch.long)
        wt = wt[:, perm]
        bb = None if bb is None else bb[perm]
    packed = _pad_split(wt, splits, ck).to(device)
    return PackedConv(packed, None if bb is None else bb.contiguous().to(device), kh, cout, splits, ck,
                      transposed=False, pixel_shuffle=pixel_shuffle)


def pack_winograd(w: torch.Tensor, device, pixel_shuffle: bool = False) -> torch.Tensor:
    """Conv2d(3x3) weight OIHW fp32 -> U = G g G^T for the Winograd F(2x2, 3x3) form of gpemsr_conv2d (descriptor.transposed = 3;
    csrc/conv_wino.hip): [cin / 8][position p = 4 xi + nu][cout][8] fp32, computed in float64 (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]).
    Concat sources need no padding here: the form requires every source to be a multiple of 8 channels."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    g = w.detach().to(torch.float64).cpu()
    if pixel_shuffle:                                                    # the same row permutation as pack_conv: (2i + j) * C/4 + c
        cq = cout // 4
        g = g[torch.tensor([4 * c + q for q in range(4) for c in range(cq)], dtype=torch.long)]
    U = torch.einsum("xa,ocab,yb->xyoc", G, g, G)                      # [xi][nu][cout][cin]
    # staged order [cin / 8][position][cout][8]: the 64 (32) couts of a workgroup's block are 2 KB (1 KB) of CONSECUTIVE memory per position and
    # chunk, so every LDS-DMA instruction reads whole cache lines (round 4's [position][cout][cin] made each lane fetch an isolated 32-byte piece
    # of a line whose other 96 bytes belonged to later chunks)
    return U.reshape(16, cout, cin // 8, 8).permute(2, 0, 1, 3).to(torch.float32).contiguous().to(device)


# Winograd F(4x4, 3x3) with the points 0, +-1, +-2, infinity (Lavin & Gray's scaling: an integer B^T, the fractions in G) -- csrc/conv_wino4.hip
WINO4_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                        [0, 0, 1]], dtype=torch.float64)
WINO4_BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                         [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
WINO4_AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def pack_winograd4(w: torch.Tensor, device, pixel_shuffle: bool = False) -> torch.Tensor:
    """Conv2d(3x3) weight OIHW fp32 (cin % 8 == 0) -> U = G g G^T for the Winograd F(4x4, 3x3) form of gpemsr_conv2d (descriptor.transposed
    = 5; csrc/conv_wino4.hip): [cin / 8][position p = 6 xi + nu][quad][cout_pad][4] fp32, folded in float64; cout_pad = 64 ceil(cout / 64), zero rows
    behind cout (the kernel does not store them).  A wave's fragment of one position (32 couts x one quad) is 512 consecutive bytes: the kernel
    reads it straight into registers."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 8 == 0 and not (pixel_shuffle and cout % 256)
    g = w.detach().to(torch.float64).cpu()
    if cout % 64:
        g = torch.cat([g, torch.zeros(64 - cout % 64, cin, 3, 3, dtype=torch.float64)])
        cout = g.shape[0]
    if pixel_shuffle:                                                    # the same row permutation as pack_conv: (2i + j) * C/4 + c
        cq = cout // 4
        g = g[torch.tensor([4 * c + q for q in range(4) for c in range(cq)], dtype=torch.long)]
    U = torch.einsum("xa,ocab,yb->xyoc", WINO4_G, g, WINO4_G)                                     # [xi][nu][cout][cin]
    U = U.reshape(36, cout, cin // 8, 2, 4).permute(2, 0, 3, 1, 4)                                    # [chunk][p][quad][cout][4]
    return U.to(torch.float32).contiguous().to(device)


def pack_rowpair7(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d(cin -> 16, 7x7, stride 1, pad 3) fp32 weight -> the ROW-PAIR form of gpemsr_conv2d (descriptor.transposed = 2):
    out(2i, x) = sum_{ky'=0..6} in(2i-3+ky') W[ky'] and out(2i+1, x) = sum_{ky'=1..7} in(2i-3+ky') W[ky'-1] share the 8-row window of
    input rows 2i-3 .. 2i+4, so one 32-row matrix tile holds the 16 couts of both rows: [tap = ky'*7+kx][32][cin_pad]."""
    cout, cin, kh, kw = w.shape
    assert cout == 16 and kh == 7 and kw == 7
    wf = w.detach().to(torch.float32).cpu()
    w8 = torch.zeros(32, cin, 8, 7, dtype=torch.float32)
    w8[:16, :, 0:7] = wf
    w8[16:, :, 1:8] = wf
    wt = w8.permute(2, 3, 0, 1).reshape(56, 32, cin)
    return _pad_split(wt, (cin,), 8).to(device)


def pack_convT(w: torch.Tensor, b: torch.Tensor, device) -> PackedConv:
    """ConvTranspose2d(k3,s2,p1,op1) [Cin,Cout,3,3] -> the kernel's phase-stacked 2x2-tap form
    [tap = 2*dy+dx][n' = (co//32)*128 + q*32 + co%32][cin_pad], q = 2*py+px:
        out(2i+py, 2j+px) = sum_{dy<=py, dx<=px} in(i+dy, j+dx) . W[:, :, py+1-2dy, px+1-2dx]
    (rows of phases that a tap does not feed stay zero and are skipped by the kernel's tap mask)."""
    cin, cout, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cout % 32 == 0
    wf = w.detach().to(torch.float32).cpu()
    out = torch.zeros(4, 4 * cout, cin, dtype=torch.float32)
    co = torch.arange(cout)
    for dy in range(2):
        for dx in range(2):
            for py in range(dy, 2):
                for px in range(dx, 2):
                    q = 2 * py + px
                    rows = (co // 32) * 128 + q * 32 + (co % 32)
                    out[2 * dy + dx, rows] = wf[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()
    return PackedConv(_pad_split(out, (cin,), 8).to(device), b.detach().to(torch.float32).contiguous().to(device), 3, cout,
                      (cin,), 8, transposed=True)


def pack_linear(w: torch.Tensor, b: Optional[torch.Tensor], device) -> PackedConv:
    return pack_conv(w.reshape(w.shape[0], w.shape[1], 1, 1), b, device)


def pack_dcn(w: torch.Tensor, b: torch.Tensor, device) -> PackedConv:
    """DCN contraction as a 1x1 conv over the tap-major column tensor [9*cin]."""
    cout, cin, kh, kw = w.shape
    wt = w.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(1, cout, kh * kw * cin)
    return PackedConv(_pad_split(wt, (kh * kw * cin,), 32).to(device), b.detach().to(torch.float32).contiguous().to(device),
                      1, cout, (kh * kw * cin,), 32)


# 1-D Winograd F(2, 7), interpolation points 0, +-1, +-2, +-1/2, infinity:  [y0, y1] = A^T [ (G g) (.) (B^T d) ],  d = 8 inputs, g = 7 taps.
# G evaluates the filter polynomial at the points (last row: leading coefficient); B^T and A^T are hard-coded in csrc/conv7_wino.hip
# (B^T = C^-T of the 8 x 8 Vandermonde matrix, A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 1]); tests/test_host_cpu.py re-derives all
# three and checks the identity in float64.
WINO7_POINTS = (0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5)
WINO7_G = torch.tensor([[p ** k for k in range(7)] for p in WINO7_POINTS] + [[0.0] * 6 + [1.0]], dtype=torch.float64)
WINO7_AT = torch.tensor([[1.0] * 7 + [0.0], list(WINO7_POINTS) + [1.0]], dtype=torch.float64)


def wino7_bt() -> torch.Tensor:
    """B^T [8][8] of F(2, 7) for WINO7_POINTS: the transpose of the inverse of the evaluation matrix of degree-7 polynomials."""
    C = torch.tensor([[p ** k for k in range(8)] for p in WINO7_POINTS] + [[0.0] * 7 + [1.0]], dtype=torch.float64)
    return torch.linalg.inv(C).T.contiguous()


def pack_winograd7(w: torch.Tensor, device) -> torch.Tensor:
    """Weights of a 7x7 stride-1 convolution [cout][cin][7][7] (cin % 8 == 0) in the F(2, 7) row form of gpemsr_conv2d (descriptor.transposed
    = 4): U[cin/8][ky][nu][quad][cout][4] fp32 with U[nu] = sum_kx G[nu][kx] w[ky][kx], folded in float64."""
    cout, cin, kh, kw = w.shape
    assert kh == 7 and kw == 7 and cin % 8 == 0
    u = torch.einsum("pk,ocyk->ocyp", WINO7_G, w.detach().to(torch.float64).cpu())           # [cout][cin][ky][nu]
    u = u.reshape(cout, cin // 8, 2, 4, 7, 8).permute(1, 4, 5, 2, 0, 3)                          # [chunk][ky][nu][quad][cout][4]
    return u.to(torch.float32).contiguous().to(device)


def pack_winograd77(w: torch.Tensor, device) -> torch.Tensor:
    """Weights of a 7x7 stride-1 convolution [cout][cin][7][7] (cin % 8 == 0) in the 2-D F(2x2, 7x7) form of gpemsr_conv2d (descriptor.transposed
    = 6): U[cin/8][64 = 8 xi + nu][quad][cout][4] fp32 with U[xi][nu] = sum_(ky, kx) G[xi][ky] w[ky][kx] G[nu][kx], folded in float64."""
    cout, cin, kh, kw = w.shape
    assert kh == 7 and kw == 7 and cin % 8 == 0
    u = torch.einsum("ay,ocyx,bx->ocab", WINO7_G, w.detach().to(torch.float64).cpu(), WINO7_G)   # [cout][cin][xi][nu]
    u = u.reshape(cout, cin // 8, 2, 4, 64).permute(1, 4, 2, 0, 3)                                # [chunk][p][quad][cout][4]
    return u.to(torch.float32).contiguous().to(device)


def pack_dcn_rows_bf16(w: torch.Tensor, device) -> torch.Tensor:
    """DCN weight [cout][cin][3][3] as plain bf16 rows [cout][tap][cin] -- the K order of the column rows gpemsr_dcn_conv_bf16 builds in LDS
    (tap-major, channel fastest: the same order as the stand-alone column tensor of gpemsr_dcn_columns_bf16)."""
    cout, cin, kh, kw = w.shape
    return w.detach().to(torch.float32).permute(0, 2, 3, 1).reshape(cout, kh * kw * cin).to(torch.bfloat16).contiguous().to(device)


def pack_vgg_first(w: torch.Tensor, b: torch.Tensor, device) -> PackedConv:
    """vgg conv1_1 applied to a 1-channel image expanded to 3 identical channels
    (model/GPEMSR.py:386,390) == a 1->64 conv with the weights summed over Cin."""
    return pack_conv(w.detach().to(torch.float32).sum(dim=1, keepdim=True), b, device)


def pack_conv_split(pc: PackedConv, w: torch.Tensor, device, pixel_shuffle: bool = False) -> torch.Tensor:
    """Split-bf16 weights for gpemsr_conv2d_split: [plane (hi, lo)][cin/16][tap][k-half][cout][8] bf16 (see _stage_order)
    with hi = bf16(w) (round to nearest even) and lo = bf16(w - hi).  Same tap / row order as the fp32 packing of ``pc``
    (incl. the PixelShuffle row permutation); no channel padding (every source must have c % 16 == 0)."""
    cout, cin, kh, kw = w.shape
    assert kh == kw and kh in (1, 3, 7) and all(c % (32 if kh == 1 else 16) == 0 for c in pc.splits)
    wt = w.detach().to(torch.float32).cpu().permute(2, 3, 0, 1).reshape(kh * kw, cout, cin)
    if pixel_shuffle:
        cq = cout // 4
        perm = torch.tensor([4 * c + q for q in range(4) for c in range(cq)], dtype=torch.long)
        wt = wt[:, perm]
    return _split_planes(wt, device)


def _stage_order(w3: torch.Tensor) -> torch.Tensor:
    """[tap][cout][cin] -> [cin/16][tap][k-half][cout][8]: the order in which the split kernel stages weights, so that one
    LDS-DMA instruction (64 lanes x 16 B, lane-linear in LDS) reads 1 KB of consecutive global memory: for a 16-channel
    chunk, a tap and a half of the chunk (8 channels = one MFMA operand register quad), the couts are contiguous."""
    t, cout, cin = w3.shape
    assert cin % 16 == 0
    return w3.reshape(t, cout, cin // 16, 2, 8).permute(2, 0, 3, 1, 4).contiguous()


def _split_planes(wt: torch.Tensor, device) -> torch.Tensor:
    hi = wt.to(torch.bfloat16)
    lo = (wt - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.stack([_stage_order(hi), _stage_order(lo)], dim=0).contiguous().to(device)


def pack_convT_split(pc: PackedConv, device) -> torch.Tensor:
    """Split-bf16 planes of the phase-stacked transposed-conv weights already packed in ``pc.w`` ([4 taps][4*Cout][cin])."""
    wt = pc.w.detach().to(torch.float32).cpu()
    assert pc.transposed and wt.shape[2] % 16 == 0
    return _split_planes(wt, device)


# ----------------------------------------------------------------------------------------------------------------------
# bf16 data path (gpemsr_conv2d_bf16): weights in the kernel's staged order [cin/CK][tap][CK/8][cout][8] bf16
# ----------------------------------------------------------------------------------------------------------------------
def bf16_chunk(splits: Sequence[int]) -> int:
    """Channel chunk of a launch: 32 when every source is a multiple of 32 channels, else 16."""
    assert all(c % 16 == 0 for c in splits), f"bf16 convolutions need source channels % 16 == 0, got {splits}"
    return 32 if all(c % 32 == 0 for c in splits) else 16


def _stage_order_bf16(w3: torch.Tensor, ck: int) -> torch.Tensor:
    """[tap][cout][cin] -> [cin/ck][tap][ck/8][cout][8]: for a chunk, a tap and an 8-channel piece the couts are contiguous, so
    one LDS-DMA instruction reads 1 KiB of consecutive memory and lands as the [tap][piece][cout][8] image the MFMA B
    fragments are read from."""
    t, cout, cin = w3.shape
    assert cin % ck == 0
    return w3.reshape(t, cout, cin // ck, ck // 8, 8).permute(2, 0, 3, 1, 4).contiguous()


def pack_conv_bf16(w: torch.Tensor, device, splits: Optional[Sequence[int]] = None, pixel_shuffle: bool = False,
                   scale: float = 1.0) -> torch.Tensor:
    """OIHW conv weight -> staged bf16 (same tap / row order as pack_conv, incl. the PixelShuffle row permutation)."""
    cout, cin, kh, kw = w.shape
    assert kh == kw and kh in (1, 3, 7)
    splits = tuple(splits) if splits is not None else (cin,)
    assert sum(splits) == cin
    wt = w.detach().to(torch.float32).cpu().permute(2, 3, 0, 1).reshape(kh * kw, cout, cin) * scale
    if pixel_shuffle:
        cq = cout // 4
        perm = torch.tensor([4 * c + q for q in range(4) for c in range(cq)], dtype=torch.long)
        wt = wt[:, perm]
    return _stage_order_bf16(wt, bf16_chunk(splits)).to(torch.bfloat16).to(device)


def pack_linear_bf16x3(w: torch.Tensor, b: Optional[torch.Tensor], device) -> PackedConv:
    """nn.Linear [out, in] fp32 -> the weight of its three-bf16-product form: with W = Whi + Wlo (+ 2^-17) and the activation split
    the same way (ops.split_hi_lo_bf16), x W^T = hi Whi^T + lo Whi^T + hi Wlo^T to ~2^-16 relative -- a 1x1 bf16 convolution over the
    sources [hi, lo, hi] with the input-channel blocks [Whi | Whi | Wlo] and fp32 accumulation (the lo*Wlo term, 2^-18, is dropped)."""
    wf = w.detach().to(torch.float32).cpu()
    whi = wf.to(torch.bfloat16).to(torch.float32)
    wlo = (wf - whi).to(torch.bfloat16).to(torch.float32)
    cout, cin = wf.shape
    w3 = torch.cat([whi, whi, wlo], dim=1).reshape(cout, 3 * cin, 1, 1)
    pc = PackedConv(torch.empty(0, device=device), None if b is None else b.detach().to(torch.float32).contiguous().to(device),
                    1, cout, (cin, cin, cin), 32)
    pc.wb = pack_conv_bf16(w3, device, (cin, cin, cin))
    pc.algo_cin = cin
    return pc


def pack_convT_bf16(w: torch.Tensor, device) -> torch.Tensor:
    """ConvTranspose2d(k3,s2,p1,op1) [Cin,Cout,3,3] -> phase-stacked 2x2-tap rows (pack_convT) in staged bf16 order."""
    cin, cout, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cout % 32 == 0 and cin % 16 == 0
    wf = w.detach().to(torch.float32).cpu()
    out = torch.zeros(4, 4 * cout, cin, dtype=torch.float32)
    co = torch.arange(cout)
    for dy in range(2):
        for dx in range(2):
            for py in range(dy, 2):
                for px in range(dx, 2):
                    rows = (co // 32) * 128 + (2 * py + px) * 32 + (co % 32)
                    out[2 * dy + dx, rows] = wf[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()
    staged = _stage_order_bf16(out, bf16_chunk((cin,))).to(torch.bfloat16)
    if cin % 32 == 0 and not convT_resident_form_ok(cin, cout):
        # behind the staged form: the COMPACT form of the loader-wave kernel's whole-chunk stages (weight_forms bit 1): only the nine
        # non-zero (tap, phase) blocks, [cin/32][tap][piece][cout/32][rows_t][8] with the phases a tap feeds in increasing order
        staged = torch.cat([staged.reshape(-1), pack_convT_compact_bf16(wf).reshape(-1)])
    if convT_resident_form_ok(cin, cout):
        # behind the staged form: the nine non-zero (tap, phase) blocks per 64-cout slab for the weights-resident kernel
        # (gpemsr_conv16_desc.weight_forms bit 0; csrc/conv_bf16.hip::convt64_resident_kernel)
        staged = torch.cat([staged.reshape(-1), pack_convT_resident_bf16(wf).reshape(-1)])
    return staged.to(device)


def convT_resident_form_ok(cin: int, cout: int) -> bool:
    return cin == 64 and cout % 64 == 0


# (phase q = 2 py + px, tap (dy, dx)) of the nine non-zero blocks, in the order the kernel walks them
CONVT_BLOCKS = ((0, 0, 0), (1, 0, 0), (1, 0, 1), (2, 0, 0), (2, 1, 0), (3, 0, 0), (3, 0, 1), (3, 1, 0), (3, 1, 1))


def pack_convT_compact_bf16(wf: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d [Cin % 32 == 0, Cout % 32 == 0, 3, 3] fp32 -> flat bf16 [cin/32][tap = 2 dy + dx][piece 4][cout/32][rows_t][8]:
    rows_t = (phases fed by the tap, increasing q = 2 py + px) x 32 couts; row (q-slot, co % 32) of block co // 32 holds
    W[:, co, py+1-2dy, px+1-2dx] for the chunk's 32 input channels in four 8-channel pieces."""
    cin, cout = wf.shape[0], wf.shape[1]
    assert cin % 32 == 0 and cout % 32 == 0
    nchunk, nblk = cin // 32, cout // 32
    parts = []
    for dy in range(2):
        for dx in range(2):
            qs = [2 * py + px for py in range(dy, 2) for px in range(dx, 2)]
            t = torch.zeros(nchunk, 4, nblk, len(qs), 32, 8, dtype=torch.float32)           # [chunk][piece][blk][q-slot][co % 32][8]
            for si, q in enumerate(qs):
                py, px = q >> 1, q & 1
                m = wf[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()                          # [cout][cin]
                t[:, :, :, si] = m.reshape(nblk, 32, nchunk, 4, 8).permute(2, 3, 0, 1, 4)
            parts.append(t.reshape(nchunk, -1))
    return torch.cat(parts, dim=1).contiguous().to(torch.bfloat16)                            # per chunk: tap 0 | tap 1 | tap 2 | tap 3


def pack_convT_resident_bf16(wf: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d [Cin = 64, Cout, 3, 3] fp32 -> [Cout/64][chunk 2][block 9][piece 4][64 couts][8] bf16: block (q, dy, dx) holds
    W[:, :, py+1-2dy, px+1-2dx]^T (rows = couts of the slab, columns = the chunk's 32 input channels in four 8-channel pieces)."""
    cin, cout = wf.shape[0], wf.shape[1]
    assert convT_resident_form_ok(cin, cout)
    out = torch.zeros(cout // 64, 2, 9, 4, 64, 8, dtype=torch.float32)
    for b, (q, dy, dx) in enumerate(CONVT_BLOCKS):
        py, px = q >> 1, q & 1
        m = wf[:, :, py + 1 - 2 * dy, px + 1 - 2 * dx].t()                   # [cout][cin]
        out[:, :, b] = m.reshape(cout // 64, 64, 2, 4, 8).permute(0, 2, 3, 1, 4)
    return out.to(torch.bfloat16)


def _tap_fragments(rows: torch.Tensor) -> torch.Tensor:
    """rows [R <= 32][64] (bf16-representable fp32) -> the A-operand fragments [4 k-steps][64 lanes][8] of one 32-row MFMA tile:
    lane l holds row l % 32, channels 16 ks + 8 (l // 32) .. + 8 (csrc/tap_sum.hip)."""
    full = torch.zeros(32, 64, dtype=torch.float32)
    full[:rows.shape[0]] = rows
    return full.reshape(32, 4, 2, 8).permute(1, 2, 0, 3).reshape(4, 64, 8).to(torch.bfloat16)


def _hi_lo(w: torch.Tensor):
    hi = w.to(torch.bfloat16).to(torch.float32)
    return hi, (w - hi).to(torch.bfloat16).to(torch.float32)


def pack_cout1_taps(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d(64 -> 1, 3x3) weight [1][64][3][3] -> tap fragments for gpemsr_conv_c64_cout1_bf16: rows 0..8 carry the bf16 hi
    halves of the nine taps, rows 16..24 their lo halves (one MFMA, summed in the epilogue)."""
    assert tuple(w.shape) == (1, 64, 3, 3)
    rows = w.detach().to(torch.float32).cpu()[0].permute(1, 2, 0).reshape(9, 64)
    hi, lo = _hi_lo(rows)
    full = torch.zeros(32, 64)
    full[0:9], full[16:25] = hi, lo
    return _tap_fragments(full).contiguous().to(device)


def compose_upconv_out(w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor]):
    """ConvTranspose2d(C -> M, k3 s2 p1 op1; w1 [C][M][3][3]) followed by Conv2d(M -> 1, 3x3 pad 1; w2 [1][M][3][3]) as one
    operator (csrc/tap_sum.hip): returns fp64 (taps [25][C] with t = ty*5+tx and o = 2 i + t - 2, S [9], b2, Wy0 [5][C], Wx0 [5][C],
    Wc [C])."""
    w1d, w2d = w1.detach().double().cpu(), w2.detach().double().cpu()[0]
    C = w1d.shape[0]
    m = torch.einsum("cmab,mde->cabde", w1d, w2d)              # [C][ky][kx][dy][dx]
    taps = torch.zeros(5, 5, C, dtype=torch.float64)
    wy0 = torch.zeros(5, C, dtype=torch.float64)
    wx0 = torch.zeros(5, C, dtype=torch.float64)
    for ky in range(3):
        for dy in range(3):
            for kx in range(3):
                for dx in range(3):
                    ty, tx = ky - dy + 2, kx - dx + 2
                    taps[ty, tx] += m[:, ky, kx, dy, dx]
                    if ky == 0 and dy == 0:
                        wy0[tx] += m[:, ky, kx, dy, dx]
                    if kx == 0 and dx == 0:
                        wx0[ty] += m[:, ky, kx, dy, dx]
    wc = m[:, 0, 0, 0, 0].clone()
    s = torch.einsum("m,mde->de", b1.detach().double().cpu(), w2d).reshape(9) if b1 is not None else torch.zeros(9, dtype=torch.float64)
    bb = float(b2.detach().double().cpu()[0]) if b2 is not None else 0.0
    return taps.reshape(25, C), s, bb, wy0, wx0, wc


def pack_upconv_out(w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor], device):
    """-> (fragments [2 sets: hi, lo][4][64][8] bf16, consts [10 + 5*64 + 5*64 + 64] fp32) for gpemsr_upconv_out_c64_bf16."""
    assert tuple(w1.shape) == (64, 64, 3, 3) and tuple(w2.shape) == (1, 64, 3, 3)
    taps, s, bb, wy0, wx0, wc = compose_upconv_out(w1, b1, w2, b2)
    hi, lo = _hi_lo(taps.to(torch.float32))
    frag = torch.stack([_tap_fragments(hi), _tap_fragments(lo)]).contiguous().to(device)
    consts = torch.cat([s, torch.tensor([bb], dtype=torch.float64), wy0.reshape(-1), wx0.reshape(-1), wc]).to(torch.float32).contiguous().to(device)
    return frag, consts


def pack_rowsum7(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d(16 -> 2, 7x7) weight [2][16][7][7] -> per-kx A-operand fragments [7][64 lanes][8] bf16 for gpemsr_conv7_c16_cout2_bf16:
    MFMA row 2 ky + co carries W[co][:, ky, kx] (bf16 hi half), row 16 + 2 ky + co its lo half; lane l holds row l % 32, channels
    8 (l // 32) .. + 8."""
    assert tuple(w.shape) == (2, 16, 7, 7)
    wf = w.detach().to(torch.float32).cpu()
    frags = []
    for kx in range(7):
        rows = wf[:, :, :, kx].permute(2, 0, 1).reshape(14, 16)                 # [(ky, co)][c]
        hi, lo = _hi_lo(rows)
        full = torch.zeros(32, 16)
        full[0:14], full[16:30] = hi, lo
        frags.append(full.reshape(32, 2, 8).permute(1, 0, 2).reshape(64, 8))      # lane = half * 32 + row
    return torch.stack(frags).to(torch.bfloat16).contiguous().to(device)


def pack_conv7_c32_cout16(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d(32 -> 16, 7x7) weight [16][32][7][7] -> the A-operand fragments of v_mfma_f32_16x16x32_bf16 for
    gpemsr_conv7_c32_cout16_bf16: [tap = 7 ky + kx][k-group 4][16 couts][8] bf16 -- lane l of the fragment of a tap holds cout l % 16,
    input channels 8 (l // 16) .. + 7."""
    assert tuple(w.shape) == (16, 32, 7, 7)
    wf = w.detach().to(torch.float32).cpu()
    return wf.permute(2, 3, 0, 1).reshape(49, 16, 4, 8).permute(0, 2, 1, 3).contiguous().to(torch.bfloat16).to(device)


def pack_conv7_c8_cout32(w: torch.Tensor, device) -> torch.Tensor:
    """Conv2d(8 -> 32, 7x7) weight [32][8][7][7] -> A-operand fragments for gpemsr_conv7_c8_cout32_bf16: [tap group j 13][cout tile m 2]
    [64 lanes][8] bf16 -- lane l holds cout 16 m + l % 16 and the 8 input channels of tap 4 j + l // 16 (taps 49..51: zeros)."""
    assert tuple(w.shape) == (32, 8, 7, 7)
    wf = w.detach().to(torch.float32).cpu().reshape(32, 8, 49)
    wt = torch.zeros(32, 8, 52)
    wt[:, :, :49] = wf
    # [cout = m*16 + l16][c][tap = j*4 + g] -> [j][m][g][l16][c]
    return wt.reshape(2, 16, 8, 13, 4).permute(3, 0, 4, 1, 2).contiguous().reshape(13, 2, 64, 8).to(torch.bfloat16).to(device)


def _tap_fragments_f32(rows: torch.Tensor) -> torch.Tensor:
    """rows [R <= 32][64] fp32 -> A-operand fragments [32 k-steps][64 lanes] of v_mfma_f32_32x32x2_f32: lane l holds row l % 32,
    channel 2 ks + l // 32."""
    full = torch.zeros(32, 64, dtype=torch.float32)
    full[:rows.shape[0]] = rows
    return full.reshape(32, 32, 2).permute(1, 2, 0).reshape(32, 64).contiguous()          # [ks][half][row] -> [ks][lane]


def pack_cout1_taps_f32(w: torch.Tensor, device) -> torch.Tensor:
    assert tuple(w.shape) == (1, 64, 3, 3)
    rows = w.detach().to(torch.float32).cpu()[0].permute(1, 2, 0).reshape(9, 64)
    return _tap_fragments_f32(rows).to(device)


def pack_upconv_out_f32(w1: torch.Tensor, b1: Optional[torch.Tensor], w2: torch.Tensor, b2: Optional[torch.Tensor], device):
    """-> (fragments [32][64] fp32, consts [10 + 5*64 + 5*64 + 64] fp32) for gpemsr_upconv_out_c64_f32 (composition in fp64)."""
    assert tuple(w1.shape) == (64, 64, 3, 3) and tuple(w2.shape) == (1, 64, 3, 3)
    taps, s, bb, wy0, wx0, wc = compose_upconv_out(w1, b1, w2, b2)
    frag = _tap_fragments_f32(taps.to(torch.float32)).to(device)
    consts = torch.cat([s, torch.tensor([bb], dtype=torch.float64), wy0.reshape(-1), wx0.reshape(-1), wc]).to(torch.float32).contiguous().to(device)
    return frag, consts


def pack_rowsum7_f32(w: torch.Tensor, device) -> torch.Tensor:
    """pack_rowsum7 for fp32 activations: [7 kx][8 k-steps][64 lanes] floats -- lane l holds row l % 32 (= 2 ky + co), channel 2 ks + l // 32."""
    assert tuple(w.shape) == (2, 16, 7, 7)
    wf = w.detach().to(torch.float32).cpu()
    frags = []
    for kx in range(7):
        rows = torch.zeros(32, 16)
        rows[0:14] = wf[:, :, :, kx].permute(2, 0, 1).reshape(14, 16)
        frags.append(rows.reshape(32, 8, 2).permute(1, 2, 0).reshape(8, 64))                 # [ks][half][row] -> [ks][lane]
    return torch.stack(frags).contiguous().to(device)
