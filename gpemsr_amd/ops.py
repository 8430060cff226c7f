"""Python-side operator wrappers over the C ABI (include/gpemsr_hip.h).

Activations are ``Act`` objects: float32 NHWC device memory owned by a torch
tensor (torch is used only as allocator / stream provider), possibly a channel
slice of a wider buffer (``ld`` > ``c``), which is how ``torch.cat(dim=1)`` in
the reference becomes free here.  Every function enqueues HIP kernels on the
current torch stream and returns immediately.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _abi
from ._abi import ACT_LRELU, ACT_LRELU_SIGMOID, ACT_NONE, ACT_RELU, ACT_SIGMOID  # noqa: F401


class Act:
    """NHWC float32 activation view: channels [off, off+c) of ``buf`` [n,h,w,ld].

    Training (gpemsr_amd/train.py): every view remembers the ``root`` allocation it was cut from; the root owns one
    zero-initialised gradient buffer of the same size (``gbuf``, created on first use) and a ``rg`` flag ("some
    producer of this memory depends on a trainable parameter"), so ``a.grad()`` is the same view over the gradient."""
    __slots__ = ("buf", "n", "h", "w", "c", "ld", "off", "_root", "gbuf", "rg", "gn")

    def __init__(self, buf: torch.Tensor, n: int, h: int, w: int, c: int, ld: int, off: int = 0, root: "Act" = None):
        self.buf, self.n, self.h, self.w, self.c, self.ld, self.off = buf, n, h, w, c, ld, off
        self._root = root            # None for a root allocation (no self reference: buffers must die by ref-count, not by gc)
        self.gbuf, self.rg = None, False
        self.gn = None               # bf16 path: (workspace, parts) GroupNorm partial sums written by the producing convolution

    @property
    def bf16(self) -> bool:
        """bf16 NHWC tensor of the precision="bf16" data path (else float32)."""
        return self.buf.dtype == torch.bfloat16

    @property
    def esize(self) -> int:
        return 2 if self.buf.dtype == torch.bfloat16 else 4

    @property
    def root(self) -> "Act":
        return self if self._root is None else self._root

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr() + self.esize * self.off

    @property
    def pixels(self) -> int:
        return self.n * self.h * self.w

    @property
    def requires_grad(self) -> bool:
        return self.root.rg

    def mark_grad(self):
        self.root.rg = True
        return self

    def grad(self) -> "Act":
        """The gradient of this view (same geometry) inside the root's zero-initialised gradient buffer."""
        r = self.root
        if r.gbuf is None:
            r.gbuf = torch.zeros(r.buf.numel(), dtype=torch.float32, device=r.buf.device)
        delta = (self.buf.data_ptr() - r.buf.data_ptr()) // 4
        assert 0 <= delta and delta + self.buf.numel() <= r.gbuf.numel()
        return Act(r.gbuf[delta:delta + self.buf.numel()], self.n, self.h, self.w, self.c, self.ld, self.off)

    def slice(self, c0: int, c: int) -> "Act":
        assert 0 <= c0 and c0 + c <= self.c
        return Act(self.buf, self.n, self.h, self.w, c, self.ld, self.off + c0, self.root)

    def images(self, i0: int, cnt: int) -> "Act":
        """Sub-range of images [i0, i0+cnt) (shares memory)."""
        assert 0 <= i0 and i0 + cnt <= self.n
        flat = self.buf.view(-1)
        start = i0 * self.h * self.w * self.ld
        return Act(flat[start:start + cnt * self.h * self.w * self.ld], cnt, self.h, self.w, self.c, self.ld, self.off, self.root)

    def reshape_hw(self, h: int, w: int) -> "Act":
        assert h * w == self.h * self.w
        return Act(self.buf, self.n, h, w, self.c, self.ld, self.off, self.root)

    def regroup(self, n: int) -> "Act":
        """View [n0,h,w,..] as n images of (n0/n)*h rows (pixel order unchanged)."""
        assert self.n % n == 0
        return Act(self.buf, n, (self.n // n) * self.h, self.w, self.c, self.ld, self.off, self.root)

    def torch(self) -> torch.Tensor:
        """[n,h,w,c] torch view (for tests / boundary)."""
        assert self.off + self.c <= self.ld
        v = self.buf.view(-1)[: self.n * self.h * self.w * self.ld].view(self.n, self.h, self.w, self.ld)
        return v[..., self.off:self.off + self.c]

    def nchw(self) -> torch.Tensor:
        """float32 NCHW copy (tests / traces / module boundary), whatever the storage format."""
        return self.torch().permute(0, 3, 1, 2).contiguous().to(torch.float32)


def new_act(n: int, h: int, w: int, c: int, ld: Optional[int] = None, device=None, bf16: bool = False, zero: bool = False) -> Act:
    ld = c if ld is None else ld
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    alloc = torch.zeros if zero else torch.empty
    return Act(alloc(n * h * w * ld, dtype=torch.bfloat16 if bf16 else torch.float32, device=dev), n, h, w, c, ld, 0)


def from_nhwc(t: torch.Tensor) -> Act:
    assert t.dtype in (torch.float32, torch.bfloat16) and t.dim() == 4 and t.is_contiguous()
    n, h, w, c = t.shape
    return Act(t, n, h, w, c, c, 0)


def from_nchw(t: torch.Tensor) -> Act:
    """NCHW -> NHWC (copy unless C == 1)."""
    assert t.dim() == 4
    if t.shape[1] == 1:
        t = t.contiguous()
        return Act(t, t.shape[0], t.shape[2], t.shape[3], 1, 1, 0)
    return from_nhwc(t.permute(0, 2, 3, 1).contiguous())


class LaunchProfiler:
    """Per-launch HIP-event timing of the conv kernels (bench.py roofline leg).  Events are
    recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.items = []

    def run(self, kernel: str, tag: str, flops: float, fn, name: Optional[str] = None, nbytes: float = 0.0, executed: Optional[float] = None):
        """kernel: family (the bench line's roofline families); name: the kernel instantiation really launched (the per-kernel table);
        nbytes: ALGORITHMIC bytes of the launch = unique inputs + outputs + weights of the layer at the dtypes it is stored in."""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        # flops: ALGORITHMIC work of the layer (direct-convolution count); executed: what the matrix pipe really multiplies (Winograd: 16/36)
        self.items.append((kernel, tag, flops, s, e, name or kernel, nbytes, flops if executed is None else executed))

    def summary(self, by_tag: bool = False, by_name: bool = False):
        torch.cuda.synchronize()
        out = {}
        for kernel, tag, flops, s, e, name, nbytes, executed in self.items:
            key = name if by_name else ((kernel, tag) if by_tag else kernel)
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "family": kernel, "name": name, "executed": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += flops
            d["bytes"] += nbytes
            d["executed"] += executed
        return out


_KNAME_CACHE: dict = {}


def _kernel_name(fn, desc, key) -> str:
    """Name of the kernel instantiation the library picks for this descriptor (gpemsr_conv2d*_kernel_name), cached per geometry."""
    nm = _KNAME_CACHE.get(key)
    if nm is None:
        buf = C.create_string_buffer(160)
        rc = fn(C.byref(desc), buf, 160)
        nm = buf.value.decode() if rc == 0 and buf.value else "?"
        _KNAME_CACHE[key] = nm
    return nm


def _layer_bytes(srcs, src_image_stride, out_elems: int, out_esize: int, w_elems: int, w_esize: int, residual, pixmul) -> float:
    """Algorithmic HBM bytes of one layer: every source once, the result once, the weights once, residual / multiplier once."""
    b = 0.0
    for i, s_ in enumerate(srcs):
        imgs = 1 if (src_image_stride is not None and int(src_image_stride[i]) == 0) else s_.n
        b += imgs * s_.h * s_.w * s_.c * s_.esize
    b += out_elems * out_esize + w_elems * w_esize
    if residual is not None:
        b += residual.n * residual.h * residual.w * residual.c * residual.esize
    if pixmul is not None:
        b += pixmul.n * pixmul.h * pixmul.w * 4
    return b


PROFILER: Optional[LaunchProfiler] = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(*acts):
    for a in acts:
        if a is not None and not a.buf.is_cuda:
            raise RuntimeError("gpemsr_amd ops need device (cuda/HIP) memory; there is no CPU path")


@dataclass
class PackedConv:
    """Weights repacked for gpemsr_conv2d: w [tap][cout][cin_pad] (cin padded per source to CK)."""
    w: torch.Tensor
    b: Optional[torch.Tensor]
    ksize: int
    cout: int
    splits: tuple
    ck: int
    transposed: bool = False
    pixel_shuffle: bool = False
    w16: Optional[torch.Tensor] = None     # split-bf16 weights [2][tap][cout][cin] (packing.pack_conv_split), optional
    wb: Optional[torch.Tensor] = None      # bf16 data path: staged-order weights (packing.pack_conv_bf16 / pack_convT_bf16)
    wtap: Optional[torch.Tensor] = None    # bf16 data path, 64 -> 1 3x3: tap fragments (packing.pack_cout1_taps)
    wrow7: Optional[torch.Tensor] = None   # bf16 data path, 16 -> 2 7x7 (SpyNet flow update): row-sum fragments (packing.pack_rowsum7)
    wtap32: Optional[torch.Tensor] = None  # fp32 activations, 64 -> 1 3x3: fp32 tap fragments (packing.pack_cout1_taps_f32)
    wrow7_32: Optional[torch.Tensor] = None  # fp32 activations, 16 -> 2 7x7: row-sum fragments (packing.pack_rowsum7_f32)
    w7c8: Optional[torch.Tensor] = None    # bf16 data path, 8 -> 32 7x7 (SpyNet stems): four taps per MFMA (packing.pack_conv7_c8_cout32)
    w7c16: Optional[torch.Tensor] = None   # bf16 data path, 32 -> 16 7x7: 16x16x32 MFMA fragments (packing.pack_conv7_c32_cout16)
    wino: Optional[torch.Tensor] = None    # fp32, 3x3 stride 1: Winograd F(2x2,3x3) weights U[16][cout][cin] (packing.pack_winograd; descriptor.transposed = 3)
    wpair7: Optional[torch.Tensor] = None  # fp32, cin -> 16 7x7: row-pair form weights (packing.pack_rowpair7; descriptor.transposed = 2)
    wino4: Optional[torch.Tensor] = None   # fp32, 3x3 stride 1, >= 64 input channels: Winograd F(4x4,3x3) weights U[cin/8][36][2][cout][4] (packing.pack_winograd4; descriptor.transposed = 5)
    wino77: Optional[torch.Tensor] = None  # fp32, 7x7 stride 1: 2-D Winograd F(2x2, 7x7) weights U[cin/8][64][2][cout][4] (packing.pack_winograd77; descriptor.transposed = 6)
    wino7: Optional[torch.Tensor] = None   # fp32, 7x7 stride 1: 1-D Winograd F(2, 7) weights U[cin/8][7][8][2][cout][4] (packing.pack_winograd7; descriptor.transposed = 4)
    wrows: Optional[torch.Tensor] = None   # bf16 data path, DCN 64 -> 64: plain rows [cout][9 taps][64 channels] bf16 (packing.pack_dcn_rows_bf16; csrc/dcn_bf16.hip)
    algo_cin: Optional[int] = None         # input channels of the ALGORITHMIC product when the packed form multiplies more (three-product linear):
                                           # the profiler's flop count uses this, so split products are not credited as extra work

    @property
    def cin(self) -> int:
        return sum(self.splits)


def conv2d(srcs, pc: "PackedConv", *args, out_u8: Optional[torch.Tensor] = None, **kw) -> "Act":
    """gpemsr_conv2d and its specialisations (see _conv2d).  out_u8 (1-channel fp32 results only): also store the reference's
    tensor2img of the result there ([n, h, w] uint8) -- fused into the 64 -> 1 tap kernel when that runs, a separate pass otherwise."""
    if out_u8 is None:
        return _conv2d(srcs, pc, *args, **kw)
    assert pc.cout == 1 and out_u8.dtype == torch.uint8 and out_u8.is_contiguous()
    fused = [False]
    r = _conv2d(srcs, pc, *args, out_u8=out_u8, _u8_fused=fused, **kw)
    if not fused[0]:
        _u8_pass(r, out_u8)
    return r


def conv_cosine_ok(src: "Act", pc: "PackedConv") -> bool:
    """Shapes gpemsr_conv_desc.cos_partials takes: 3x3 stride 1, 33..64 output channels, height % 16 == 0, width % 32 == 0, aligned rows."""
    return (not src.bf16 and pc.ksize == 3 and 32 < pc.cout <= 64 and pc.cout % 4 == 0 and not pc.transposed and not pc.pixel_shuffle
            and src.h % 16 == 0 and src.w % 32 == 0 and src.ld % 4 == 0 and src.ptr % 16 == 0 and len(pc.splits) == 1 and src.c % 8 == 0)


def _conv2d_cosine(srcs, pc: "PackedConv", act: int, a: "Act", tag: str, winograd: bool = False, winograd4: bool = True) -> "Act":
    """act(conv(src)) is NOT stored: its 16x16-patch cosine against `a` (R:model/GPEMSR.py:387-395) comes from partial sums formed in the
    convolution's epilogue (gpemsr_conv_desc.cos_partials) + gpemsr_patch_cosine_finish -> [n, h/16, w/16, 1]."""
    lib = _abi.load()
    s0 = srcs[0]
    n, h, w = s0.n, s0.h, s0.w
    assert conv_cosine_ok(s0, pc) and (a.n, a.h, a.w, a.c) == (n, h, w, pc.cout) and a.ld % 4 == 0 and a.ptr % 16 == 0 and not a.bf16
    ws = torch.empty(n * (h // 4) * (w // 16) * 4, dtype=torch.float32, device=s0.buf.device)
    d = _abi.ConvDesc()
    d.n, d.h, d.w, d.nsrc = n, h, w, 1
    d.src[0].ptr, d.src[0].ld, d.src[0].c = s0.ptr, s0.ld, s0.c
    d.src_image_stride[0] = -1
    d.cout, d.ksize, d.stride, d.transposed = pc.cout, 3, 1, 0
    d.weight, d.weight_image_stride = pc.w.data_ptr(), 0
    d.bias = pc.b.data_ptr() if pc.b is not None else None
    d.act = act
    d.residual, d.res_ld = a.ptr, a.ld
    d.out, d.out_ld = a.ptr, a.ld                  # (never written in this mode; the descriptor wants a valid aligned row layout)
    d.cos_partials = ws.data_ptr()
    flops = 2.0 * n * h * w * pc.cout * pc.cin * 9.0
    executed = None
    if winograd and pc.wino is not None and pc.cout == 64 and s0.c % 8 == 0 and h % 16 == 0 and w % 32 == 0:
        d.transposed, d.weight = 3, pc.wino.data_ptr()          # the Winograd form leaves the same records (csrc/conv_wino.hip)
        executed = flops * 16.0 / 36.0
        if winograd4 and pc.wino4 is not None and act in (ACT_NONE, ACT_RELU, ACT_LRELU) and _wino4_addressable([s0], a):
            d.transposed, d.weight = 5, pc.wino4.data_ptr()     # ... and so does the F(4x4) form (csrc/conv_wino4.hip)
            executed = flops * 36.0 / 144.0
    if PROFILER is not None:
        nm = _kernel_name(lib.gpemsr_conv2d_kernel_name, d, ("f32cos", n, h, w, s0.c, pc.cout, int(d.transposed)))
        nb = 4.0 * (n * h * w * (s0.c + pc.cout) + pc.cout * pc.cin * 9)          # source + the operand map read; the result is not stored
        PROFILER.run("conv_mfma", tag, flops, lambda: _abi.check(lib.gpemsr_conv2d(C.byref(d), _stream()), "conv2d"), name=nm, nbytes=nb, executed=executed)
    else:
        _abi.check(lib.gpemsr_conv2d(C.byref(d), _stream()), "conv2d")
    out = new_act(n, h // 16, w // 16, 1, device=s0.buf.device)
    _abi.check(lib.gpemsr_patch_cosine_finish(ws.data_ptr(), n, h // 16, w // 16, out.ptr, _stream()), "patch_cosine_finish")
    return out


def _conv2d(srcs, pc: PackedConv, act: int = ACT_NONE, stride: int = 1, residual: Optional[Act] = None,
           pixmul: Optional[Act] = None, out: Optional[Act] = None, weight_image_stride: int = 0,
           src_image_stride: Optional[Sequence[int]] = None, force_mfma: bool = False, tag: str = "",
           precision: str = "fp32", out_u8: Optional[torch.Tensor] = None, _u8_fused: Optional[list] = None, **kw16) -> Act:
    lib = _abi.load()
    if isinstance(srcs, Act):
        srcs = [srcs]
    _require_gpu(*srcs)
    assert len(srcs) == len(pc.splits) and all(s.c == c for s, c in zip(srcs, pc.splits)), \
        f"source channels {[s.c for s in srcs]} != packed splits {pc.splits}"
    if precision == "bf16":
        return conv2d_bf16(srcs, pc, act, stride, residual, pixmul, out, weight_image_stride, src_image_stride, force_mfma, tag, out_u8=out_u8,
                           _u8_fused=_u8_fused, **kw16)
    gn_stats = bool(kw16.pop("gn_stats", False))          # fp32 too: GroupNorm partial sums from the epilogue (gpemsr_conv_desc.gn_partials)
    winograd = bool(kw16.pop("winograd", False))          # fp32: the Winograd F(2x2,3x3) form where the layer qualifies (winograd_ok)
    winograd4 = bool(kw16.pop("winograd4", True))         # ... and the F(4x4,3x3) form where `pc.wino4` is packed (winograd4_ok)
    direct7 = bool(kw16.pop("direct7", False))            # fp32: keep the direct form of a 7x7 layer that has F(2, 7) weights packed (A/B, tests)
    a_affine32 = kw16.pop("a_affine", None) if precision == "fp32" else None   # fp32: (scale, shift, relu) tables folded into the F(4x4) form's input transform
    cos_with = kw16.pop("cos_with", None)                  # fp32: patch cosine of the result against this tensor, result not stored
    if cos_with is not None:
        return _conv2d_cosine(srcs, pc, act, cos_with, tag, winograd=winograd, winograd4=winograd4)
    assert not kw16 or not any(kw16.values()), f"{sorted(kw16)} are options of the bf16 data path"
    s0 = srcs[0]
    n, h, w = s0.n, s0.h, s0.w
    k = pc.ksize
    if pc.transposed:
        oh, ow = 2 * h, 2 * w
    else:
        oh, ow = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    OH, OW, oc = (2 * oh, 2 * ow, pc.cout // 4) if pc.pixel_shuffle else (oh, ow, pc.cout)
    if out is None:
        out = new_act(n, OH, OW, oc, device=s0.buf.device)
    assert (out.n, out.h, out.w, out.c) == (n, OH, OW, oc), f"out geometry {(out.n, out.h, out.w, out.c)} != {(n, OH, OW, oc)}"
    use_direct = (not force_mfma and pc.cout <= 16 and (pc.cout <= 2 or pc.cin <= 16) and len(srcs) == 1 and k >= 3
                  and pc.ck == 8 and not pc.transposed
                  and not pc.pixel_shuffle and pixmul is None and weight_image_stride == 0 and src_image_stride is None)
    if use_direct and precision in ("bf16", "bf16op") and pc.w16 is not None and k == 7 and stride == 1 and s0.c % 16 == 0 and residual is not None:
        use_direct = False          # SpyNet's 16 -> 2 7x7 output conv: even padded to 32 couts the bf16 matrix pipe beats the VALU kernel
    if stride == 4:
        assert use_direct, "stride 4 is only available through the direct kernel"
    use_stem = (not force_mfma and len(srcs) == 1 and s0.c == 1 and s0.ld == 1 and k == 3 and stride == 1 and pc.ck == 8
                and pc.cout % 4 == 0 and pc.cout > 16 and not pc.transposed and not pc.pixel_shuffle and pixmul is None
                and residual is None and weight_image_stride == 0 and src_image_stride is None and out.ld % 4 == 0)
    if use_stem:
        def _go_stem():
            _abi.check(lib.gpemsr_conv2d_stem1(s0.ptr, n, h, w, pc.w.data_ptr(), pc.b.data_ptr() if pc.b is not None else None,
                                               pc.cout, act, out.ptr, out.ld, _stream()), "conv2d_stem1")
        if PROFILER is not None:
            PROFILER.run("conv_stem1", tag, 2.0 * n * oh * ow * pc.cout * 9, _go_stem, name="conv_stem1_kernel<float>",
                         nbytes=4.0 * (n * h * w + n * oh * ow * pc.cout))
        else:
            _go_stem()
        return out
    # algorithmic FLOPs of this launch (2*MAC, un-padded channel counts)
    taps = 9.0 / 4.0 if pc.transposed else float(k * k)
    flops = 2.0 * n * oh * ow * pc.cout * (pc.algo_cin or pc.cin) * taps
    if (use_direct and pc.wrow7_32 is not None and pc.cin == 16 and s0.c == 16 and pc.cout == 2 and k == 7 and stride == 1 and act == ACT_NONE and s0.ld % 4 == 0
            and not s0.bf16 and (residual is None or residual.c == 2)):
        def _go_row7_32():      # SpyNet's flow update as row sums on the fp32 matrix pipe (csrc/tap_sum.hip)
            _abi.check(lib.gpemsr_conv7_c16_cout2_f32(s0.ptr, n, h, w, s0.ld, pc.wrow7_32.data_ptr(), pc.b.data_ptr() if pc.b is not None else None,
                                                      residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                                                      out.ptr, out.ld, _stream()), "conv7_c16_cout2_f32")
        if PROFILER is not None:
            PROFILER.run("tap_sum", tag, flops, _go_row7_32, name="rowsum7_kernel<float>", nbytes=4.0 * n * h * w * (16 + 2 + (2 if residual is not None else 0)))
        else:
            _go_row7_32()
        return out
    if use_direct and pc.wtap32 is not None and pc.cin == 64 and pc.cout == 1 and k == 3 and stride == 1 and s0.ld % 4 == 0 and not s0.bf16:
        def _go_taps32():       # 64 -> 1 as tap partial products on the fp32 matrix pipe (csrc/tap_sum.hip): the tensor is read once
            _abi.check(lib.gpemsr_conv_c64_cout1_f32(s0.ptr, n, h, w, s0.ld, pc.wtap32.data_ptr(), pc.b.data_ptr() if pc.b is not None else None, act,
                                                     residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                                                     out.ptr, out.ld, out_u8.data_ptr() if (out_u8 is not None and out.ld == 1) else None, _stream()),
                       "conv_c64_cout1_f32")
        if out_u8 is not None and out.ld == 1:
            _u8_fused[0] = True
        if PROFILER is not None:
            PROFILER.run("tap_sum", tag, flops, _go_taps32, name="tap_sum_kernel<false,float>", nbytes=4.0 * n * h * w * (64 + 1 + (1 if residual is not None else 0)))
        else:
            _go_taps32()
        return out
    if use_direct:
        def _go():
            _abi.check(lib.gpemsr_conv2d_direct(s0.ptr, n, h, w, s0.ld, s0.c, pc.w.data_ptr(),
                                                pc.b.data_ptr() if pc.b is not None else None, pc.cout, k, stride, act,
                                                residual.ptr if residual is not None else None,
                                                residual.ld if residual is not None else 0, out.ptr, out.ld, _stream()),
                       "conv2d_direct")
        if PROFILER is not None:
            PROFILER.run("conv_direct", tag, flops, _go, name="conv_direct_kernel<float>", nbytes=4.0 * (n * h * w * s0.c + n * OH * OW * oc))
        else:
            _go()
        return out
    d = _abi.ConvDesc()
    d.n, d.h, d.w, d.nsrc = n, h, w, len(srcs)
    for i, s in enumerate(srcs):
        assert (s.n, s.h, s.w) == (n, h, w) or (src_image_stride is not None)
        d.src[i].ptr, d.src[i].ld, d.src[i].c = s.ptr, s.ld, s.c
        d.src_image_stride[i] = -1 if src_image_stride is None else int(src_image_stride[i])
    d.cout, d.ksize, d.stride, d.transposed = pc.cout, k, stride, int(pc.transposed)
    d.weight, d.weight_image_stride = pc.w.data_ptr(), int(weight_image_stride)
    d.bias = pc.b.data_ptr() if pc.b is not None else None
    d.act = act
    if residual is not None:
        assert (residual.n, residual.h, residual.w, residual.c) == (n, OH, OW, oc)
        d.residual, d.res_ld = residual.ptr, residual.ld
    if pixmul is not None:
        assert pixmul.c == 1 and pixmul.ld == 1 and (pixmul.n, pixmul.h, pixmul.w) == (n, OH, OW)
        d.pixmul = pixmul.ptr
    d.pixel_shuffle = int(pc.pixel_shuffle)
    d.out, d.out_ld = out.ptr, out.ld
    gemm16 = (k == 1 and stride == 1 and not pc.transposed and not pc.pixel_shuffle)
    use_split = (precision in ("bf16x3", "bf16", "bf16op") and pc.w16 is not None and (k in (3, 7) or gemm16) and (stride == 1 or pc.transposed)
                 and not (pc.transposed and pixmul is not None) and (weight_image_stride == 0 or gemm16) and src_image_stride is None
                 and all(s.c % (32 if gemm16 else 16) == 0 and s.ld % 4 == 0 and s.ptr % 16 == 0 for s in srcs))
    if use_split:
        nsplit = 2 if precision == "bf16x3" else 1
        if weight_image_stride != 0:                 # per-image B from split_pack_rows: [n][plane][k/16][half][rows][8]
            assert pc.w16.dim() == 6 and pc.w16.shape[1] == 2
            plane = pc.w16[0, 0].numel()
            d.weight_image_stride = 2 * plane          # bf16 elements
        else:
            plane = pc.w16[0].numel()
        def _go_split():
            _abi.check(lib.gpemsr_conv2d_split(C.byref(d), pc.w16.data_ptr(), plane, nsplit, _stream()), "conv2d_split")
        if PROFILER is not None:
            PROFILER.run("conv_split", tag, flops, _go_split, name="conv_split_kernel",
                         nbytes=_layer_bytes(srcs, src_image_stride, n * OH * OW * oc, 4, int(pc.cout * pc.cin * taps), 2 * nsplit, residual, pixmul))
        else:
            _go_split()
        return out
    executed = None
    if winograd and precision == "fp32" and winograd_ok(srcs, pc, stride, out, residual) and weight_image_stride == 0 and src_image_stride is None:
        d.transposed, d.weight = 3, pc.wino.data_ptr()          # 16 multiplies per 2x2 outputs instead of 36 (csrc/conv_wino.hip)
        executed = flops * 16.0 / 36.0
        if winograd4 and act in (ACT_NONE, ACT_RELU, ACT_LRELU) and winograd4_ok(srcs, pc, residual, pixmul, out):
            d.transposed, d.weight = 5, pc.wino4.data_ptr()     # 36 multiplies per 4x4 outputs instead of 144 (csrc/conv_wino4.hip)
            executed = flops * 36.0 / 144.0
    if (winograd and winograd4 and int(d.transposed) == 0 and precision == "fp32" and pc.wino4 is not None and pc.cout % 64 != 0 and act in (ACT_NONE, ACT_RELU, ACT_LRELU)
            and winograd4_padded_ok(srcs, pc, stride, out, residual, pixmul) and weight_image_stride == 0 and src_image_stride is None):
        d.transposed, d.weight = 5, pc.wino4.data_ptr()         # cout % 64 != 0 (the 216-channel offset convs of the DCN packs): zero rows in U, no store
        executed = flops * 36.0 / 144.0 * (-(-pc.cout // 64) * 64) / pc.cout
    if a_affine32 is not None:
        assert int(d.transposed) == 5 and conv_affine_source_ok32(srcs, pc, act, residual, pixmul), "a_affine: only the F(4x4,3x3) form folds a source GroupNorm (conv_affine_source_ok32)"
        sc, sh, relu = a_affine32
        assert relu and sc.numel() == n * srcs[0].c and sh.numel() == n * srcs[0].c and sc.dtype == torch.float32
        d.a_scale, d.a_shift, d.a_relu = sc.data_ptr(), sh.data_ptr(), 1
    if (gn_stats and precision == "fp32" and act == ACT_NONE and residual is None and pixmul is None and not pc.transposed and not pc.pixel_shuffle
            and pc.cout % 4 == 0 and out.ld % 4 == 0 and out.ptr % 16 == 0 and (pc.b is None or pc.b.data_ptr() % 16 == 0)
            # the partial-sum epilogue exists only in the LDS-DMA instantiations (conv_mfma.hip: GP_REQUIRE(dma && BN >= 32)): sources it
            # cannot stage by DMA take the statistics pass instead of failing the launch
            and all(s.ld % 4 == 0 and s.ptr % 16 == 0 and s.c % (32 if k == 1 else 8) == 0 for s in srcs)):
        parts = lib.gpemsr_conv2d_gn_parts(C.byref(d))
        if parts > 0:                                   # (< 0: this shape takes the general epilogue -- the statistics pass runs instead)
            gws = torch.empty(n * parts * pc.cout * 2, dtype=torch.float32, device=s0.buf.device)
            d.gn_partials = gws.data_ptr()
            out.gn = (gws, parts)
    if (pc.wpair7 is not None and precision == "fp32" and k == 7 and stride == 1 and pc.cout == 16 and not pc.transposed and not pc.pixel_shuffle
            and pixmul is None and weight_image_stride == 0 and src_image_stride is None and out.ld % 4 == 0 and out.ptr % 16 == 0
            and (residual is None or (residual.ld % 4 == 0 and residual.ptr % 16 == 0))):
        d.transposed, d.weight = 2, pc.wpair7.data_ptr()       # row-pair form: both halves of the 32-row matrix tile do useful work
    if (pc.wino7 is not None and not direct7 and precision == "fp32" and winograd7_ok(srcs, pc, stride, out, residual, pixmul) and weight_image_stride == 0
            and src_image_stride is None and not d.gn_partials):
        d.transposed, d.weight = 4, pc.wino7.data_ptr()         # 8 multiplies per output pair and filter row instead of 14 (csrc/conv7_wino.hip)
        executed = flops * 8.0 / 14.0
    if (pc.wino77 is not None and not direct7 and precision == "fp32" and int(d.transposed) in (0, 2, 4) and winograd77_ok(srcs, pc, stride, out, residual, pixmul, act)
            and weight_image_stride == 0 and src_image_stride is None and not d.gn_partials):
        d.transposed, d.weight = 6, pc.wino77.data_ptr()        # 64 multiplies per 2x2 outputs instead of 196 (csrc/conv7_wino2d.hip)
        executed = flops * 64.0 / 196.0
    if PROFILER is not None:
        nm = _kernel_name(lib.gpemsr_conv2d_kernel_name, d, ("f32", n, h, w, tuple((s_.c, s_.ld % 4, s_.ptr % 16) for s_ in srcs), pc.cout, k, stride, int(d.transposed),
                                                             int(pc.pixel_shuffle), bool(d.gn_partials), out.ld % 4, residual is not None, pixmul is not None, act, a_affine32 is not None))
        wimgs = n if weight_image_stride != 0 else 1
        nb = _layer_bytes(srcs, src_image_stride, n * OH * OW * oc, 4, int(wimgs * pc.cout * pc.cin * taps), 4, residual, pixmul)
        PROFILER.run("conv_mfma", tag, flops, lambda: _abi.check(lib.gpemsr_conv2d(C.byref(d), _stream()), "conv2d"), name=nm, nbytes=nb, executed=executed)
    else:
        _abi.check(lib.gpemsr_conv2d(C.byref(d), _stream()), "conv2d")
    return out


def winograd4_ok(srcs, pc: "PackedConv", residual: Optional["Act"] = None, pixmul: Optional["Act"] = None, out: Optional["Act"] = None) -> bool:
    """Layers (among those winograd_ok accepts) the F(4x4, 3x3) form takes: `pc.wino4` packed (the engine packs it for layers of >= 64 input
    channels), cout % 64 == 0 (a padded form exists for other couts >= 128: winograd4_padded_ok); plain store, + residual (+ pixel
    multiplier), or PixelShuffle (cout % 256 == 0, nothing else); output / residual rows 8-byte aligned.  (Whether a small map is worth its
    16 x 32 pixel tiles is the caller's call: `winograd4=False`.)"""
    if pc.wino4 is None or pc.cout % 64 != 0 or (pixmul is not None and residual is None) or not _wino4_addressable(srcs, out, residual):
        return False                                        # (cout % 64 != 0: winograd4_padded_ok)
    return not pc.pixel_shuffle or (pc.cout % 256 == 0 and residual is None and pixmul is None)


def _wino4_addressable(srcs, *pair_tensors) -> bool:
    """The F(4x4) kernel's image DMA uses 32-bit byte offsets and a 24-bit pixel index per source image; its epilogue stores (and reads the
    residual in) cout PAIRS: 8-byte aligned rows for `pair_tensors` = output / residual (csrc/conv_wino4.hip: GP_REQUIRE)."""
    return (all(s_.h * s_.w < (1 << 24) and s_.ld < (1 << 22) and s_.h * s_.w * s_.ld * 4 < (1 << 32) for s_ in srcs)
            and all(t is None or (t.ld % 2 == 0 and t.ptr % 8 == 0) for t in pair_tensors))


def winograd4_padded_ok(srcs, pc: "PackedConv", stride: int = 1, out: Optional["Act"] = None, residual: Optional["Act"] = None, pixmul: Optional["Act"] = None) -> bool:
    """Layers with cout % 64 != 0 the F(4x4, 3x3) form takes through zero-padded weights (`pc.wino4` packed): 3x3 stride 1, fp32 sources of
    c % 8 == 0 with 16-byte aligned rows, plain store or + residual."""
    if (pc.wino4 is None or pc.ksize != 3 or stride != 1 or pc.transposed or pc.pixel_shuffle or pixmul is not None or pc.cout % 2 != 0
            or not _wino4_addressable(srcs, out, residual)):
        return False
    if any(s_.bf16 or s_.c % 8 != 0 or s_.ld % 4 != 0 or s_.ptr % 16 != 0 for s_ in srcs):
        return False
    return (out is None or not out.bf16) and (residual is None or not residual.bf16)


def conv_affine_source_ok32(srcs, pc: "PackedConv", act: int = ACT_NONE, residual: Optional["Act"] = None, pixmul: Optional["Act"] = None) -> bool:
    """fp32 path: can ``conv2d(..., winograd=True, a_affine=...)`` fold the GroupNorm + ReLU of its source?  One fp32 source, no activation /
    residual / multiplier / PixelShuffle, and the layer runs in the F(4x4,3x3) form (winograd_ok and winograd4_ok)."""
    if isinstance(srcs, Act):
        srcs = [srcs]
    return (len(srcs) == 1 and act == ACT_NONE and residual is None and pixmul is None and not pc.pixel_shuffle
            and winograd_ok(srcs, pc) and winograd4_ok(srcs, pc))


def winograd7_ok(srcs, pc: "PackedConv", stride: int = 1, out: Optional["Act"] = None, residual: Optional["Act"] = None, pixmul: Optional["Act"] = None) -> bool:
    """Layers the F(2, 7) row form of gpemsr_conv2d takes: 7x7, stride 1, ONE fp32 source of c % 8 == 0 channels with 16-byte aligned rows,
    cout % 32 == 0, plain store, images at least 64 pixels wide (the tile is 4 x 64 pixels); `pc.wino7` packed."""
    if pc.wino7 is None or pc.ksize != 7 or stride != 1 or pc.transposed or pc.pixel_shuffle or pc.cout % 32 != 0 or len(srcs) != 1:
        return False
    s_ = srcs[0]
    if s_.bf16 or s_.c % 8 != 0 or s_.ld % 4 != 0 or s_.ptr % 16 != 0 or s_.w < 64 or residual is not None or pixmul is not None:
        return False
    return out is None or not out.bf16


def winograd77_ok(srcs, pc: "PackedConv", stride: int = 1, out: Optional["Act"] = None, residual: Optional["Act"] = None, pixmul: Optional["Act"] = None,
                  act: int = ACT_NONE) -> bool:
    """Layers the 2-D F(2x2, 7x7) form of gpemsr_conv2d takes: 7x7, stride 1, ONE fp32 source of c % 8 == 0 channels with 16-byte aligned rows,
    cout % 16 == 0 (32 couts per workgroup, or 16 on the 16x16x4 MFMA shape), act NONE / RELU / LRELU, plain store into 8-byte aligned rows, a
    map that fills at least 2/3 of its 8 x 16 pixel tiles;
    `pc.wino77` packed."""
    if pc.wino77 is None or pc.ksize != 7 or stride != 1 or pc.transposed or pc.pixel_shuffle or pc.cout % 16 != 0 or len(srcs) != 1:
        return False
    s_ = srcs[0]
    if s_.bf16 or s_.c % 8 != 0 or s_.ld % 4 != 0 or s_.ptr % 16 != 0 or residual is not None or pixmul is not None or act not in (ACT_NONE, ACT_RELU, ACT_LRELU):
        return False
    if 3 * s_.h * s_.w < 2 * (-(-s_.h // 8) * 8) * (-(-s_.w // 16) * 16) or s_.h * s_.w * s_.ld * 4 >= (1 << 32):
        return False
    return out is None or (not out.bf16 and out.ld % 2 == 0 and out.ptr % 8 == 0)


def winograd_ok(srcs, pc: "PackedConv", stride: int = 1, out: Optional["Act"] = None, residual: Optional["Act"] = None) -> bool:
    """Layers the Winograd form of gpemsr_conv2d takes: 3x3, stride 1, fp32 sources of c % 8 == 0 with 16-byte aligned rows, cout % 32 == 0,
    plain store; `pc.wino` packed."""
    if pc.wino is None or pc.ksize != 3 or stride != 1 or pc.transposed or pc.cout % 32 != 0:
        return False
    if pc.pixel_shuffle and (pc.cout % 64 != 0 or residual is not None):
        return False
    if any(s_.bf16 or s_.c % 8 != 0 or s_.ld % 4 != 0 or s_.ptr % 16 != 0 for s_ in srcs):
        return False
    if out is not None and (out.ld % 4 != 0 or out.ptr % 16 != 0):
        return False
    if residual is not None and (residual.ld % 4 != 0 or residual.ptr % 16 != 0):
        return False
    return pc.b is None or pc.b.data_ptr() % 16 == 0


def groupnorm_relu(x: Act, gamma: torch.Tensor, beta: torch.Tensor, relu: bool = True, residual: Optional[Act] = None,
                   out: Optional[Act] = None, groups: int = 32, eps: float = 1e-6) -> Act:
    lib = _abi.load()
    _require_gpu(x)
    hw = x.h * x.w
    if x.bf16:
        # first pass: from the producing convolution's epilogue when it left partial sums (x.gn), else one read of x
        dev = x.buf.device
        if x.gn is not None:
            ws, parts = x.gn
        else:
            parts = max(1, min(64, hw // 64))
            ws = torch.empty(x.n * parts * x.c * 2, dtype=torch.float32, device=dev)
            _abi.check(lib.gpemsr_groupnorm_stats_bf16(x.ptr, x.n, hw, x.c, x.ld, ws.data_ptr(), parts, _stream()), "groupnorm_stats_bf16")
        mr = torch.empty(x.n * groups * 2, dtype=torch.float32, device=dev)
        _abi.check(lib.gpemsr_groupnorm_finish(ws.data_ptr(), x.n, hw, x.c, groups, parts, eps, mr.data_ptr(), _stream()), "groupnorm_finish")
        if out is None:
            out = new_act(x.n, x.h, x.w, x.c, device=dev, bf16=True)
        assert out.bf16 and (residual is None or residual.bf16)
        _abi.check(lib.gpemsr_groupnorm_apply_bf16(x.ptr, x.n, hw, x.c, x.ld, groups, mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                   int(relu), residual.ptr if residual is not None else None,
                                                   residual.ld if residual is not None else 0, out.ptr, out.ld, _stream()), "groupnorm_apply_bf16")
        out.gn = None
        return out
    parts = max(1, min(64, hw // 64))
    ws = torch.empty(x.n * parts * x.c * 2 + x.n * groups * 2, dtype=torch.float32, device=x.buf.device)
    mr = ws[x.n * parts * x.c * 2:]
    if x.gn is not None:        # the producing convolution left the partial sums (gpemsr_conv_desc.gn_partials): no statistics pass
        gws, gparts = x.gn
        _abi.check(lib.gpemsr_groupnorm_finish(gws.data_ptr(), x.n, hw, x.c, groups, gparts, eps, mr.data_ptr(), _stream()), "groupnorm_finish")
        x.gn = None
    else:
        _abi.check(lib.gpemsr_groupnorm_stats(x.ptr, x.n, hw, x.c, x.ld, groups, eps, ws.data_ptr(), parts, mr.data_ptr(),
                                              _stream()), "groupnorm_stats")
    if out is None:
        out = new_act(x.n, x.h, x.w, x.c, device=x.buf.device)
    _abi.check(lib.gpemsr_groupnorm_apply(x.ptr, x.n, hw, x.c, x.ld, groups, mr.data_ptr(), gamma.data_ptr(),
                                          beta.data_ptr(), int(relu), residual.ptr if residual is not None else None,
                                          residual.ld if residual is not None else 0, out.ptr, out.ld, _stream()),
               "groupnorm_apply")
    return out


def softmax_rows_(x: torch.Tensor, rows: int, cols: int, ld: Optional[int] = None):
    if ld is None or ld == cols:
        _abi.check(_abi.load().gpemsr_softmax_rows(x.data_ptr(), rows, cols, _stream()), "softmax_rows")
    else:
        _abi.check(_abi.load().gpemsr_softmax_rows_ld(x.data_ptr(), rows, cols, ld, _stream()), "softmax_rows_ld")


def argmax_rows(x: Act) -> torch.Tensor:
    assert x.ld == x.c
    idx = torch.empty(x.pixels, dtype=torch.int32, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_argmax_rows(x.ptr, x.pixels, x.c, idx.data_ptr(), _stream()), "argmax_rows")
    return idx


def gather_rows(table: torch.Tensor, idx: torch.Tensor, n: int, h: int, w: int) -> Act:
    dim = table.shape[1]
    out = new_act(n, h, w, dim, device=table.device)
    _abi.check(_abi.load().gpemsr_gather_rows(table.data_ptr(), dim, idx.data_ptr(), n * h * w, out.ptr, out.ld, _stream()),
               "gather_rows")
    return out


def bilinear(x: Act, oh: int, ow: int, align_corners: bool = False, mul: float = 1.0, out: Optional[Act] = None) -> Act:
    _require_gpu(x)
    if x.bf16:
        if out is None:
            out = new_act(x.n, oh, ow, x.c, device=x.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_bilinear_bf16(x.ptr, x.n, x.h, x.w, x.c, x.ld, oh, ow, int(align_corners), mul, out.ptr,
                                                    out.ld, _stream()), "bilinear_bf16")
        return out
    if out is None:
        out = new_act(x.n, oh, ow, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_bilinear(x.ptr, x.n, x.h, x.w, x.c, x.ld, oh, ow, int(align_corners), mul, out.ptr,
                                           out.ld, _stream()), "bilinear")
    return out


def avgpool2(x: Act) -> Act:
    out = new_act(x.n, x.h // 2, x.w // 2, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_avgpool2(x.ptr, x.n, x.h, x.w, x.c, x.ld, out.ptr, out.ld, _stream()), "avgpool2")
    return out


def pool3s2_maxavg(x: Act) -> Act:
    oh, ow = (x.h - 1) // 2 + 1, (x.w - 1) // 2 + 1
    if x.bf16:
        out = new_act(x.n, oh, ow, 2 * x.c, device=x.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_pool3s2_maxavg_bf16(x.ptr, x.n, x.h, x.w, x.c, x.ld, out.ptr, out.ld, _stream()), "pool3s2_bf16")
        return out
    out = new_act(x.n, oh, ow, 2 * x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_pool3s2_maxavg(x.ptr, x.n, x.h, x.w, x.c, x.ld, out.ptr, out.ld, _stream()), "pool3s2")
    return out


def spynet_prep(ref: Act, supp: Act, flow_coarse: Optional[Act], mean3, std3, pad16: bool = False):
    assert ref.c == 1 and ref.ld == 1 and supp.c == 1 and supp.ld == 1
    up = new_act(ref.n, ref.h, ref.w, 2, device=ref.buf.device)
    inp = new_act(ref.n, ref.h, ref.w, 16 if pad16 else 8, device=ref.buf.device)   # pad16: channels 8..15 are zeros
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    if flow_coarse is not None:
        assert flow_coarse.ld == 2 and flow_coarse.h == ref.h // 2 and flow_coarse.w == ref.w // 2
    _abi.check(_abi.load().gpemsr_spynet_prep(ref.ptr, supp.ptr, flow_coarse.ptr if flow_coarse is not None else None,
                                              ref.n, ref.h, ref.w, m, s, up.ptr, inp.ptr, inp.ld, _stream()), "spynet_prep")
    return up, inp


def dcn_columns(x: Act, om: Act, groups: int) -> Act:
    if x.bf16:
        assert not om.bf16, "deformable offsets / mask logits stay fp32"
        col = new_act(x.n, x.h, x.w, 9 * x.c, device=x.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_dcn_columns_bf16(x.ptr, x.n, x.h, x.w, x.c, x.ld, om.ptr, om.ld, groups, col.ptr, _stream()),
                   "dcn_columns_bf16")
        return col
    col = new_act(x.n, x.h, x.w, 9 * x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_dcn_columns(x.ptr, x.n, x.h, x.w, x.c, x.ld, om.ptr, om.ld, groups, col.ptr, _stream()),
               "dcn_columns")
    return col


def dcn_conv_ok(x: Act, om: Act, pc: "PackedConv", groups: int) -> bool:
    """True when gpemsr_dcn_conv_bf16 (deformable sampling + contraction in one kernel) takes this layer."""
    return (x.bf16 and not om.bf16 and pc.wrows is not None and x.c == 64 and pc.cout == 64 and groups == 8 and om.c >= 216 and x.ld % 8 == 0
            and x.ptr % 16 == 0 and (om.n, om.h, om.w) == (x.n, x.h, x.w))


def dcn_conv_bf16(x: Act, om: Act, pc: "PackedConv", act: int = ACT_NONE, out: Optional[Act] = None, tag: str = "") -> Act:
    """DCNv2 after its conv_offset (basicsr DCNv2Pack -> torchvision deform_conv2d; R:model/GPEMSR.py:79-94) on the bf16 path: x bf16
    [n][h][w][64], om fp32 [n][h][w][216] (offsets | mask logits) -> bf16 [n][h][w][64]; the column tensor stays in LDS."""
    _require_gpu(x, om)
    assert dcn_conv_ok(x, om, pc, 8)
    if out is None:
        out = new_act(x.n, x.h, x.w, 64, device=x.buf.device, bf16=True)
    assert out.bf16 and (out.n, out.h, out.w, out.c) == (x.n, x.h, x.w, 64) and out.ld % 8 == 0 and out.ptr % 16 == 0
    lib = _abi.load()

    def _go():
        _abi.check(lib.gpemsr_dcn_conv_bf16(x.ptr, x.n, x.h, x.w, x.ld, om.ptr, om.ld, pc.wrows.data_ptr(), pc.b.data_ptr() if pc.b is not None else None, act,
                                            out.ptr, out.ld, _stream()), "dcn_conv_bf16")
    if PROFILER is not None:
        px = float(x.n * x.h * x.w)
        PROFILER.run("conv_bf16", tag, 2.0 * px * 64 * 576, _go, name="dcn_fused16_kernel", nbytes=px * (2.0 * 64 + 4.0 * 216 + 2.0 * 64) + 2.0 * 64 * 576)
    else:
        _go()
    return out


def patch_cosine(a: Act, b: Act) -> Act:
    assert a.ld == a.c and b.ld == b.c
    out = new_act(a.n, a.h // 16, a.w // 16, 1, device=a.buf.device)
    if a.bf16:
        assert b.bf16
        _abi.check(_abi.load().gpemsr_patch_cosine_bf16(a.ptr, b.ptr, a.n, a.h, a.w, a.c, out.ptr, _stream()), "patch_cosine_bf16")
        return out
    _abi.check(_abi.load().gpemsr_patch_cosine(a.ptr, b.ptr, a.n, a.h, a.w, a.c, out.ptr, _stream()), "patch_cosine")
    return out


def temporal_gate(aligned: Act, emb: Act, emb_ref: Act, b: int, t: int) -> Act:
    """aligned/emb: [b*t,h,w,c]; emb_ref: [b,h,w,c] -> af [b,h,w,t*c]."""
    assert aligned.ld == aligned.c and emb.ld == emb.c and emb_ref.ld == emb_ref.c
    if aligned.bf16:
        assert emb.bf16 and emb_ref.bf16
        af = new_act(b, aligned.h, aligned.w, t * aligned.c, device=aligned.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_temporal_gate_bf16(aligned.ptr, emb.ptr, emb_ref.ptr, b, t, aligned.h * aligned.w, aligned.c,
                                                         af.ptr, _stream()), "temporal_gate_bf16")
        return af
    af = new_act(b, aligned.h, aligned.w, t * aligned.c, device=aligned.buf.device)
    _abi.check(_abi.load().gpemsr_temporal_gate(aligned.ptr, emb.ptr, emb_ref.ptr, b, t, aligned.h * aligned.w, aligned.c,
                                                af.ptr, _stream()), "temporal_gate")
    return af


def frame_mix_lrelu(af: Act, t: int, m: torch.Tensor, bias: torch.Tensor) -> Act:
    assert af.ld == af.c and af.c % t == 0
    if af.bf16:
        out = new_act(af.n, af.h, af.w, af.c, device=af.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_frame_mix_lrelu_bf16(af.ptr, af.pixels, t, af.c // t, m.data_ptr(), bias.data_ptr(), out.ptr,
                                                           _stream()), "frame_mix_bf16")
        return out
    out = new_act(af.n, af.h, af.w, af.c, device=af.buf.device)
    _abi.check(_abi.load().gpemsr_frame_mix_lrelu(af.ptr, af.pixels, t, af.c // t, m.data_ptr(), bias.data_ptr(), out.ptr,
                                                  _stream()), "frame_mix")
    return out


def threeda_combine(feat: Act, attn: Act, attn_add: Act, f2: Act, f3: Act) -> Act:
    for a in (feat, attn, attn_add, f2, f3):
        assert a.ld == a.c
    if feat.bf16:
        assert all(a.bf16 for a in (attn, attn_add, f2, f3))
        out = new_act(feat.n, feat.h, feat.w, feat.c, device=feat.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_threeda_combine_bf16(feat.ptr, attn.ptr, attn_add.ptr, f2.ptr, f3.ptr, feat.pixels * feat.c,
                                                           out.ptr, _stream()), "threeda_combine_bf16")
        return out
    out = new_act(feat.n, feat.h, feat.w, feat.c, device=feat.buf.device)
    _abi.check(_abi.load().gpemsr_threeda_combine(feat.ptr, attn.ptr, attn_add.ptr, f2.ptr, f3.ptr, feat.pixels * feat.c,
                                                  out.ptr, _stream()), "threeda_combine")
    return out


def _u8_pass(out: "Act", out_u8: torch.Tensor):
    """tensor2img of a 1-channel fp32 result as its own pass (paths that do not end in the tap kernel)."""
    assert not out.bf16 and out.c == 1 and out.ld == 1 and out_u8.numel() == out.pixels
    _abi.check(_abi.load().gpemsr_tensor2img_u8(out.ptr, out.pixels, out_u8.data_ptr(), _stream()), "tensor2img")


def tensor2img_u8(x: torch.Tensor) -> torch.Tensor:
    """util/util.py:145-163 on device: clamp -> *255 -> round-half-even -> uint8 (same shape)."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _abi.check(_abi.load().gpemsr_tensor2img_u8(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "tensor2img")
    return out


def copy_channels(src: Act, dst: Act):
    assert src.pixels == dst.pixels and src.c == dst.c
    if src.bf16 or dst.bf16:
        assert src.bf16 and dst.bf16
        _abi.check(_abi.load().gpemsr_copy_channels_bf16(src.ptr, src.ld, dst.ptr, dst.ld, src.pixels, src.c, _stream()), "copy_channels_bf16")
        return
    _abi.check(_abi.load().gpemsr_copy_channels(src.ptr, src.ld, dst.ptr, dst.ld, src.pixels, src.c, _stream()),
               "copy_channels")


def maxpool2(x: Act) -> Act:
    """nn.MaxPool2d(2, 2), floor mode."""
    if x.bf16:
        out = new_act(x.n, x.h // 2, x.w // 2, x.c, device=x.buf.device, bf16=True)
        _abi.check(_abi.load().gpemsr_maxpool2_bf16(x.ptr, x.n, x.h, x.w, x.c, x.ld, out.ptr, out.ld, _stream()), "maxpool2_bf16")
        return out
    out = new_act(x.n, x.h // 2, x.w // 2, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_maxpool2(x.ptr, x.n, x.h, x.w, x.c, x.ld, out.ptr, out.ld, _stream()), "maxpool2")
    return out


def normalize3(x: Act, mean3, std3) -> Act:
    """(x - mean_c) / std_c on a 3-channel image (ContextualLoss.forward)."""
    assert x.c == 3
    out = new_act(x.n, x.h, x.w, 3, device=x.buf.device)
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    sd = (C.c_float * 3)(*[float(v) for v in std3])
    _abi.check(_abi.load().gpemsr_normalize3(x.ptr, x.pixels, x.ld, m, sd, out.ptr, out.ld, _stream()), "normalize3")
    return out


def cx_normalized_pair(x: Act, y: Act):
    """compute_cosine_distance's operands: (x - mean_y) and (y - mean_y), L2-normalised over channels per pixel."""
    assert x.c == y.c
    dev = x.buf.device
    nblk = (y.pixels + 1023) // 1024
    ws = torch.empty(nblk * y.c, dtype=torch.float32, device=dev)
    mu = torch.empty(y.c, dtype=torch.float32, device=dev)
    lib = _abi.load()
    _abi.check(lib.gpemsr_cx_channel_mean(y.ptr, y.pixels, y.c, y.ld, ws.data_ptr(), ws.numel(), mu.data_ptr(), _stream()),
               "cx_channel_mean")
    xn = new_act(x.n, x.h, x.w, x.c, device=dev)
    yn = new_act(y.n, y.h, y.w, y.c, device=dev)
    _abi.check(lib.gpemsr_cx_center_normalize(x.ptr, mu.data_ptr(), x.pixels, x.c, x.ld, xn.ptr, xn.ld, _stream()), "cx_center_normalize")
    _abi.check(lib.gpemsr_cx_center_normalize(y.ptr, mu.data_ptr(), y.pixels, y.c, y.ld, yn.ptr, yn.ld, _stream()), "cx_center_normalize")
    return xn, yn


def cx_from_similarity(sim: torch.Tensor, band_width: float):
    """sim [n, Px, Py] (cosine similarities) -> (loss [1], c [n, Py]) by model/contextual.py:41-52."""
    n, rows, cols = sim.shape
    dev = sim.device
    lib = _abi.load()
    cx = torch.empty_like(sim)
    _abi.check(lib.gpemsr_cx_rows(sim.data_ptr(), n * rows, cols, float(band_width), cx.data_ptr(), _stream()), "cx_rows")
    nslab = (rows + 127) // 128
    ws = torch.empty(2 * n * nslab * cols, dtype=torch.float32, device=dev)
    rmax = torch.empty(n, cols, dtype=torch.float32, device=dev)
    cw = torch.empty(n, cols, dtype=torch.float32, device=dev)
    cxn = torch.empty(n, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    _abi.check(lib.gpemsr_cx_reduce(cx.data_ptr(), sim.data_ptr(), n, rows, cols, float(band_width), ws.data_ptr(), ws.numel(),
                                    rmax.data_ptr(), cw.data_ptr(), cxn.data_ptr(), loss.data_ptr(), _stream()), "cx_reduce")
    return loss, cw, cxn


def split_pack_rows(a: Act) -> torch.Tensor:
    """Rows of a dense NHWC act ([n][h*w rows][c = K]) -> the split kernel's per-image B operand
    [n][plane (hi, lo)][K/16][k-half][rows][8] bf16 (attention: k for q.k^T, v^T for P.v)."""
    assert a.ld == a.c and a.off == 0 and a.c % 16 == 0
    rows = a.h * a.w
    out = torch.empty(a.n, 2, a.c // 16, 2, rows, 8, dtype=torch.bfloat16, device=a.buf.device)
    _abi.check(_abi.load().gpemsr_split_pack_rows(a.ptr, a.n, rows, a.c, a.ld, rows * a.ld, out.data_ptr(), _stream()),
               "split_pack_rows")
    return out


def gather_images(src: Act, idx: torch.Tensor) -> Act:
    """dst image j = src image idx[j] (idx: int32 device tensor; dense NHWC images, ld == c)."""
    assert src.ld == src.c and src.off == 0
    assert idx.dtype == torch.int32 and idx.is_cuda and idx.is_contiguous()
    n_dst = idx.numel()
    dst = new_act(n_dst, src.h, src.w, src.c, device=src.buf.device, bf16=src.bf16)
    units = src.h * src.w * src.c * src.esize // 4            # a raw copy in 4-byte units (bf16 images: c % 8 == 0)
    _abi.check(_abi.load().gpemsr_gather_images(src.ptr, idx.data_ptr(), dst.ptr, n_dst, units, _stream()),
               "gather_images")
    return dst


def copy_images(src: Act, n_dst: int, div: int, mul: int, add: int) -> Act:
    """dst image j = src image (j // div) * mul + add (dense NHWC images, ld == c)."""
    assert src.ld == src.c and src.off == 0
    dst = new_act(n_dst, src.h, src.w, src.c, device=src.buf.device, bf16=src.bf16)
    _abi.check(_abi.load().gpemsr_copy_images(src.ptr, dst.ptr, n_dst, src.h * src.w * src.c * src.esize // 4, div, mul, add, _stream()),
               "copy_images")
    return dst


# ----------------------------------------------------------------------------------------------------------------------
# Stage-3 training step: raw wrappers of the backward entry points (include/gpemsr_hip.h, "backward + optimizer").
# Gradient arguments named d* are ACCUMULATED into (zero-initialised) buffers; gpemsr_amd/train.py drives them.
# ----------------------------------------------------------------------------------------------------------------------
_WS = {}


def _workspace(floats: int, device) -> torch.Tensor:
    """One growing scratch buffer per device (stream-ordered reuse: every consumer finishes before the next launch)."""
    key = str(device)
    t = _WS.get(key)
    if t is None or t.numel() < floats:
        t = torch.empty(max(int(floats), 1 << 20), dtype=torch.float32, device=device)
        _WS[key] = t
    return t


def axpy(src: Act, dst: Act, alpha: float = 1.0):
    assert src.pixels == dst.pixels and src.c == dst.c
    _abi.check(_abi.load().gpemsr_axpy(src.ptr, src.ld, dst.ptr, dst.ld, src.pixels, src.c, float(alpha), _stream()), "axpy")


def mul_pix(x: Act, m: Act, out: Optional[Act] = None) -> Act:
    assert m.c == 1 and m.ld == 1 and m.pixels == x.pixels
    if out is None:
        out = new_act(x.n, x.h, x.w, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_mul_pix(x.ptr, x.ld, m.ptr, x.pixels, x.c, out.ptr, out.ld, _stream()), "mul_pix")
    return out


def mul_pix_bwd(dy: Act, x: Act, m: Act, dx: Optional[Act], dm: Optional[Act]):
    _abi.check(_abi.load().gpemsr_mul_pix_bwd(dy.ptr, dy.ld, x.ptr, x.ld, m.ptr, x.pixels, x.c, dx.ptr if dx is not None else None,
                                              dx.ld if dx is not None else 0, dm.ptr if dm is not None else None, _stream()),
               "mul_pix_bwd")


def act_bwd(dy: Act, y: Act, n: int, h: int, w: int, c: int, act: int, pixel_shuffle: bool) -> Act:
    dz = new_act(n, h, w, c, device=dy.buf.device)
    _abi.check(_abi.load().gpemsr_act_bwd(dy.ptr, dy.ld, y.ptr, y.ld, n, h, w, c, act, int(pixel_shuffle), dz.ptr, dz.ld, _stream()),
               "act_bwd")
    return dz


def bias_grad(dz: Act, db: torch.Tensor):
    ws = _workspace(512 * dz.c, dz.buf.device)
    _abi.check(_abi.load().gpemsr_bias_grad(dz.ptr, dz.pixels, dz.c, dz.ld, ws.data_ptr(), ws.numel(), db.data_ptr(), _stream()),
               "bias_grad")


def conv2d_wgrad(x: Act, dz: Act, ksize: int, stride: int, dw: torch.Tensor, cin_total: int, cin_off: int, tag: str = ""):
    """dw (OIHW [cout][cin_total][k][k], contiguous) += wgrad of a conv whose input slice is ``x`` and output gradient ``dz``."""
    lib = _abi.load()
    assert x.n == dz.n and dw.is_contiguous()
    need = lib.gpemsr_conv2d_wgrad_workspace(x.c, dz.c, ksize, dz.n, dz.h, dz.w)
    ws = _workspace(min(need, 1 << 28), x.buf.device)
    flops = 2.0 * dz.pixels * dz.c * x.c * ksize * ksize

    def _go():
        _abi.check(lib.gpemsr_conv2d_wgrad(x.ptr, x.ld, x.c, dz.ptr, dz.ld, dz.c, x.n, x.h, x.w, dz.h, dz.w, ksize, stride,
                                           ws.data_ptr(), ws.numel(), dw.data_ptr(), cin_total, cin_off, _stream()), "conv2d_wgrad")
    if PROFILER is not None:
        PROFILER.run("conv_wgrad", tag, flops, _go)
    else:
        _go()


def bilinear_bwd(dy: Act, dx: Act, align_corners: bool = False, mul: float = 1.0):
    _abi.check(_abi.load().gpemsr_bilinear_bwd(dy.ptr, dy.ld, dx.n, dx.h, dx.w, dx.c, dy.h, dy.w, int(align_corners), float(mul),
                                               dx.ptr, dx.ld, _stream()), "bilinear_bwd")


def dcn_columns_bwd(x: Act, om: Act, groups: int, dcol: Act, dx: Optional[Act], dom: Optional[Act], deterministic: bool = True):
    """torchvision deform_conv2d backward w.r.t. input / offsets / mask logits.  deterministic (default): the scatter into dx accumulates
    64-bit fixed-point integers (gpemsr_dcn_columns_bwd_det: bit-stable run to run); False: float atomics (the round-2 form, for A/B)."""
    assert dcol.ld == dcol.c == 9 * x.c
    if deterministic:
        absmax = dcol.buf.abs().amax().reshape(1)                  # device scalar, no host sync; amax is order-independent
        fix = torch.zeros(x.n * x.h * x.w * x.c, dtype=torch.int64, device=x.buf.device) if dx is not None else None
        _abi.check(_abi.load().gpemsr_dcn_columns_bwd_det(x.ptr, x.n, x.h, x.w, x.c, x.ld, om.ptr, om.ld, groups, dcol.ptr, absmax.data_ptr(),
                                                          fix.data_ptr() if fix is not None else None,
                                                          dx.ptr if dx is not None else None, dx.ld if dx is not None else 0,
                                                          dom.ptr if dom is not None else None, dom.ld if dom is not None else 0, _stream()),
                   "dcn_columns_bwd_det")
        return
    _abi.check(_abi.load().gpemsr_dcn_columns_bwd(x.ptr, x.n, x.h, x.w, x.c, x.ld, om.ptr, om.ld, groups, dcol.ptr,
                                                  dx.ptr if dx is not None else None, dx.ld if dx is not None else 0,
                                                  dom.ptr if dom is not None else None, dom.ld if dom is not None else 0, _stream()),
               "dcn_columns_bwd")


def temporal_gate_bwd(aligned: Act, emb: Act, emb_ref: Act, daf: Act, b: int, t: int, d_aligned: Act, d_emb: Act, d_emb_ref: Act):
    for a in (aligned, emb, emb_ref, daf, d_aligned, d_emb, d_emb_ref):
        assert a.ld == a.c
    _abi.check(_abi.load().gpemsr_temporal_gate_bwd(aligned.ptr, emb.ptr, emb_ref.ptr, daf.ptr, b, t, aligned.h * aligned.w, aligned.c,
                                                    d_aligned.ptr, d_emb.ptr, d_emb_ref.ptr, _stream()), "temporal_gate_bwd")


def frame_mix_lrelu_bwd(af: Act, out: Act, dout: Act, t: int, m: torch.Tensor, d_af: Act, dm: torch.Tensor, dbias: torch.Tensor):
    ws = _workspace(1024 * 32, af.buf.device)
    _abi.check(_abi.load().gpemsr_frame_mix_lrelu_bwd(af.ptr, out.ptr, dout.ptr, af.pixels, t, af.c // t, m.data_ptr(), d_af.ptr,
                                                      dm.data_ptr(), dbias.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "frame_mix_bwd")


def pool3s2_maxavg_bwd(x: Act, dy: Act, dx: Act):
    _abi.check(_abi.load().gpemsr_pool3s2_maxavg_bwd(x.ptr, x.n, x.h, x.w, x.c, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, _stream()),
               "pool3s2_bwd")


def threeda_combine_bwd(feat: Act, attn: Act, dout: Act, dfeat: Act, dattn: Act, dadd: Act, df2: Act, df3: Act):
    for a in (feat, attn, dout, dfeat, dattn, dadd, df2, df3):
        assert a.ld == a.c
    _abi.check(_abi.load().gpemsr_threeda_combine_bwd(feat.ptr, attn.ptr, dout.ptr, feat.pixels * feat.c, dfeat.ptr, dattn.ptr,
                                                      dadd.ptr, df2.ptr, df3.ptr, _stream()), "threeda_combine_bwd")


def maxpool2_bwd(x: Act, dy: Act, dx: Act):
    _abi.check(_abi.load().gpemsr_maxpool2_bwd(x.ptr, x.n, x.h, x.w, x.c, x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, _stream()), "maxpool2_bwd")


def scatter_add_images(dsrc: Act, idx: torch.Tensor, dtarget: Act):
    assert dsrc.ld == dsrc.c and dtarget.ld == dtarget.c and dsrc.off == 0 and dtarget.off == 0
    assert idx.dtype == torch.int32 and idx.numel() == dsrc.n
    _abi.check(_abi.load().gpemsr_scatter_add_images(dsrc.ptr, idx.data_ptr(), dtarget.ptr, dtarget.n, dsrc.n,
                                                     dsrc.h * dsrc.w * dsrc.c, _stream()), "scatter_add_images")


def l1_loss(sr: torch.Tensor, gt: torch.Tensor, grad_scale: float, dsr: Optional[torch.Tensor]) -> torch.Tensor:
    assert sr.numel() == gt.numel() and sr.is_contiguous() and gt.is_contiguous()
    ws = _workspace(1024, sr.device)
    loss = torch.empty(1, dtype=torch.float32, device=sr.device)
    _abi.check(_abi.load().gpemsr_l1_loss(sr.data_ptr(), gt.data_ptr(), sr.numel(), float(grad_scale),
                                          dsr.data_ptr() if dsr is not None else None, ws.data_ptr(), ws.numel(), loss.data_ptr(),
                                          _stream()), "l1_loss")
    return loss


def gray_normalize3(x: Act, mean3, std3) -> Act:
    assert x.c == 1 and x.ld == 1
    out = new_act(x.n, x.h, x.w, 3, device=x.buf.device)
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    sd = (C.c_float * 3)(*[float(v) for v in std3])
    _abi.check(_abi.load().gpemsr_gray_normalize3(x.ptr, x.pixels, m, sd, out.ptr, _stream()), "gray_normalize3")
    return out


def gray_normalize3_bwd(g: Act, std3, dx: Act):
    assert g.c == 3 and g.ld == 3 and dx.c == 1 and dx.ld == 1
    sd = (C.c_float * 3)(*[float(v) for v in std3])
    _abi.check(_abi.load().gpemsr_gray_normalize3_bwd(g.ptr, g.pixels, sd, dx.ptr, _stream()), "gray_normalize3_bwd")


def transpose_images(a: Act) -> Act:
    """[n][rows = h*w][cols = c] -> [n][c][h*w] as an Act with h*w channels."""
    assert a.ld == a.c and a.off == 0
    rows = a.h * a.w
    out = new_act(a.n, 1, a.c, rows, device=a.buf.device)
    _abi.check(_abi.load().gpemsr_transpose_images(a.ptr, out.ptr, a.n, rows, a.c, _stream()), "transpose_images")
    return out


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
              weight_decay: float, step: int):
    _abi.check(_abi.load().gpemsr_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
                                            float(beta2), float(eps), float(weight_decay), int(step), _stream()), "adam_step")


# ----------------------------------------------------------------------------------------------------------------------
# bf16 data path (precision = "bf16"): include/gpemsr_hip.h, "bf16 data path".  bf16 NHWC activations; 1-channel images,
# flows, deformable offsets and logits stay fp32.  The functions above dispatch here when they are handed bf16 Acts.
# ----------------------------------------------------------------------------------------------------------------------
def conv2d_bf16(srcs, pc: PackedConv, act: int = ACT_NONE, stride: int = 1, residual: Optional[Act] = None,
                pixmul: Optional[Act] = None, out: Optional[Act] = None, weight_image_stride: int = 0,
                src_image_stride: Optional[Sequence[int]] = None, force_mfma: bool = False, tag: str = "",
                out_f32: bool = False, out32: Optional[Act] = None, gn_stats: bool = False, kpack: bool = False,
                variant: int = 0, out_u8: Optional[torch.Tensor] = None, _u8_fused: Optional[list] = None,
                a_affine: Optional[tuple] = None, argmax: bool = False) -> Act:
    """Convolution of the bf16 path.  out_f32: fp32 result (logits input, deformable offsets, flows, 1-channel images);
    out32: additionally store the un-rounded fp32 result there; gn_stats: leave GroupNorm partial sums on the result
    (``out.gn``); kpack: store as the B operand [n][cout/8][pixels][8] of a later 1x1 product (returns the raw tensor);
    a_affine = (scale[n][cin], shift[n][cin], relu): the source is read as relu?(scale * x + shift) -- a folded GroupNorm
    apply (``groupnorm_scale_shift``); only where ``conv_affine_source_ok`` says so."""
    lib = _abi.load()
    s0 = srcs[0]
    n, h, w = s0.n, s0.h, s0.w
    k = pc.ksize
    dev = s0.buf.device
    if pc.transposed:
        oh, ow = 2 * h, 2 * w
    else:
        oh, ow = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
    OH, OW, oc = (2 * oh, 2 * ow, pc.cout // 4) if pc.pixel_shuffle else (oh, ow, pc.cout)
    plain = (not pc.transposed and not pc.pixel_shuffle and pixmul is None and weight_image_stride == 0 and src_image_stride is None
             and out32 is None and not gn_stats and not kpack and len(srcs) == 1)
    # 1 -> C stem: fp32 1-channel image in, bf16 out
    if (plain and not force_mfma and not s0.bf16 and s0.c == 1 and s0.ld == 1 and k == 3 and stride == 1 and pc.cout % 8 == 0 and pc.cout > 16
            and residual is None and not out_f32):
        if out is None:
            out = new_act(n, OH, OW, oc, device=dev, bf16=True)
        assert out.bf16 and (out.n, out.h, out.w, out.c) == (n, OH, OW, oc)

        def _go_stem():
            _abi.check(lib.gpemsr_conv2d_stem1_bf16(s0.ptr, n, h, w, pc.w.data_ptr(), pc.b.data_ptr() if pc.b is not None else None,
                                                    pc.cout, act, out.ptr, out.ld, _stream()), "conv2d_stem1_bf16")
        if PROFILER is not None:
            PROFILER.run("conv_stem1", tag, 2.0 * n * oh * ow * pc.cout * 9, _go_stem, name="conv_stem1_kernel<bf16>",
                         nbytes=4.0 * n * h * w + 2.0 * n * oh * ow * pc.cout)
        else:
            _go_stem()
        return out
    taps = 9.0 / 4.0 if pc.transposed else float(k * k)
    flops = 2.0 * n * oh * ow * pc.cout * (pc.algo_cin or pc.cin) * taps
    # SpyNet's flow update: 16 -> 2, 7x7, fp32 result + fp32 residual (csrc/tap_sum.hip, rowsum7_kernel)
    if (pc.wrow7 is not None and plain and not force_mfma and s0.bf16 and s0.c == 16 and pc.cout == 2 and k == 7 and stride == 1 and act == ACT_NONE
            and (residual is None or (not residual.bf16 and residual.c == 2))):
        if out is None:
            out = new_act(n, OH, OW, 2, device=dev)
        assert not out.bf16 and (out.n, out.h, out.w, out.c) == (n, OH, OW, 2)

        def _go_row7():
            _abi.check(lib.gpemsr_conv7_c16_cout2_bf16(s0.ptr, n, h, w, s0.ld, pc.wrow7.data_ptr(), pc.b.data_ptr() if pc.b is not None else None,
                                                       residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                                                       out.ptr, out.ld, _stream()), "conv7_c16_cout2_bf16")
        if PROFILER is not None:
            PROFILER.run("tap_sum", tag, flops, _go_row7, name="rowsum7_kernel<bf16>", nbytes=n * h * w * (2.0 * 16 + 4.0 * 2 + (8.0 if residual is not None else 0.0)))
        else:
            _go_row7()
        return out
    # SpyNet's 32 -> 16 7x7 layers: the 16x16x32 MFMA shape with resident weights (csrc/conv7_bf16.hip); variant 9 keeps the ring kernel
    if (pc.w7c16 is not None and plain and s0.bf16 and s0.c == 32 and pc.cout == 16 and k == 7 and stride == 1 and residual is None and not out_f32
            and act in (ACT_NONE, ACT_RELU, ACT_LRELU) and variant != 9 and s0.ld % 8 == 0 and s0.ptr % 16 == 0):
        if out is None:
            out = new_act(n, OH, OW, 16, device=dev, bf16=True)
        assert out.bf16 and (out.n, out.h, out.w, out.c) == (n, OH, OW, 16) and out.ld % 4 == 0 and out.ptr % 8 == 0

        def _go_c7():
            _abi.check(lib.gpemsr_conv7_c32_cout16_bf16(s0.ptr, n, h, w, s0.ld, pc.w7c16.data_ptr(), pc.b.data_ptr() if pc.b is not None else None, act,
                                                        out.ptr, out.ld, _stream()), "conv7_c32_cout16_bf16")
        if PROFILER is not None:
            PROFILER.run("conv_bf16", tag, flops, _go_c7, name="conv7_c32_cout16_kernel", nbytes=2.0 * (n * h * w * 48 + 49 * 512))
        else:
            _go_c7()
        return out
    # SpyNet's 8 -> 32 7x7 stems (input padded to a 16-channel tensor by spynet_prep_bf16): four taps per 16x16x32 MFMA; variant 9: ring kernel
    if (pc.w7c8 is not None and plain and s0.bf16 and s0.c == 16 and pc.cout == 32 and k == 7 and stride == 1 and residual is None and not out_f32
            and act in (ACT_NONE, ACT_RELU, ACT_LRELU) and variant != 9 and s0.ld % 8 == 0 and s0.ptr % 16 == 0):
        if out is None:
            out = new_act(n, OH, OW, 32, device=dev, bf16=True)
        assert out.bf16 and (out.n, out.h, out.w, out.c) == (n, OH, OW, 32) and out.ld % 4 == 0 and out.ptr % 8 == 0

        def _go_c8():
            _abi.check(lib.gpemsr_conv7_c8_cout32_bf16(s0.ptr, n, h, w, s0.ld, pc.w7c8.data_ptr(), pc.b.data_ptr() if pc.b is not None else None, act,
                                                       out.ptr, out.ld, _stream()), "conv7_c8_cout32_bf16")
        if PROFILER is not None:
            # (algorithmic work of the LAYER: 8 input channels; the ring kernel's table counted its zero-padded 16)
            PROFILER.run("conv_bf16", tag, 2.0 * n * h * w * 32 * 8 * 49, _go_c8, name="conv7_c8_cout32_kernel", nbytes=2.0 * (n * h * w * (8 + 32) + 49 * 256))
        else:
            _go_c8()
        return out
    # tiny channel counts: VALU kernel with fp32 packed weights (1-channel results are fp32 images)
    use_direct = (plain and not force_mfma and pc.cout <= 16 and (pc.cout <= 2 or pc.cin <= 16) and k >= 3 and pc.ck == 8
                  and (residual is None or (pc.cin == 64 and pc.cout == 1 and k == 3 and stride == 1 and not residual.bf16)))
    if use_direct:
        o32 = out_f32 or pc.cout <= 2
        if out is None:
            out = new_act(n, OH, OW, oc, device=dev, bf16=not o32)
        assert (out.n, out.h, out.w, out.c) == (n, OH, OW, oc) and out.bf16 == (not o32)

        def _go_taps():
            _abi.check(lib.gpemsr_conv_c64_cout1_bf16(s0.ptr, n, h, w, s0.ld, pc.wtap.data_ptr(), pc.b.data_ptr() if pc.b is not None else None,
                                                      act, residual.ptr if residual is not None else None,
                                                      residual.ld if residual is not None else 0, out.ptr, out.ld,
                                                      out_u8.data_ptr() if (out_u8 is not None and out.ld == 1) else None, _stream()),
                       "conv_c64_cout1_bf16")
        if pc.wtap is not None and s0.bf16 and pc.cin == 64 and pc.cout == 1 and k == 3 and stride == 1:
            if out_u8 is not None and out.ld == 1:
                _u8_fused[0] = True
            if PROFILER is not None:
                PROFILER.run("tap_sum", tag, flops, _go_taps, name="tap_sum_kernel<false,bf16>", nbytes=n * h * w * (2.0 * 64 + 4.0 + (4.0 if residual is not None else 0.0)))
            else:
                _go_taps()
            return out

        def _go_direct():
            _abi.check(lib.gpemsr_conv2d_direct_bf16(s0.ptr, int(not s0.bf16), n, h, w, s0.ld, s0.c, pc.w.data_ptr(),
                                                     pc.b.data_ptr() if pc.b is not None else None, pc.cout, k, stride, act,
                                                     residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                                                     out.ptr, int(o32), out.ld, _stream()), "conv2d_direct_bf16")
        if PROFILER is not None:
            PROFILER.run("conv_direct", tag, flops, _go_direct, name="conv_direct_kernel<bf16>",
                         nbytes=float(n * h * w * s0.c * s0.esize + n * OH * OW * oc * (4 if o32 else 2)))
        else:
            _go_direct()
        return out
    assert pc.wb is not None, f"{tag}: no bf16 weights packed for this layer"
    assert all(s.bf16 for s in srcs), f"{tag}: the bf16 MFMA kernel takes bf16 sources"
    d = _abi.ConvDesc16()
    d.n, d.h, d.w, d.nsrc = n, h, w, len(srcs)
    for i, s in enumerate(srcs):
        assert (s.n, s.h, s.w) == (n, h, w) or (src_image_stride is not None)
        d.src[i].ptr, d.src[i].ld, d.src[i].c = s.ptr, s.ld, s.c
        d.src_image_stride[i] = -1 if src_image_stride is None else int(src_image_stride[i])
    d.cout, d.ksize, d.stride, d.transposed = pc.cout, k, stride, int(pc.transposed)
    d.weight, d.weight_image_stride = pc.wb.data_ptr(), int(weight_image_stride)
    if pc.transposed and pc.wb.numel() == 16 * pc.cin * pc.cout + 9 * pc.cin * pc.cout:
        # a second form follows the staged one (packing.pack_convT_bf16): bit 0 = resident slabs (64 -> 64k), bit 1 = compact stage images
        d.weight_forms = 1 if (pc.cin == 64 and pc.cout % 64 == 0) else 2
    d.bias = pc.b.data_ptr() if pc.b is not None else None
    d.act = act
    if residual is not None:
        assert (residual.n, residual.h, residual.w, residual.c) == (n, OH, OW, oc)
        d.residual, d.res_ld, d.res_f32 = residual.ptr, residual.ld, int(not residual.bf16)
    if pixmul is not None:
        assert not pixmul.bf16 and pixmul.c == 1 and pixmul.ld == 1 and (pixmul.n, pixmul.h, pixmul.w) == (n, OH, OW)
        d.pixmul = pixmul.ptr
    d.pixel_shuffle = int(pc.pixel_shuffle)
    d.kpack = int(kpack)
    d.variant = int(variant)
    ret = None
    if argmax:
        # the product is not stored: int32 column of every row's maximum (gpemsr_conv16_desc.rowmax + gpemsr_rowmax_finish)
        assert out is None and k == 1 and not kpack and residual is None and pixmul is None and act == ACT_NONE and not gn_stats
        d.out, d.out_ld, d.out_f32 = srcs[0].ptr, pc.cout, 1          # (never written; the descriptor wants a valid aligned pointer)
        d.rowmax = 1                                                   # (placeholder: the geometry query below looks at it)
        parts = lib.gpemsr_conv2d_bf16_rowmax_parts(C.byref(d))
        if parts < 1:
            _abi.check(parts, "conv2d_bf16_rowmax_parts")
        rows = n * OH * OW
        ws = torch.empty(rows * parts * 2, dtype=torch.float32, device=dev)
        d.rowmax = ws.data_ptr()
        idx = torch.empty(rows, dtype=torch.int32, device=dev)

        def _go_rm():
            _abi.check(lib.gpemsr_conv2d_bf16(C.byref(d), _stream()), "conv2d_bf16")
            _abi.check(lib.gpemsr_rowmax_finish(ws.data_ptr(), rows, parts, idx.data_ptr(), _stream()), "rowmax_finish")
        if PROFILER is not None:
            nm = _kernel_name(lib.gpemsr_conv2d_bf16_kernel_name, d, ("bf16rowmax", n, h, w, tuple((s_.c, s_.ld) for s_ in srcs), pc.cout))
            nb = _layer_bytes(srcs, src_image_stride, rows, 4, int(pc.cout * pc.cin), 2, None, None)
            PROFILER.run("conv_bf16", tag, flops, _go_rm, name=nm, nbytes=nb)
        else:
            _go_rm()
        return idx
    if kpack:
        assert out is None and not out_f32
        ret = torch.empty(n, oc // 8, OH * OW, 8, dtype=torch.bfloat16, device=dev)
        d.out, d.out_ld, d.out_f32 = ret.data_ptr(), 8, 0
    else:
        if out is None:
            out = new_act(n, OH, OW, oc, device=dev, bf16=not out_f32)
        assert (out.n, out.h, out.w, out.c) == (n, OH, OW, oc), f"out geometry {(out.n, out.h, out.w, out.c)} != {(n, OH, OW, oc)}"
        assert out.bf16 == (not out_f32), f"{tag}: output format mismatch"
        d.out, d.out_ld, d.out_f32 = out.ptr, out.ld, int(out_f32)
        ret = out
    if out32 is not None:
        assert not out32.bf16 and (out32.n, out32.h, out32.w, out32.c) == (n, OH, OW, oc)
        d.out32, d.out32_ld = out32.ptr, out32.ld
    if a_affine is not None:
        sc, sh, relu = a_affine
        assert sc.dtype == torch.float32 and sh.dtype == torch.float32 and sc.numel() == n * srcs[0].c == sh.numel() and len(srcs) == 1
        d.a_scale, d.a_shift, d.a_relu = sc.data_ptr(), sh.data_ptr(), int(relu)
    if gn_stats:
        parts = lib.gpemsr_conv2d_bf16_gn_parts(C.byref(d))
        if parts < 1:
            _abi.check(parts, "conv2d_bf16_gn_parts")
        ws = torch.empty(n * parts * pc.cout * 2, dtype=torch.float32, device=dev)
        d.gn_partials = ws.data_ptr()
        d.gn_cpg = pc.cout // 32 if pc.cout % 32 == 0 else 1       # GroupNorm(32 groups) everywhere in the model (model/blocks.py:5-6)
        out.gn = (ws, parts)

    def _go():
        _abi.check(lib.gpemsr_conv2d_bf16(C.byref(d), _stream()), "conv2d_bf16")
    if PROFILER is not None:
        nm = _kernel_name(lib.gpemsr_conv2d_bf16_kernel_name, d, ("bf16", n, h, w, tuple((s_.c, s_.ld) for s_ in srcs), pc.cout, k, stride, int(pc.transposed),
                                                                  int(pc.pixel_shuffle), int(kpack), int(out_f32), out32 is not None, gn_stats, variant,
                                                                  a_affine is not None, residual is not None and residual.bf16, pixmul is not None, act,
                                                                  weight_image_stride != 0, None if src_image_stride is None else tuple(src_image_stride)))
        wimgs = n if weight_image_stride != 0 else 1
        nb = _layer_bytes(srcs, src_image_stride, n * OH * OW * oc, 4 if out_f32 else 2, int(wimgs * pc.cout * pc.cin * taps), 2, residual, pixmul)
        if out32 is not None:
            nb += 4.0 * n * OH * OW * oc
        PROFILER.run("conv_bf16", tag, flops, _go, name=nm, nbytes=nb)
    else:
        _go()
    return ret


def conv_affine_source_ok(x: Act, pc: PackedConv) -> bool:
    """True when gpemsr_conv2d_bf16 has a kernel that applies a per-(image, channel) affine map + ReLU to its source while
    staging it (3x3, stride 1, one dense bf16 source of 64 or k*32 channels, plain bf16 store)."""
    if not (x.bf16 and pc.wb is not None and pc.ksize == 3 and not pc.transposed and not pc.pixel_shuffle and x.c % 32 == 0 and x.c >= 64):
        return False
    d = _abi.ConvDesc16()
    d.n, d.h, d.w, d.nsrc = x.n, x.h, x.w, 1
    d.src[0].ptr, d.src[0].ld, d.src[0].c = x.ptr, x.ld, x.c
    d.src_image_stride[0] = -1
    d.cout, d.ksize, d.stride, d.transposed = pc.cout, 3, 1, 0
    d.weight = pc.wb.data_ptr()
    d.bias = pc.b.data_ptr() if pc.b is not None else None
    d.out, d.out_ld = x.ptr, pc.cout                     # (geometry only: nothing is launched)
    return _abi.load().gpemsr_conv2d_bf16_axf_ok(C.byref(d)) == 1


def groupnorm_scale_shift(x: Act, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-6):
    """GroupNorm(groups, eps, affine) of the tensor ``x`` (bf16, or fp32 on the exact path) as per-(image, channel) scale / shift tables
    [n][c] (fp32): normalised = scale * x + shift.  Statistics from the producing convolution's epilogue (``x.gn``) or one pass over x."""
    lib = _abi.load()
    _require_gpu(x)
    hw, dev = x.h * x.w, x.buf.device
    if not x.bf16:
        parts = max(1, min(64, hw // 64))
        ws = torch.empty(x.n * parts * x.c * 2 + x.n * groups * 2, dtype=torch.float32, device=dev)
        mr = ws[x.n * parts * x.c * 2:]
        if x.gn is not None:
            gws, gparts = x.gn
            _abi.check(lib.gpemsr_groupnorm_finish(gws.data_ptr(), x.n, hw, x.c, groups, gparts, eps, mr.data_ptr(), _stream()), "groupnorm_finish")
        else:
            _abi.check(lib.gpemsr_groupnorm_stats(x.ptr, x.n, hw, x.c, x.ld, groups, eps, ws.data_ptr(), parts, mr.data_ptr(), _stream()), "groupnorm_stats")
        ss = torch.empty(2, x.n * x.c, dtype=torch.float32, device=dev)
        _abi.check(lib.gpemsr_groupnorm_scale_shift(mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.n, x.c, groups, ss[0].data_ptr(), ss[1].data_ptr(),
                                                    _stream()), "groupnorm_scale_shift")
        x.gn = None
        return ss[0], ss[1]
    if x.gn is not None:
        ws, parts = x.gn
    else:
        parts = max(1, min(64, hw // 64))
        ws = torch.empty(x.n * parts * x.c * 2, dtype=torch.float32, device=dev)
        _abi.check(lib.gpemsr_groupnorm_stats_bf16(x.ptr, x.n, hw, x.c, x.ld, ws.data_ptr(), parts, _stream()), "groupnorm_stats_bf16")
    mr = torch.empty(x.n * groups * 2, dtype=torch.float32, device=dev)
    _abi.check(lib.gpemsr_groupnorm_finish(ws.data_ptr(), x.n, hw, x.c, groups, parts, eps, mr.data_ptr(), _stream()), "groupnorm_finish")
    ss = torch.empty(2, x.n * x.c, dtype=torch.float32, device=dev)
    _abi.check(lib.gpemsr_groupnorm_scale_shift(mr.data_ptr(), gamma.data_ptr(), beta.data_ptr(), x.n, x.c, groups, ss[0].data_ptr(), ss[1].data_ptr(),
                                                _stream()), "groupnorm_scale_shift")
    x.gn = None
    return ss[0], ss[1]


def upconv_out_bf16(x: Act, frag: torch.Tensor, consts: torch.Tensor, out: Optional[Act] = None, tag: str = "") -> Act:
    """ConvTranspose2d(64 -> 64, k3 s2 p1 op1) + Conv2d(64 -> 1, 3x3) as one operator (csrc/tap_sum.hip; packing.pack_upconv_out):
    bf16 x [n][h][w][64] -> fp32 image [n][2h][2w]."""
    _require_gpu(x)
    assert x.bf16 and x.c == 64
    if out is None:
        out = new_act(x.n, 2 * x.h, 2 * x.w, 1, device=x.buf.device)
    assert not out.bf16 and (out.n, out.h, out.w, out.c) == (x.n, 2 * x.h, 2 * x.w, 1)

    def _go():
        _abi.check(_abi.load().gpemsr_upconv_out_c64_bf16(x.ptr, x.n, x.h, x.w, x.ld, frag.data_ptr(), consts.data_ptr(), out.ptr, out.ld,
                                                          _stream()), "upconv_out_c64_bf16")
    if PROFILER is not None:
        # the arithmetic of the layered form it replaces: 2.25 taps x 64 x 64 for the up-block + 9 x 64 for the output conv
        PROFILER.run("tap_sum", tag, 2.0 * x.n * 4 * x.h * x.w * (64 * 64 * 2.25 + 64 * 9), _go, name="tap_sum_kernel<true,bf16>",
                     nbytes=x.n * x.h * x.w * (2.0 * 64 + 4.0 * 4))
    else:
        _go()
    return out


def upconv_out_f32(x: Act, frag: torch.Tensor, consts: torch.Tensor, out: Optional[Act] = None, tag: str = "") -> Act:
    """upconv_out_bf16 for fp32 activations (the exact-fp32 path): v_mfma_f32_32x32x2_f32, fp32 composed weights."""
    _require_gpu(x)
    assert not x.bf16 and x.c == 64 and x.ld % 4 == 0
    if out is None:
        out = new_act(x.n, 2 * x.h, 2 * x.w, 1, device=x.buf.device)
    assert (out.n, out.h, out.w, out.c) == (x.n, 2 * x.h, 2 * x.w, 1)

    def _go():
        _abi.check(_abi.load().gpemsr_upconv_out_c64_f32(x.ptr, x.n, x.h, x.w, x.ld, frag.data_ptr(), consts.data_ptr(), out.ptr, out.ld,
                                                         _stream()), "upconv_out_c64_f32")
    if PROFILER is not None:
        PROFILER.run("tap_sum", tag, 2.0 * x.n * 4 * x.h * x.w * (64 * 64 * 2.25 + 64 * 9), _go, name="tap_sum_kernel<true,float>",
                     nbytes=x.n * x.h * x.w * (4.0 * 64 + 4.0 * 4))
    else:
        _go()
    return out


def cast_bf16(x: Act) -> Act:
    out = new_act(x.n, x.h, x.w, x.c, device=x.buf.device, bf16=True)
    _abi.check(_abi.load().gpemsr_cast_f32_bf16(x.ptr, x.pixels, x.c, x.ld, out.ptr, out.ld, _stream()), "cast_f32_bf16")
    return out


def cast_f32(x: Act) -> Act:
    out = new_act(x.n, x.h, x.w, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_cast_bf16_f32(x.ptr, x.pixels, x.c, x.ld, out.ptr, out.ld, _stream()), "cast_bf16_f32")
    return out


def split_hi_lo_bf16(x: Act):
    """fp32 activation -> (hi, lo) bf16 activations with x = hi + lo to 2^-17 relative (gpemsr_split_f32_bf16x2)."""
    assert not x.bf16
    hi = new_act(x.n, x.h, x.w, x.c, device=x.buf.device, bf16=True)
    lo = new_act(x.n, x.h, x.w, x.c, device=x.buf.device, bf16=True)
    _abi.check(_abi.load().gpemsr_split_f32_bf16x2(x.ptr, x.pixels, x.c, x.ld, hi.ptr, hi.ld, lo.ptr, lo.ld, _stream()), "split_f32_bf16x2")
    return hi, lo


def pack_rows_bf16(a: Act, perm16: bool = False) -> torch.Tensor:
    """bf16 rows [n][h*w][c] -> [n][c/8][h*w][8]: the per-image B operand of a 1x1 product (attention).  perm16: the rows of every
    16-group in the order 0-3, 8-11, 4-7, 12-15 (what ``flash_attention_bf16`` wants of v^T's columns)."""
    assert a.bf16 and a.c % 8 == 0
    rows = a.h * a.w
    out = torch.empty(a.n, a.c // 8, rows, 8, dtype=torch.bfloat16, device=a.buf.device)
    _abi.check(_abi.load().gpemsr_pack_rows_bf16_ex(a.ptr, a.n, rows, a.c, a.ld, rows * a.ld, out.data_ptr(), int(perm16), _stream()), "pack_rows_bf16")
    return out


def flash_attention_ok(tokens: int, channels: int) -> bool:
    return channels == 512 and tokens % 128 == 0 and tokens >= 128


def flash_attention_bf16(q: Act, kp: torch.Tensor, vtp: torch.Tensor, bias_v: Optional[torch.Tensor], out: Optional[Act] = None, tag: str = "") -> Act:
    """softmax(q k^T) v without the score matrix in memory (csrc/attn_bf16.hip).  q [n][T][C] bf16 Act (scale folded in), kp [n][C/8][T][8]
    (a ``kpack`` convolution result), vtp [n][T/8][C][8] with perm16 key order (the v^T product over ``pack_rows_bf16(hn, perm16=True)``)."""
    _require_gpu(q)
    n, T, c = q.n, q.h * q.w, q.c
    assert q.bf16 and flash_attention_ok(T, c) and tuple(kp.shape) == (n, c // 8, T, 8) and tuple(vtp.shape) == (n, T // 8, c, 8)
    if out is None:
        out = new_act(n, q.h, q.w, c, device=q.buf.device, bf16=True)
    assert out.bf16 and (out.n, out.h * out.w, out.c) == (n, T, c)

    def _go():
        _abi.check(_abi.load().gpemsr_flash_attention_bf16(q.ptr, q.ld, kp.data_ptr(), vtp.data_ptr(), bias_v.data_ptr() if bias_v is not None else None,
                                                           n, T, c, out.ptr, out.ld, _stream()), "flash_attention_bf16")
    if PROFILER is not None:
        PROFILER.run("conv_bf16", tag, 4.0 * n * T * T * c, _go, name="flash_attn512_kernel",     # q.k^T and P.v: 2 x (2 T^2 C) FLOPs per image
                     nbytes=2.0 * 4 * n * T * c)                                                  # q, k, v^T read, the result written (bf16)
    else:
        _go()
    return out


def softmax_rows_bf16(s: Act) -> Act:
    """Row softmax of a score tensor [n][rows][cols] (fp32 or bf16) -> P bf16 (in place when S is bf16)."""
    rows, cols = s.pixels, s.c
    out = s if s.bf16 else new_act(s.n, s.h, s.w, s.c, device=s.buf.device, bf16=True)
    _abi.check(_abi.load().gpemsr_softmax_rows_bf16(s.ptr, int(not s.bf16), rows, cols, s.ld, out.ptr, out.ld, _stream()), "softmax_rows_bf16")
    return out


def gather_rows_bf16(table: torch.Tensor, idx: torch.Tensor, n: int, h: int, w: int) -> Act:
    dim = table.shape[1]
    out = new_act(n, h, w, dim, device=table.device, bf16=True)
    _abi.check(_abi.load().gpemsr_gather_rows_bf16(table.data_ptr(), dim, idx.data_ptr(), n * h * w, out.ptr, out.ld, _stream()),
               "gather_rows_bf16")
    return out


def copy_channels_f32_bf16(src: Act, dst: Act):
    assert not src.bf16 and dst.bf16 and src.pixels == dst.pixels and src.c == dst.c
    _abi.check(_abi.load().gpemsr_copy_channels_f32_bf16(src.ptr, src.ld, dst.ptr, dst.ld, src.pixels, src.c, _stream()),
               "copy_channels_f32_bf16")


def spynet_prep_bf16(ref: Act, supp: Act, flow_coarse: Optional[Act], mean3, std3):
    assert ref.c == 1 and ref.ld == 1 and supp.c == 1 and supp.ld == 1 and not ref.bf16
    up = new_act(ref.n, ref.h, ref.w, 2, device=ref.buf.device)
    inp = new_act(ref.n, ref.h, ref.w, 16, device=ref.buf.device, bf16=True)
    m = (C.c_float * 3)(*[float(v) for v in mean3])
    s = (C.c_float * 3)(*[float(v) for v in std3])
    if flow_coarse is not None:
        assert flow_coarse.ld == 2 and not flow_coarse.bf16 and flow_coarse.h == ref.h // 2 and flow_coarse.w == ref.w // 2
    _abi.check(_abi.load().gpemsr_spynet_prep_bf16(ref.ptr, supp.ptr, flow_coarse.ptr if flow_coarse is not None else None,
                                                   ref.n, ref.h, ref.w, m, s, up.ptr, inp.ptr, _stream()), "spynet_prep_bf16")
    return up, inp


def vgg_mask_bf16(ref_img: Act, lr: Act, scale: int, w1: torch.Tensor, b1: torch.Tensor, w2b: torch.Tensor, b2: torch.Tensor, tag: str = "vgg_mask") -> Act:
    """model/GPEMSR.py:385-395 fused: VGG relu1_2 of the prior image and of the bilinearly up-sampled LR slice + 16x16 patch
    cosine -> [n, sH/16, sW/16, 1] fp32; no feature map in HBM."""
    assert not ref_img.bf16 and not lr.bf16 and ref_img.c == 1 and lr.c == 1 and ref_img.ld == 1 and lr.ld == 1
    n, h, w = lr.n, lr.h, lr.w
    assert (ref_img.n, ref_img.h, ref_img.w) == (n, h * scale, w * scale)
    out = new_act(n, h * scale // 16, w * scale // 16, 1, device=lr.buf.device)

    def _go():
        _abi.check(_abi.load().gpemsr_vgg_mask_bf16(ref_img.ptr, lr.ptr, n, h, w, scale, w1.data_ptr(), b1.data_ptr(), w2b.data_ptr(), b2.data_ptr(),
                                                    out.ptr, _stream()), "vgg_mask_bf16")
    flops = 2.0 * 2 * n * (h * scale) * (w * scale) * 64 * (64 * 9 + 9)          # both images: conv1_2 + conv1_1
    if PROFILER is not None:
        PROFILER.run("vgg_mask", tag, flops, _go, name="vgg_mask2_kernel",
                     nbytes=4.0 * n * (h * scale * w * scale + h * w) + 2.0 * 64 * 64 * 9)        # the two 1-channel images + W2; the cosine map is tiny
    else:
        _go()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# adversarial phase of stage-1 training (csrc/stage1_adv.hip): PatchGAN discriminator pieces, fp32 NHWC
# ----------------------------------------------------------------------------------------------------------------------
def im2col4(x: Act, stride: int, kp: int) -> Act:
    """Conv2d(k4, stride, padding 0) columns: [n][oh][ow][kp], k = (ky*4 + kx)*c + ci, zero padded to kp."""
    assert not x.bf16
    oh, ow = (x.h - 4) // stride + 1, (x.w - 4) // stride + 1
    col = new_act(x.n, oh, ow, kp, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_im2col4(x.ptr, x.n, x.h, x.w, x.c, x.ld, stride, col.ptr, kp, _stream()), "im2col4")
    return col


def col2im4(dcol: Act, h: int, w: int, c: int, stride: int, dx: Act, accumulate: bool):
    assert (dx.n, dx.h, dx.w, dx.c) == (dcol.n, h, w, c) and dcol.ld == dcol.c
    _abi.check(_abi.load().gpemsr_col2im4(dcol.ptr, dcol.n, h, w, c, stride, dcol.c, dx.ptr, dx.ld, int(accumulate), _stream()), "col2im4")


def lrelu_slope(x: Act, slope: float) -> Act:
    assert x.ld == x.c
    y = new_act(x.n, x.h, x.w, x.c, device=x.buf.device)
    _abi.check(_abi.load().gpemsr_lrelu_slope(x.ptr, x.pixels * x.c, float(slope), y.ptr, _stream()), "lrelu_slope")
    return y


def lrelu_slope_bwd(dy: Act, y: Act, slope: float, dx: Optional[Act] = None, accumulate: bool = False) -> Act:
    assert dy.ld == dy.c and y.ld == y.c
    if dx is None:
        dx = new_act(y.n, y.h, y.w, y.c, device=y.buf.device)
    _abi.check(_abi.load().gpemsr_lrelu_slope_bwd(dy.ptr, y.ptr, y.pixels * y.c, float(slope), dx.ptr, int(accumulate), _stream()), "lrelu_slope_bwd")
    return dx


def sum_scaled(x: torch.Tensor, scale: float, square: bool = False, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """out[0] (+)= scale * sum(x) or scale * sum(x^2) (one workgroup, fixed order)."""
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    _abi.check(_abi.load().gpemsr_sum_scaled(x.data_ptr(), x.numel(), float(scale), int(square), out.data_ptr(), int(accumulate), _stream()), "sum_scaled")
    return out


def instnorm(x: Act, eps: float = 1e-5):
    """nn.InstanceNorm2d(c) (no affine): -> (normalised Act, mean_rstd [n][c][2])."""
    lib = _abi.load()
    assert not x.bf16 and x.ld == x.c
    hw, c, dev = x.h * x.w, x.c, x.buf.device
    parts = max(1, min(64, hw // 64))
    ws = torch.empty(x.n * parts * c * 2, dtype=torch.float32, device=dev)
    mr = torch.empty(x.n * c * 2, dtype=torch.float32, device=dev)
    _abi.check(lib.gpemsr_groupnorm_stats(x.ptr, x.n, hw, c, x.ld, c, float(eps), ws.data_ptr(), parts, mr.data_ptr(), _stream()), "groupnorm_stats")
    one, zero = _const_vec(c, 1.0, dev), _const_vec(c, 0.0, dev)
    out = new_act(x.n, x.h, x.w, c, device=dev)
    _abi.check(lib.gpemsr_groupnorm_apply(x.ptr, x.n, hw, c, x.ld, c, mr.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, None, 0, out.ptr, out.ld, _stream()),
               "groupnorm_apply")
    return out, mr


def instnorm_bwd(x: Act, mr: torch.Tensor, dy: Act) -> Act:
    """dx of nn.InstanceNorm2d given its input, forward statistics and dy."""
    lib = _abi.load()
    hw, c, dev = x.h * x.w, x.c, x.buf.device
    parts = max(1, min(64, hw // 64))
    w2 = _workspace(x.n * parts * c * 2 + x.n * c * 2 + x.n * c * 2, dev)
    dx = new_act(x.n, x.h, x.w, c, device=dev, zero=True)
    one, zero = _const_vec(c, 1.0, dev), _const_vec(c, 0.0, dev)
    _abi.check(lib.gpemsr_groupnorm_bwd(x.ptr, x.ld, dy.ptr, dy.ld, x.n, hw, c, c, mr.data_ptr(), one.data_ptr(), zero.data_ptr(), 0, w2.data_ptr(), w2.numel(),
                                        dx.ptr, dx.ld, None, None, _stream()), "groupnorm_bwd")
    return dx


def instnorm_bwd_bwd(x: Act, mr: torch.Tensor, dy: Act, g: Act, gx: Act, accumulate_gx: bool) -> Act:
    """Second-order terms of nn.InstanceNorm2d: returns gdy = dL/d(dy); gx (+)= dL/dx (csrc/stage1_adv.hip)."""
    gdy = new_act(x.n, x.h, x.w, x.c, device=x.buf.device)
    assert x.ld == x.c and dy.ld == x.c and g.ld == x.c and gx.ld == x.c
    _abi.check(_abi.load().gpemsr_instnorm_bwd_bwd(x.ptr, dy.ptr, g.ptr, mr.data_ptr(), x.n, x.h * x.w, x.c, gx.ptr, gdy.ptr, int(accumulate_gx), _stream()),
               "instnorm_bwd_bwd")
    return gdy


_CONST_VECS = {}


def _const_vec(c: int, v: float, dev) -> torch.Tensor:
    key = (c, v, str(dev))
    if key not in _CONST_VECS:
        _CONST_VECS[key] = torch.full((c,), v, dtype=torch.float32, device=dev)
    return _CONST_VECS[key]
