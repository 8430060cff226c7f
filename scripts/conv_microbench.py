"""Dev helper (GPU box): HIP-event timing of gpemsr_conv2d on the layer shapes that dominate the forward.
usage: python scripts/conv_microbench.py [reps]   (GPEMSR_LIB_PATH selects an alternative .so build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpemsr_amd import ops
from gpemsr_amd.packing import pack_conv, pack_convT

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else None
precision = sys.argv[3] if len(sys.argv) > 3 else "fp32"
SHAPES = [  # name, n, cin, cout, k, stride, h, w, kind, residual
    ("rb64_128", 80, 64, 64, 3, 1, 128, 128, "conv", True),
    ("rb64_512", 20, 64, 64, 3, 1, 512, 512, "conv", True),
    ("vgg_1024", 6, 64, 64, 3, 1, 1024, 1024, "conv", False),
    ("vq256_128", 20, 256, 256, 3, 1, 128, 128, "conv", False),
    ("vq512_64", 20, 512, 512, 3, 1, 64, 64, "conv", False),
    ("vq128_256", 20, 128, 128, 3, 1, 256, 256, "conv", False),
    ("fuse128_512", 20, 128, 64, 3, 1, 512, 512, "conv", False),
    ("spy32_64_512", 20, 32, 64, 7, 1, 512, 512, "conv", False),
    ("pw512_64", 20, 512, 512, 1, 1, 64, 64, "conv", True),
    ("lin1024", 20, 512, 1024, 1, 1, 64, 64, "conv", False),
    ("up256_ps", 16, 64, 256, 3, 1, 512, 512, "ps", False),
    ("convT64_512", 20, 64, 64, 3, 1, 512, 512, "convT", False),
    ("down_s2", 20, 64, 64, 3, 2, 512, 512, "conv", False),
    ("attn_qk", 4, 512, 4096, 1, 1, 128, 32, "bmm", False),
]
print(f"{'shape':14s} {'ms':>8s} {'TFLOP/s':>8s}")
for name, n, cin, cout, k, stride, h, w, kind, use_res in SHAPES:
    if only and name not in only:
        continue
    x = ops.from_nhwc(torch.randn(n, h, w, cin, device=dev))
    if kind == "convT":
        pc = pack_convT(torch.randn(cin, cout, 3, 3) * 0.05, torch.randn(cout), dev)
        if precision != "fp32":
            from gpemsr_amd.packing import pack_convT_split
            pc.w16 = pack_convT_split(pc, dev)
        flops = 2.0 * n * h * w * cin * cout * 9
    elif kind == "bmm":
        wt = torch.randn(n, cout, cin, device=dev)
        pc = ops.PackedConv(wt, None, 1, cout, (cin,), 32)
        if precision != "fp32":
            pc.w16 = ops.split_pack_rows(ops.Act(wt, n, cout // 32, 32, cin, cin, 0))
        flops = 2.0 * n * h * w * cin * cout
    else:
        wfull = torch.randn(cout, cin, k, k) * 0.05
        pc = pack_conv(wfull, torch.randn(cout), dev, pixel_shuffle=(kind == "ps"))
        if precision != "fp32" and (k in (3, 7) and cin % 16 == 0 or k == 1 and cin % 32 == 0) and stride == 1:
            from gpemsr_amd.packing import pack_conv_split
            pc.w16 = pack_conv_split(pc, wfull, dev, pixel_shuffle=(kind == "ps"))
        flops = 2.0 * n * (h // stride) * (w // stride) * cin * cout * k * k
    kw = dict(stride=stride, precision=precision)
    if kind == "bmm":
        kw["weight_image_stride"] = cout * cin
    out = ops.conv2d([x], pc, ops.ACT_LRELU, **kw)
    res = ops.from_nhwc(torch.randn(out.n, out.h, out.w, out.c, device=dev)) if use_res else None
    for _ in range(2):
        ops.conv2d([x], pc, ops.ACT_LRELU, residual=res, out=out, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.conv2d([x], pc, ops.ACT_LRELU, residual=res, out=out, **kw)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{name:14s} {ms:8.3f} {flops / ms / 1e9:8.1f}", flush=True)
    del x, out, res
