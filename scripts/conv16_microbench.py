#!/usr/bin/env python3
"""Per-shape micro-benchmark of gpemsr_conv2d_bf16 (bf16 activations) on the layer shapes of the 8x forward at batch 16,
interleaved with round 1's conv_split kernel (fp32 activations, bf16 operands) on the same shapes and with the tile
variants of the new kernel -- all in ONE process, rounds interleaved (guide rule 24), random operands (rule 25).
    python3 scripts/conv16_microbench.py [--rounds 5] [--only substr]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import ops  # noqa: E402
from gpemsr_amd.packing import pack_conv, pack_conv_bf16, pack_conv_split, pack_convT, pack_convT_bf16, pack_convT_split  # noqa: E402

SHAPES = [
    # name, kind, n, cin, cout, k, stride, h, w, variants
    ("fe 64->64 3x3 @128^2 x80", "conv", 80, 64, 64, 3, 1, 128, 128, (0, 6)),
    ("dec 64->64 3x3 @512^2 x80", "conv", 80, 64, 64, 3, 1, 512, 512, (0, 6)),
    ("recon 64->64 3x3 @128^2 x16", "conv", 16, 64, 64, 3, 1, 128, 128, (0, 6)),
    ("mpf 64->64 3x3 @512^2 x8", "conv", 8, 64, 64, 3, 1, 512, 512, (0, 6)),
    ("vgg/HR 64->64 3x3 @1024^2 x4", "conv", 4, 64, 64, 3, 1, 1024, 1024, (0, 6)),
    ("dec 128->128 3x3 @256^2 x16", "conv", 16, 128, 128, 3, 1, 256, 256, (0, 5)),
    ("vq 256->256 3x3 @128^2 x80", "conv", 80, 256, 256, 3, 1, 128, 128, (0, 5)),
    ("vq 512->512 3x3 @64^2 x80", "conv", 80, 512, 512, 3, 1, 64, 64, (0, 5)),
    ("mpf 128->64 3x3 @512^2 x8", "conv", 8, 128, 64, 3, 1, 512, 512, (0, 2)),
    ("up 64->256 3x3+ps @512^2 x4", "ps", 4, 64, 256, 3, 1, 512, 512, (0, 6)),
    ("1x1 512->512 @64^2 x80", "conv", 80, 512, 512, 1, 1, 64, 64, (0, 1)),
    ("1x1 320->64 @128^2 x16", "conv", 16, 320, 64, 1, 1, 128, 128, (0, 1)),
    ("down 256->512 3x3 s2 @128^2 x80", "conv", 80, 256, 512, 3, 2, 128, 128, (0, 12)),
    ("down 64->64 3x3 s2 @512^2 x16", "conv", 16, 64, 64, 3, 2, 512, 512, (0, 2)),
    ("down 64->64 3x3 s2 @512^2 x80", "conv", 80, 64, 64, 3, 2, 512, 512, (0, 2)),
    ("down 128->64 3x3 s2 @256^2 x80", "conv", 80, 128, 64, 3, 2, 256, 256, (0, 2, 12)),
    ("convT 64->64 @512^2 x8", "convT", 8, 64, 64, 3, 1, 512, 512, (0, 3)),
    ("convT 512->256 @64^2 x80", "convT", 80, 512, 256, 3, 1, 64, 64, (0, 2)),
    ("convT 256->128 @128^2 x80", "convT", 80, 256, 128, 3, 1, 128, 128, (0, 2)),
    ("convT 128->64 @256^2 x80", "convT", 80, 128, 64, 3, 1, 256, 256, (0, 2)),
    ("convT 64->64 @256^2 x80", "convT", 80, 64, 64, 3, 1, 256, 256, (0, 3)),
    ("convT 64->64 @128^2 x80", "convT", 80, 64, 64, 3, 1, 128, 128, (0, 3)),
    ("spy 32->64 7x7 @512^2 x16", "conv", 16, 32, 64, 7, 1, 512, 512, (0, 7, 8)),
    ("spy 64->32 7x7 @512^2 x16", "conv", 16, 64, 32, 7, 1, 512, 512, (0, 1, 4)),
    ("spy 16->32 7x7 @512^2 x16", "conv", 16, 16, 32, 7, 1, 512, 512, (0, 8)),
    ("spy 32->16 7x7 @512^2 x16", "conv", 16, 32, 16, 7, 1, 512, 512, (0, 9)),
    ("spy 16->32 7x7 @512^2 x80", "conv", 80, 16, 32, 7, 1, 512, 512, (0, 8)),
    ("spy 32->16 7x7 @512^2 x80", "conv", 80, 32, 16, 7, 1, 512, 512, (0, 9)),
    ("spy 32->16 7x7 @256^2 x80", "conv", 80, 32, 16, 7, 1, 256, 256, (0, 9)),
    ("off 32->64 3x3 @128^2 x80", "conv", 80, 32, 64, 3, 1, 128, 128, (0, 8)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--no-old", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':40s} {'variant':>10s} {'ms':>9s} {'TFLOP/s':>9s} {'% bf16 peak':>11s}", flush=True)
    for name, kind, n, cin, cout, k, stride, h, w, variants in SHAPES:
        if args.only and args.only not in name:
            continue
        wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
        b = torch.rand(cout, generator=g) - 0.5
        x32 = ops.from_nhwc(((torch.rand(n, h, w, cin, generator=g) * 2 - 1)).to(dev))
        x16 = ops.cast_bf16(x32)
        if kind == "convT":
            wT = wt.permute(1, 0, 2, 3).contiguous()
            pc = pack_convT(wT, b, dev)
            pc.wb = pack_convT_bf16(wT, dev)
            pc.w16 = pack_convT_split(pc, dev)
            flops = 2.0 * n * (2 * h) * (2 * w) * cout * cin * 2.25
        else:
            ps = kind == "ps"
            pc = pack_conv(wt, b, dev, pixel_shuffle=ps)
            pc.wb = pack_conv_bf16(wt, dev, pixel_shuffle=ps)
            if (cout, cin, k) == (16, 32, 7):
                from gpemsr_amd.packing import pack_conv7_c32_cout16
                pc.w7c16 = pack_conv7_c32_cout16(wt, dev)          # variant 9 keeps the ring kernel
            if stride == 1 and (k in (3, 7) and cin % 16 == 0 or k == 1 and cin % 32 == 0):
                pc.w16 = pack_conv_split(pc, wt, dev, pixel_shuffle=ps)
            oh, ow = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
            flops = 2.0 * n * oh * ow * cout * cin * k * k
        arms = {}
        for v in variants:
            arms[f"new v{v}"] = (lambda v=v: ops.conv2d([x16], pc, 1, stride=stride, precision="bf16", variant=v, force_mfma=True))
        if not args.no_old and pc.w16 is not None:
            arms["old split"] = (lambda: ops.conv2d([x32], pc, 1, stride=stride, precision="bf16op", force_mfma=True))
        times = {a: [] for a in arms}
        for a, fn in arms.items():
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for a, fn in arms.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(args.reps):
                    fn()
                e.record()
                torch.cuda.synchronize()
                times[a].append(s.elapsed_time(e) / args.reps)
        for a in arms:
            ms = sorted(times[a])[len(times[a]) // 2]
            tf = flops / (ms * 1e-3) / 1e12
            print(f"{name:40s} {a:>10s} {ms:9.3f} {tf:9.1f} {100 * tf / 2500:10.1f}%", flush=True)
        del x32, x16, pc
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
