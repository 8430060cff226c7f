#!/usr/bin/env python3
"""Timing of the PNG edges on the device (csrc/png.hip): the encoder on a batch of 16 network outputs (1024 x 1024, 8 bit) and the decoder on
80 LR slices (128 x 128), next to the host codecs they replace (zlib level 1 / 6 via Pillow on one core).  python3 scripts/png_bench.py"""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpemsr_amd import png  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
y, x = np.mgrid[0:1024, 0:1024]
hr = np.stack([((np.sin(x / 9.0 + i) + np.cos(y / 7.0) + 2) * 55 + rng.integers(0, 24, (1024, 1024))).astype(np.uint8) for i in range(16)])
u8 = torch.from_numpy(hr).to(dev)
for _ in range(3):
    files = png.encode_gray8(u8)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    files = png.encode_gray8(u8)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
nbytes = files.numel()
print(f"encode 16 x 1024^2: {ms:.3f} ms per batch = {16 / ms * 1e3:.0f} images/s, {nbytes / 16} bytes per file; algorithmic bytes (image read by the "
      f"assemble and Adler passes, file written once and read once by the CRC pass) {4 * nbytes / 1e6:.1f} MB -> {4 * nbytes / ms / 1e6:.0f} GB/s")
for _ in range(3):
    cfiles, csizes = png.encode_gray8_compressed(u8)
torch.cuda.synchronize()
s.record()
for _ in range(20):
    cfiles, csizes = png.encode_gray8_compressed(u8)
e.record(); torch.cuda.synchronize()
cms = s.elapsed_time(e) / 20
print(f"encode 16 x 1024^2, compressed (one dynamic-Huffman block per image): {cms:.3f} ms per batch = {16 / cms * 1e3:.0f} images/s, "
      f"{float(csizes.float().mean()):.0f} bytes per file on average ({float(csizes.float().mean()) / (nbytes / 16):.3f} of the stored size)")
try:
    from PIL import Image
    for level in (1, 6):
        t0 = time.perf_counter()
        for i in range(4):
            buf = io.BytesIO(); Image.fromarray(hr[i]).save(buf, format="PNG", compress_level=level)
        dt = (time.perf_counter() - t0) / 4
        print(f"host Pillow compress_level={level}: {dt * 1e3:.1f} ms per image on one core ({len(buf.getvalue())} bytes)")
except ImportError:
    pass
lr = [hr[i % 16, :128, :128].copy() for i in range(80)]
blobs = []
for a in lr:
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="PNG"); blobs.append(buf.getvalue())
h, w, payloads = png.device_decodable(blobs)
for _ in range(3):
    xs, st = png.decode_gray8(payloads, h, w, dev)
torch.cuda.synchronize()
png.check_status(st)
t0 = time.perf_counter()
for _ in range(10):
    xs, st = png.decode_gray8(payloads, h, w, dev)
torch.cuda.synchronize()
print(f"decode 80 x 128^2 (Pillow-written files, {sum(map(len, payloads)) / 80:.0f} compressed bytes each): {(time.perf_counter() - t0) * 100:.3f} ms per batch incl. the upload")
t0 = time.perf_counter()
for b in blobs:
    np.array(Image.open(io.BytesIO(b))).astype(np.float32) / 255.
print(f"host Pillow decode of the same 80 files: {(time.perf_counter() - t0) * 1e3:.1f} ms on one core")
