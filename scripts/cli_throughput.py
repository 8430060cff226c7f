#!/usr/bin/env python3
"""End-to-end throughput of the CLI (PNG files in -> PNG files out) on a synthetic 240-slice volume of 128 x 128 LR slices -> 1024 x 1024, bf16
path, for the three output codecs: device Huffman (default), device stored blocks, host codec (Pillow here, cv2 when installed).  Wall time of
the whole process includes start-up (imports, weight packing); the loop time is printed by the CLI run under GPEMSR_CLI_TIMING=1.
python3 scripts/cli_throughput.py"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import yaml
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = 288
tmp = tempfile.mkdtemp(prefix="gpemsr_cli_")
rng = np.random.default_rng(0)
y, x = np.mgrid[0:128, 0:128]
for sub, size in (("LQ", 128), ("GT", 1024)):
    os.makedirs(os.path.join(tmp, sub))
    for i in range(n):
        if sub == "LQ":
            a = ((np.sin(x / 9.0 + 0.2 * i) + np.cos(y / 7.0) + 2) * 55 + rng.integers(0, 24, (128, 128))).astype(np.uint8)
        else:
            a = np.zeros((8, 8), np.uint8)                      # only the file names of the GT folder are read
        Image.fromarray(a).save(os.path.join(tmp, sub, f"{i}.png"))
opt = yaml.safe_load(open(os.path.join(ROOT, "option", "output_GPEMSR_x8.yml")))
opt["dataset"]["dataroot_GT"], opt["dataset"]["dataroot_LQ"] = os.path.join(tmp, "GT"), os.path.join(tmp, "LQ")
opt["pretrain_path"] = os.path.join(tmp, "missing.pth")
opt["synthetic_weights_if_missing"] = True
opt["precision"] = "bf16"
opt["volume_block"] = 48
for name, dev, zipped in (("device Huffman", True, True), ("device stored", True, False), ("host codec", False, False)):
    opt["save_path"] = os.path.join(tmp, "sr_" + name.replace(" ", "_"))
    opt["png_on_device"], opt["png_compress"] = dev, zipped
    yml = os.path.join(tmp, "cli.yml")
    yaml.safe_dump(opt, open(yml, "w"))
    env = dict(os.environ, GPEMSR_CLI_TIMING="1")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "output_GPEMSR.py"), "-opt", yml], env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    loop = [l for l in r.stderr.splitlines() if l.startswith("[gpemsr_amd] loop")]
    sizes = [os.path.getsize(os.path.join(opt["save_path"], f"{k}.png")) for k in range(n)]
    print(f"{name:15s}: process {dt:6.2f} s; {loop[-1] if loop else ''}; {np.mean(sizes) / 1e3:.0f} KB per file", flush=True)
