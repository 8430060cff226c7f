#!/usr/bin/env python3
"""Inference entry point -- drop-in for the reference's ``output_GPEMSR.py``:

    python output_GPEMSR.py -opt option/output_GPEMSR_x8.yml

Same CLI, same option-file keys, same outputs (``save_path/{k}.png``, 8-bit grayscale,
k = 0..n-1 over the LQ volume; the first/last two slices use replicated neighbours exactly
as /root/reference/GPEMSR-CREMI/GPEMSR/output_GPEMSR.py:54-84,98-128), but the model is
``gpemsr_amd.GPEMSR`` running on the MI355X HIP kernels.  Differences, all additive:
  * PNG I/O uses Pillow when OpenCV is not installed (same uint8 data either way);
  * volume mode (default; option ``volume_cache: false`` restores one independent forward per window): every LQ slice
    goes through the per-slice half of the network once and the sliding windows share the cached features
    (bit-identical output, ~2.8x less work per slice); ``volume_block`` windows are processed per call;
  * ``tile_batch`` (option key) > 1 batches several 5-slice windows per forward call (without the cache);
  * uint8 conversion, PNG decoding of the LR slices and PNG encoding of the outputs run on the device (csrc/png.hip; `png_on_device:
    false`: host codec in a small thread pool); the files hold the same pixels as cv2.imwrite's, Huffman-compressed on the device (`png_compress: false`: stored deflate blocks);
  * with several ranks (torchrun), the volume is sharded along z over GPUs: contiguous output slices per rank, each rank reading
    its own LR slices plus a 2-slice halo (gpemsr_amd.dist.plan_volume_shard);
  * ``synthetic_weights_if_missing: true`` lets the script run without the authors'
    Google-Drive checkpoints (deterministic synthetic weights) -- for smoke runs only.
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)

from gpemsr_amd import dist as gdist                      # noqa: E402
from gpemsr_amd.config import build_model, dict_to_nonedict, load_options   # noqa: E402
from gpemsr_amd.imgutil import tensor2img                 # noqa: E402


def read_img(path: str) -> np.ndarray:
    """data/util.py:75-88: uint8 image -> float32 HWC in [0,1] (grayscale -> HxWx1)."""
    try:
        import cv2
        img = cv2.imread(path, cv2.IMREAD_UNCHANGED)
    except ImportError:
        from PIL import Image
        img = np.array(Image.open(path))
    img = img.astype(np.float32) / 255.
    if img.ndim == 2:
        img = np.expand_dims(img, axis=2)
    if img.shape[2] > 3:
        img = img[:, :, :3]
    return img


def save_img(img: np.ndarray, path: str):
    try:
        import cv2
        cv2.imwrite(path, img)
    except ImportError:
        from PIL import Image
        Image.fromarray(img).save(path)


def seek_path(idx, dir_path, center):            # output_GPEMSR.py:216-222
    cur = center + idx
    p = osp.join(dir_path, str(cur) + '.png')
    if osp.exists(p):
        return p
    return seek_path(idx - 1, dir_path, center)


class CREMIWindows:
    """The val-phase behaviour of the reference's CREMIDataset (output_GPEMSR.py:132-214):
    item i = the N-slice LQ window centred on the (i + N//2)-th GT slice."""

    def __init__(self, opt):
        self.N = opt['N_frames']
        self.LQ_root = opt['dataroot_LQ']
        ids = sorted(int(f[:-4]) for f in os.listdir(opt['dataroot_GT']) if f.endswith('.png'))
        half = (self.N - 1) // 2
        self.centres = ids[half:len(ids) - half] if half else ids
        self.offsets = list(range(-half, half + 1))

    def __len__(self):
        return len(self.centres)

    def paths(self, i):
        c = self.centres[i]
        return [seek_path(o, self.LQ_root, c) for o in self.offsets]

    def __getitem__(self, i):
        return load_frames(self.paths(i))


def load_frames(paths) -> torch.Tensor:
    lq = np.stack([read_img(p) for p in paths], axis=0)                                      # N,H,W,1
    return torch.from_numpy(np.ascontiguousarray(np.transpose(lq, (0, 3, 1, 2)))).float()   # N,1,H,W


def window_paths(ds: CREMIWindows):
    """The LQ files of all 5-slice windows of the volume in output order; the first/last two outputs replicate edge
    slices exactly as the reference (output_GPEMSR.py:54-84, 98-128)."""
    first, last = ds.paths(0), ds.paths(len(ds) - 1)
    wins = [[first[k] for k in (0, 0, 0, 1, 2)], [first[k] for k in (0, 0, 1, 2, 3)]]       # slices 0 and 1
    wins += [ds.paths(i) for i in range(len(ds))]
    n = len(last)
    wins += [[last[k] for k in (n - 4, n - 3, n - 2, n - 1, n - 1)], [last[k] for k in (n - 3, n - 2, n - 1, n - 1, n - 1)]]
    return wins


def build_windows(ds: CREMIWindows):
    """All 5-slice windows of the volume as tensors (one independent window per entry)."""
    return [load_frames(p) for p in window_paths(ds)]


def index_windows(win_paths):
    """Distinct LQ files of a run of windows (first-use order) and the [Wn, N] matrix of frame numbers into them."""
    order, number = [], {}
    rows = []
    for paths in win_paths:
        row = []
        for p in paths:
            if p not in number:
                number[p] = len(order); order.append(p)
            row.append(number[p])
        rows.append(row)
    return order, torch.tensor(rows, dtype=torch.int32)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    parser.add_argument('--local_rank', type=int, default=os.getenv('LOCAL_RANK', -1))
    args = parser.parse_args()
    opt = load_options(args.opt)
    im_path_SR, scale = opt['save_path'], opt['scale']
    os.makedirs(im_path_SR, exist_ok=True)
    dataset_dict = dict(opt['dataset']); dataset_dict['scale'] = scale
    ds = CREMIWindows(dict_to_nonedict(dataset_dict))
    assert ds.N == 5, "the edge handling of the reference script is written for N_frames = 5"

    rank, world, local = gdist.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("output_GPEMSR.py (gpemsr_amd) needs an MI355X / HIP device: there is no CPU path")
    torch.cuda.set_device(max(local, 0))
    device = torch.device("cuda", max(local, 0))
    # the reference raises when a prior file is missing (model/GPEMSR.py:275-276); only the explicit opt-in below may
    # fall back to the deterministic synthetic prior
    syn = bool(opt.get('synthetic_weights_if_missing', False))
    net = opt['network']
    have_prior = all(p and osp.exists(str(p)) for p in (net['ref_path_G'], net['ref_path_Indexer']))
    model = build_model(opt, load_prior_files=have_prior or not syn).eval().to(device)
    pretrain_path = opt['pretrain_path']
    if pretrain_path and osp.exists(pretrain_path):
        model.load_state_dict(torch.load(pretrain_path, map_location="cpu"), strict=True)   # output_GPEMSR.py:52
    elif not opt.get('synthetic_weights_if_missing', False):
        raise FileNotFoundError(pretrain_path)
    elif rank == 0:
        print(f"[gpemsr_amd] {pretrain_path} not found: using deterministic synthetic weights", file=sys.stderr)

    from concurrent.futures import ThreadPoolExecutor
    from gpemsr_amd import ops
    wpaths = window_paths(ds)
    # z-sharding (gpemsr_amd.dist.plan_volume_shard): rank r owns the contiguous output slices [lo, hi); the blocks below load its
    # own LR slices plus the 2-slice halo its edge windows reach into (no collective on the data path: every rank reads its own
    # files and writes its own PNGs)
    lo, hi = gdist.shard_range(len(wpaths), rank, world)
    if world > 1 and len(wpaths) >= 5:
        files, rows = index_windows(wpaths[lo:hi])
        p_lo, p_hi, s_lo, s_hi, _ = gdist.plan_volume_shard(len(wpaths), rank, world)
        assert (p_lo, p_hi) == (lo, hi) and len(files) <= (hi - lo) + 4, "window shard and slice halo disagree with the shard plan"
    tb = int(opt.get('tile_batch', 1) or 1)
    use_cache = opt.get('volume_cache', True) is not False
    block = int(opt.get('volume_block', 64) or 64) if use_cache else tb
    writers = ThreadPoolExecutor(max_workers=int(opt.get('writer_threads', 4) or 4))
    loaders = ThreadPoolExecutor(max_workers=1)

    # PNG edges on the device (gpemsr_amd/png.py, csrc/png.hip; option `png_on_device: false` restores host decode / encode): the loader
    # thread only reads file BYTES, the device inflates + unfilters + converts the 8-bit grayscale slices; the 8-bit outputs become complete
    # PNG files in HBM (stored deflate blocks) and the writer threads copy bytes to disk -- no zlib on the host at 150+ images per second
    # and GPU.  Files of another flavour (16-bit, colour, interlaced) are decoded on the host, as the reference does (cv2 / Pillow).
    png_dev = opt.get('png_on_device', True) is not False
    png_zip = opt.get('png_compress', True) is not False   # Huffman-compressed stream (files of zlib's size; `false`: stored blocks, 4x faster to make)
    from gpemsr_amd import png as gpng

    def load_block(b0):
        b1 = min(hi, b0 + block)
        files, win = index_windows(wpaths[b0:b1])
        if png_dev:
            blobs = []
            for f in files:
                with open(f, 'rb') as fh:
                    blobs.append(fh.read())
            dec = gpng.device_decodable(blobs)
            if dec is not None:
                return {"png": dec, "files": files, "win": win}
        if use_cache:
            return {"host": load_frames(files).pin_memory(), "win": win}
        return {"host": torch.stack([load_frames(p) for p in wpaths[b0:b1]], dim=0).pin_memory(), "win": None}

    def to_device(blk):
        """-> (model input on the device, window matrix or None, decode status or None)"""
        if "png" in blk:
            h, w, payloads = blk["png"]
            x, status = gpng.decode_gray8(payloads, h, w, device)                    # [files, 1, h, w] float32 / 255
            if use_cache:
                return x, blk["win"], (status, blk["files"])
            return x[blk["win"].to(device).long()], None, (status, blk["files"])      # [windows, N, 1, h, w]
        return blk["host"].to(device, non_blocking=True), blk["win"], None

    def write_bytes(buf, path):
        with open(path, 'wb') as fh:
            fh.write(buf)

    pending = []

    def flush(item):
        # the block's output has landed in pinned host memory once its event has passed: hand it to the writers
        hostbuf, ev, c0, c1, status, hostsizes, devfull = item
        ev.synchronize()
        if status is not None:
            gpng.check_status(status[0], status[1])
        if hostsizes is not None and int(hostsizes.max()) > hostbuf.shape[1]:
            # the trimmed D2H copy assumed a near-optimal prefix code; the encoder limits code lengths with zlib's overflow rule, not
            # package-merge, so the bound is not enforced anywhere: a block with a longer file is copied again at the encoder's full capacity
            hostbuf = devfull.cpu()
            assert int(hostsizes.max()) <= hostbuf.shape[1], "PNG encoder reported a file larger than its own capacity"
        u8 = hostbuf.numpy()
        for j in range(c1 - c0):
            path = osp.join(im_path_SR, '{}.png'.format(c0 + j))
            if png_dev:
                data = u8[j].tobytes() if hostsizes is None else u8[j, :int(hostsizes[j])].tobytes()
                pending.append(writers.submit(write_bytes, data, path))
            else:
                pending.append(writers.submit(save_img, u8[j], path))

    # the next block's upload + PNG decode run on a side stream while the current block's forward occupies the main one (the inflate is one
    # lane per image: it needs 80 wave slots for a few milliseconds, not the machine)
    side = torch.cuda.Stream(device=device)

    def start_upload(blk):
        with torch.cuda.stream(side):
            x, win, status = to_device(blk)
            ev = torch.cuda.Event()
            ev.record(side)
        return x, win, status, ev

    import time as _time
    torch.cuda.synchronize()
    t_loop = _time.perf_counter()
    marks = []
    with torch.no_grad():
        cur = start_upload(load_block(lo)) if lo < hi else None
        inflight = None
        for b0 in range(lo, hi, block):
            b1 = min(hi, b0 + block)
            x, win, status, ev_up = cur
            torch.cuda.current_stream().wait_event(ev_up)
            x.record_stream(torch.cuda.current_stream())
            nxt = loaders.submit(load_block, b1) if b1 < hi else None       # read (host path: decode) the next block while the GPU works
            # the 8-bit image is written by the network's last kernel (conv_last + bilinear base + tensor2img): 1 B/pixel D2H
            if use_cache:
                _, _, u8 = model.forward_volume(x, win, want_u8=True)
            else:
                _, _, u8 = model(x, want_u8=True)
            hostsizes, devfull = None, None
            if png_dev and png_zip:
                hh, ww = int(u8.shape[-2]), int(u8.shape[-1])
                u8, fsz = gpng.encode_gray8_compressed(u8.reshape(-1, hh, ww))      # [slices, capacity], sizes
                devfull = u8
                # an optimal prefix code is never longer than the 8-bit one by more than a bit per 256 symbols: the D2H copy stops there, not
                # at the encoder's worst-case capacity (2 bytes per symbol)
                u8 = u8[:, :min(u8.shape[1], gpng.png_size(hh, ww) + hh * (ww + 1) // 128 + 1024)]
                hostsizes = torch.empty(fsz.shape, dtype=torch.int64, pin_memory=True)
                hostsizes.copy_(fsz, non_blocking=True)
            elif png_dev:
                u8 = gpng.encode_gray8(u8.reshape(-1, u8.shape[-2], u8.shape[-1]))      # [slices, file bytes]
            # asynchronous D2H into pinned memory; the host only waits for block k-1 AFTER block k has been enqueued, so the
            # device never idles between blocks (R:output_GPEMSR.py:88-95 moves every slice synchronously)
            hostbuf = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
            hostbuf.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            if nxt is not None:
                cur = start_upload(nxt.result())                           # enqueued behind nothing: overlaps with this block's forward
            if inflight is not None:
                flush(inflight)
                marks.append((_time.perf_counter(), inflight[3]))           # (time, slices written so far): the loop is paced by the device
            inflight = (hostbuf, ev, b0, b1, status, hostsizes, devfull)
        if inflight is not None:
            flush(inflight)
    for f in pending:
        f.result()
    writers.shutdown(); loaders.shutdown()
    if os.environ.get("GPEMSR_CLI_TIMING") and hi > lo:
        dt = _time.perf_counter() - t_loop
        steady = ""
        if len(marks) >= 2:                                # without the first block (one-time costs: first launches, allocator growth)
            (ta, sa), (tb, sb) = marks[0], (_time.perf_counter(), hi)
            steady = f"; after the first block: {(sb - sa) / (tb - ta):.1f} slices/s"
        print(f"[gpemsr_amd] loop: {hi - lo} slices in {dt:.3f} s = {(hi - lo) / dt:.1f} slices/s (files read -> files written){steady}", file=sys.stderr)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
