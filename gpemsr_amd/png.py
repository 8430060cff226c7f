"""PNG edges of the inference loop on the device (csrc/png.hip through the C ABI).

* ``encode_gray8`` replaces ``cv2.imwrite(path, output)`` of R:output_GPEMSR.py:95: the 8-bit images the network's last kernel left in
  HBM become complete PNG files in HBM (stored deflate blocks, checksums on the device); the host copies bytes to disk.
* ``decode_gray8`` replaces ``cv2.imread(path, IMREAD_UNCHANGED).astype(float32) / 255`` of R:data/util.py:75-88 for non-interlaced
  8-bit grayscale files (the CREMI slices): the host reads the chunk lengths and the 13 IHDR bytes, the device inflates, unfilters and
  converts.  Files of another flavour (16-bit, colour, palette, interlaced) are reported as such; the caller reads them on the host, as the
  reference does.
"""
from __future__ import annotations

import struct
import zlib
from typing import List, Optional, Sequence, Tuple

import torch

from . import _abi

_SIG = b"\x89PNG\r\n\x1a\n"


def png_size(h: int, w: int) -> int:
    return int(_abi.load().gpemsr_png_gray8_size(h, w))


def encode_gray8(u8: torch.Tensor) -> torch.Tensor:
    """[n, h, w] (or [n, 1, h, w]) uint8 on the device -> [n, png_size(h, w)] uint8 on the device: one complete PNG file per row."""
    lib = _abi.load()
    if not u8.is_cuda:
        raise RuntimeError("gpemsr_amd.png: images must live on a cuda/HIP device (there is no CPU path)")
    assert u8.dtype == torch.uint8
    if u8.dim() == 4:
        assert u8.shape[1] == 1
        u8 = u8[:, 0]
    if u8.stride(2) != 1 or u8.stride(1) < u8.shape[2] or (u8.shape[0] > 1 and u8.stride(0) < 0):
        u8 = u8.contiguous()                               # rows must be dense; row / image strides are passed through (crops of a larger buffer)
    n, h, w = u8.shape
    size = png_size(h, w)
    stride = (size + 15) // 16 * 16                       # every file starts 16-byte aligned (vector stores)
    out = torch.empty((n, stride), dtype=torch.uint8, device=u8.device)
    ws = torch.empty(int(lib.gpemsr_png_encode_workspace(n, h, w)) // 8 + 1, dtype=torch.int64, device=u8.device)
    _abi.check(lib.gpemsr_png_encode_gray8(u8.data_ptr(), n, h, w, u8.stride(0) if n > 1 else h * w, u8.stride(1), out.data_ptr(), stride, ws.data_ptr(), ws.numel() * 8,
                                           torch.cuda.current_stream().cuda_stream), "png_encode_gray8")
    return out[:, :size]


def encode_gray8_compressed(u8: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[n, h, w] uint8 on the device -> ([n, capacity] uint8, sizes [n] int64), both on the device: file i = row i[:sizes[i]].  The zlib stream
    is one dynamic-Huffman block of literals (csrc/png_huff.hip): the files are about as small as zlib's for 8-bit EM slices, where LZ77
    matches are rare."""
    lib = _abi.load()
    if not u8.is_cuda:
        raise RuntimeError("gpemsr_amd.png: images must live on a cuda/HIP device (there is no CPU path)")
    assert u8.dtype == torch.uint8
    if u8.dim() == 4:
        assert u8.shape[1] == 1
        u8 = u8[:, 0]
    if u8.stride(2) != 1 or u8.stride(1) < u8.shape[2] or (u8.shape[0] > 1 and u8.stride(0) < 0):
        u8 = u8.contiguous()
    n, h, w = u8.shape
    cap = int(lib.gpemsr_png_huff_capacity(h, w))
    out = torch.empty((n, cap), dtype=torch.uint8, device=u8.device)
    sizes = torch.empty(n, dtype=torch.int64, device=u8.device)
    ws = torch.empty(int(lib.gpemsr_png_huff_workspace(n, h, w)) // 8 + 2, dtype=torch.int64, device=u8.device)      # (torch allocations are 256-byte aligned)
    _abi.check(lib.gpemsr_png_encode_gray8_huff(u8.data_ptr(), n, h, w, u8.stride(0) if n > 1 else h * w, u8.stride(1), out.data_ptr(), cap,
                                                sizes.data_ptr(), ws.data_ptr(), ws.numel() * 8, torch.cuda.current_stream().cuda_stream),
               "png_encode_gray8_huff")
    return out, sizes


def parse_chunks(data: bytes) -> Tuple[int, int, int, int, int, bytes]:
    """-> (width, height, bit depth, colour type, interlace, concatenated IDAT payload); chunk CRCs verified (host: 4 bytes per chunk)."""
    if data[:8] != _SIG:
        raise ValueError("not a PNG file")
    pos, hdr, idat = 8, None, []
    while pos + 12 <= len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        kind = data[pos + 4:pos + 8]
        if pos + 12 + n > len(data):
            raise ValueError(f"PNG chunk {kind!r}: truncated")
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        # libpng's default CRC action (cv2.imread, the reference's reader): an error on every CRITICAL chunk (upper-case first letter: IHDR, PLTE,
        # IDAT, IEND), a warning only on ancillary ones
        if kind[:1].isupper() and crc != (zlib.crc32(kind + body) & 0xFFFFFFFF):
            raise ValueError(f"PNG chunk {kind!r}: CRC mismatch")
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError("PNG without IHDR / IDAT")
    w, h, depth, ctype, _, _, interlace = hdr
    return w, h, depth, ctype, interlace, b"".join(idat)


def device_decodable(files: Sequence[bytes]) -> Optional[Tuple[int, int, List[bytes]]]:
    """(h, w, IDAT payloads) when every file is a well-formed non-interlaced 8-bit grayscale PNG of one size, else None (host codec)."""
    payloads, hw = [], None
    for data in files:
        try:
            w, h, depth, ctype, interlace, idat = parse_chunks(data)
        except (ValueError, struct.error, IndexError):
            return None             # anything the chunk walk cannot take goes to the host codec, which reports (or reads) it as the reference does
        if (depth, ctype, interlace) != (8, 0, 0) or w > 16384 or (hw is not None and hw != (h, w)):      # (two scanlines live in LDS)
            return None
        hw = (h, w)
        payloads.append(idat)
    return hw[0], hw[1], payloads


def decode_gray8(payloads: Sequence[bytes], h: int, w: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """IDAT payloads of n files -> ([n, 1, h, w] float32 in [0, 1] on the device, status [n] int32 on the device).  Asynchronous: check
    `status` (all zero) once the stream has been synchronised -- `check_status`."""
    lib = _abi.load()
    n = len(payloads)
    offs = [0]
    for p in payloads:
        offs.append(offs[-1] + len(p))
    host = torch.empty(max(offs[-1], 1), dtype=torch.uint8).pin_memory()
    host[:offs[-1]] = torch.frombuffer(bytearray(b"".join(payloads)), dtype=torch.uint8)
    z = host.to(device, non_blocking=True)
    o = torch.tensor(offs, dtype=torch.int64).pin_memory().to(device, non_blocking=True)
    raw = torch.empty(n * h * (w + 1), dtype=torch.uint8, device=device)
    out = torch.empty((n, 1, h, w), dtype=torch.float32, device=device)
    status = torch.empty(n, dtype=torch.int32, device=device)
    _abi.check(lib.gpemsr_png_decode_gray8(z.data_ptr(), o.data_ptr(), n, h, w, raw.data_ptr(), out.data_ptr(), 255.0, status.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "png_decode_gray8")
    return out, status


_STATUS = {9: "image too wide for the device path", 1: "bad zlib header", 2: "bad deflate block", 3: "bad Huffman table", 4: "bad symbol or distance", 5: "size mismatch",
           6: "input exhausted", 7: "Adler-32 mismatch", 8: "bad scanline filter"}


def check_status(status: torch.Tensor, names: Optional[Sequence[str]] = None):
    st = status.cpu().tolist()
    for i, s in enumerate(st):
        if s != 0:
            raise ValueError(f"PNG decode failed for {names[i] if names else i}: {_STATUS.get(s, s)}")
