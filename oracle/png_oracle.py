"""CPU restatement of the PNG edges (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else).

R:output_GPEMSR.py:95 writes every slice with ``cv2.imwrite(path, uint8 HxW)`` and R:data/util.py:75-88 reads the LR slices with
``cv2.imread(path, IMREAD_UNCHANGED)``: OpenCV delegates to libpng (PNG 1.2 / RFC 2083 file format, zlib RFC 1950 container, deflate
RFC 1951).  The file format is the contract; cv2 itself is not installed in this image (neither here nor on the GPU box), so the pins are:
the published format (chunks, CRC-32, Adler-32, stored deflate blocks, the five scanline filters), Python's zlib for the checksums and
as an independent inflater / deflater, and Pillow (libpng-compatible reader / writer) as a second decoder and as the producer of test files.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def encode_gray8_stored(img: np.ndarray) -> bytes:
    """What gpemsr_png_encode_gray8 must emit, byte for byte: filter type 0 on every scanline, zlib header 78 01, stored blocks of at
    most 65535 bytes (BFINAL on the last), Adler-32, one IDAT chunk."""
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img], axis=1).tobytes()
    z = bytearray(b"\x78\x01")
    nblk = (len(raw) + 65534) // 65535
    for b in range(nblk):
        part = raw[b * 65535:(b + 1) * 65535]
        z += bytes([1 if b == nblk - 1 else 0]) + struct.pack("<HH", len(part), len(part) ^ 0xFFFF) + part
    z += struct.pack(">I", zlib.adler32(raw) & 0xFFFFFFFF)
    return SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", bytes(z)) + _chunk(b"IEND", b"")


def parse(data: bytes):
    """-> (width, height, bit depth, colour type, interlace, concatenated IDAT payload); every chunk CRC is verified."""
    assert data[:8] == SIGNATURE, "not a PNG"
    pos, idat, hdr = 8, bytearray(), None
    while pos < len(data):
        (n,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == (zlib.crc32(kind + body) & 0xFFFFFFFF), f"CRC of {kind!r}"
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat += body
        elif kind == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    return w, h, depth, ctype, interlace, bytes(idat)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def unfilter_gray8(raw: bytes, h: int, w: int) -> np.ndarray:
    """The five scanline filters of the PNG specification (section 6; one byte per pixel)."""
    out = np.zeros((h, w), np.int32)
    for y in range(h):
        ft = raw[y * (w + 1)]
        line = raw[y * (w + 1) + 1:(y + 1) * (w + 1)]
        for x in range(w):
            a = out[y, x - 1] if x else 0
            b = out[y - 1, x] if y else 0
            c = out[y - 1, x - 1] if (x and y) else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out[y, x] = (line[x] + pred) & 255
    return out.astype(np.uint8)


def filter_gray8(img: np.ndarray, types) -> bytes:
    """Scanlines of `img` filtered with the given type per row (test-file producer: exercises all five filters)."""
    h, w = img.shape
    src = img.astype(np.int32)
    out = bytearray()
    for y in range(h):
        ft = types[y % len(types)]
        out.append(ft)
        for x in range(w):
            a = src[y, x - 1] if x else 0
            b = src[y - 1, x] if y else 0
            c = src[y - 1, x - 1] if (x and y) else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out.append((src[y, x] - pred) & 255)
    return bytes(out)


def make_png(img: np.ndarray, types=(0,), level: int = 6, strategy: int = zlib.Z_DEFAULT_STRATEGY, idat_split: int = 0) -> bytes:
    """A PNG of `img` with chosen scanline filters and zlib settings (level 0 = stored blocks, Z_FIXED = fixed Huffman codes, default =
    dynamic codes with matches); `idat_split` > 0 cuts the stream into IDAT chunks of that many bytes."""
    h, w = img.shape
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(filter_gray8(img, types)) + co.flush()
    body = b"".join(_chunk(b"IDAT", z[i:i + idat_split]) for i in range(0, len(z), idat_split)) if idat_split else _chunk(b"IDAT", z)
    return SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + body + _chunk(b"IEND", b"")


def decode_gray8(data: bytes) -> np.ndarray:
    w, h, depth, ctype, interlace, idat = parse(data)
    assert (depth, ctype, interlace) == (8, 0, 0)
    return unfilter_gray8(zlib.decompress(idat), h, w)
