"""PNG edges on the device (csrc/png.hip through the C ABI) against the oracle and against Pillow / zlib.
Encode: byte-identical to oracle.png_oracle.encode_gray8_stored, readable by Pillow.  Decode: bit-identical to
`np.array(Image.open(f)).astype(float32) / 255` (R:data/util.py:75-88 with cv2 replaced by the other libpng-compatible reader) for files
with stored / fixed / dynamic deflate blocks, all five scanline filters, split IDAT chunks; corrupted streams are reported."""
import io
import os
import sys
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import png_oracle as po      # noqa: E402
from test_png_cpu import _img      # noqa: E402

Image = pytest.importorskip("PIL.Image")


def _dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("n,h,w", [(1, 1, 1), (3, 5, 7), (2, 128, 128), (1, 255, 257), (2, 1024, 1024), (1, 63, 1039), (1, 65534, 1)])
def test_encode_matches_the_oracle_byte_for_byte(n, h, w):
    from gpemsr_amd import png
    imgs = np.stack([_img(h, w, 100 + 7 * i + h, "noise" if i % 2 == 0 else "smooth") for i in range(n)])
    files = png.encode_gray8(torch.from_numpy(imgs).to(_dev()))
    torch.cuda.synchronize()
    assert files.shape == (n, png.png_size(h, w))
    host = files.cpu().numpy()
    for i in range(n):
        want = po.encode_gray8_stored(imgs[i])
        got = host[i].tobytes()
        assert len(got) == len(want)
        if got != want:
            bad = next(k for k in range(len(want)) if got[k] != want[k])
            raise AssertionError(f"image {i}: first difference at byte {bad} of {len(want)}: {got[bad]:#x} != {want[bad]:#x}")
        assert np.array_equal(np.array(Image.open(io.BytesIO(got))), imgs[i])


def test_encode_of_the_networks_uint8_output_round_trips():
    """[n, 1, h, w] layout as forward(..., want_u8=True) returns it; decoding our own files on the device gives the pixels back."""
    from gpemsr_amd import png
    u8 = torch.from_numpy(np.stack([_img(96, 160, 3 + i) for i in range(4)])[:, None]).to(_dev())
    files = png.encode_gray8(u8).cpu().numpy()
    got = png.device_decodable([f.tobytes() for f in files])
    assert got is not None and got[:2] == (96, 160)
    x, status = png.decode_gray8(got[2], 96, 160, _dev())
    torch.cuda.synchronize()
    png.check_status(status)
    assert np.array_equal(x.cpu().numpy(), u8.cpu().numpy().astype(np.float32) / 255.)      # numpy's IEEE division (torch's device division is not)


def _files():
    out = []
    smooth, noise = _img(128, 128, 21, "smooth"), _img(128, 128, 22)
    for img in (smooth, noise):
        out.append(("stored", img, po.make_png(img, level=0)))
        out.append(("fixed", img, po.make_png(img, types=(1,), strategy=zlib.Z_FIXED)))
        out.append(("dynamic", img, po.make_png(img, types=(0, 1, 2, 3, 4))))
        out.append(("paeth-split", img, po.make_png(img, types=(4,), level=9, idat_split=1000)))
        for level in (1, 6, 9):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="PNG", compress_level=level)
            out.append((f"pillow-{level}", img, buf.getvalue()))
    return out


def test_decode_matches_numpy_bit_for_bit():
    from gpemsr_amd import png
    cases = _files()
    got = png.device_decodable([c[2] for c in cases])
    assert got is not None
    h, w, payloads = got
    x, status = png.decode_gray8(payloads, h, w, _dev())
    torch.cuda.synchronize()
    png.check_status(status, [c[0] for c in cases])
    for i, (name, img, data) in enumerate(cases):
        want = np.array(Image.open(io.BytesIO(data))).astype(np.float32) / 255.        # data/util.py:82 with the other libpng-compatible reader
        assert np.array_equal(want, img.astype(np.float32) / 255.)
        assert np.array_equal(x[i, 0].cpu().numpy(), want), name


@pytest.mark.parametrize("h,w", [(1, 1), (3, 2), (17, 1), (1, 300), (200, 333)])
def test_decode_ragged_sizes(h, w):
    from gpemsr_amd import png
    img = _img(h, w, 40 + h, "smooth")
    datas = [po.make_png(img, types=(t,)) for t in range(5)]
    hh, ww, payloads = png.device_decodable(datas)
    x, status = png.decode_gray8(payloads, hh, ww, _dev())
    torch.cuda.synchronize()
    png.check_status(status)
    for i in range(5):
        assert np.array_equal(x[i, 0].cpu().numpy(), img.astype(np.float32) / 255.)


def test_decode_reports_corruption_and_other_flavours():
    from gpemsr_amd import png
    img = _img(64, 64, 9, "smooth")
    good = po.make_png(img)
    _, _, _, _, _, idat = po.parse(good)
    broken = bytearray(idat); broken[len(broken) // 2] ^= 0x5A                 # a flipped byte inside the deflate stream
    wrong_adler = bytearray(idat); wrong_adler[-1] ^= 1
    truncated = idat[:len(idat) // 2]
    x, status = png.decode_gray8([idat, bytes(broken), bytes(wrong_adler), truncated], 64, 64, _dev())
    torch.cuda.synchronize()
    st = status.cpu().tolist()
    assert st[0] == 0 and np.array_equal(x[0, 0].cpu().numpy(), img.astype(np.float32) / 255.)
    assert st[1] != 0 and st[2] == 7 and st[3] != 0
    with pytest.raises(ValueError):
        png.check_status(status)
    rgb = io.BytesIO(); Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(rgb, format="PNG")
    assert png.device_decodable([good, rgb.getvalue()]) is None                 # the caller reads those on the host, as the reference does
    bad_crc = bytearray(good); bad_crc[40] ^= 1
    with pytest.raises(ValueError):
        png.parse_chunks(bytes(bad_crc))


def test_decode_differential_against_zlib_over_deflate_flavours():
    """120 streams of one image size produced by zlib with every combination of level {0, 1, 4, 9}, strategy {default, filtered, huffman-only,
    rle, fixed} and window {512 B, 4 KB, 32 KB}, on smooth, noisy and constant images with random filter types per row: the device inflater
    must reproduce zlib's output (and so the pixels) for all of them."""
    import struct
    from gpemsr_amd import png
    rng = np.random.default_rng(7)
    h, w = 48, 200
    imgs = [_img(h, w, 70, "smooth"), _img(h, w, 71), np.full((h, w), 131, np.uint8), (np.arange(h * w, dtype=np.int64).reshape(h, w) % 7 * 36).astype(np.uint8)]
    payloads, want = [], []
    for level in (0, 1, 4, 9):
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
            for wbits in (9, 12, 15):
                for k in range(2):
                    img = imgs[(len(payloads) + k) % len(imgs)]
                    types = tuple(int(t) for t in rng.integers(0, 5, 5))
                    co = zlib.compressobj(level, zlib.DEFLATED, wbits, int(rng.integers(1, 10)), strategy)
                    z = co.compress(po.filter_gray8(img, types)) + co.flush()
                    assert zlib.decompress(z) == po.filter_gray8(img, types)
                    payloads.append(z); want.append(img)
    x, status = png.decode_gray8(payloads, h, w, _dev())
    torch.cuda.synchronize()
    png.check_status(status)
    got = x[:, 0].cpu().numpy()
    for i, img in enumerate(want):
        assert np.array_equal(got[i], img.astype(np.float32) / 255.), i


def test_encode_crops_of_a_larger_buffer():
    """Row and image strides of the source are honoured (a crop of a wider / taller buffer is encoded without a copy), including source
    addresses that are not 4-byte aligned (the assemble kernel reads aligned dwords and funnel-shifts)."""
    from gpemsr_amd import png
    big = torch.from_numpy(np.stack([_img(150, 301, 60 + i) for i in range(3)])).to(_dev())
    for (y0, x0, h, w) in ((0, 0, 150, 301), (7, 13, 100, 200), (1, 3, 64, 257), (20, 2, 33, 17)):
        crop = big[:, y0:y0 + h, x0:x0 + w]
        assert not crop.is_contiguous() or (h, w) == (150, 301)
        files = png.encode_gray8(crop).cpu().numpy()
        for i in range(3):
            assert files[i].tobytes() == po.encode_gray8_stored(crop[i].cpu().numpy()), (y0, x0, h, w, i)


@pytest.mark.parametrize("n,h,w", [(1, 1, 1), (3, 5, 7), (2, 128, 128), (1, 255, 257), (2, 1024, 1024), (1, 63, 1039), (1, 4097, 1)])
def test_compressed_encoder_writes_valid_small_pngs(n, h, w):
    """gpemsr_png_encode_gray8_huff: every chunk CRC and the Adler-32 verify (oracle.parse / zlib), zlib inflates the stream to the scanlines
    filtered with type 0 or 1 throughout, Pillow reads the pixels back, the device decoder reads them back, and the file is no larger than
    1.01 x what zlib's Huffman-only strategy makes of the same scanlines (+ the stored-block size as the upper bound for noise)."""
    from gpemsr_amd import png
    kinds = ("smooth", "noise", "const")
    imgs = []
    for i in range(n):
        k = kinds[i % 3]
        imgs.append(np.full((h, w), 77, np.uint8) if k == "const" else _img(h, w, 300 + 11 * i + h, k))
    imgs = np.stack(imgs)
    files, sizes = png.encode_gray8_compressed(torch.from_numpy(imgs).to(_dev()))
    torch.cuda.synchronize()
    host, sz = files.cpu().numpy(), sizes.cpu().tolist()
    blobs = []
    for i in range(n):
        data = host[i, :sz[i]].tobytes()
        blobs.append(data)
        pw, ph, depth, ctype, interlace, idat = po.parse(data)                 # chunk CRCs
        assert (pw, ph, depth, ctype, interlace) == (w, h, 8, 0, 0)
        raw = zlib.decompress(idat)                                            # Adler-32, stream validity
        ft = raw[0]
        assert ft in (0, 1) and raw == po.filter_gray8(imgs[i], (ft,))
        assert np.array_equal(np.array(Image.open(io.BytesIO(data))), imgs[i])
        co = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_HUFFMAN_ONLY)
        best = min(len(co.compress(po.filter_gray8(imgs[i], (f,))) + co.flush()) if f == ft else 1 << 60 for f in (0, 1))
        assert len(idat) <= 1.01 * best + 160, (len(idat), best)
        assert sz[i] <= png.png_size(h, w) + 1024            # uniform noise does not compress: 8 bits per symbol + the block header
    dec = png.device_decodable(blobs)
    assert dec is not None
    x, status = png.decode_gray8(dec[2], h, w, _dev())
    torch.cuda.synchronize()
    png.check_status(status)
    assert np.array_equal(x[:, 0].cpu().numpy(), imgs.astype(np.float32) / 255.)


def test_compressed_encoder_length_limit():
    """A Fibonacci-like histogram makes the unrestricted Huffman code deeper than 15 bits: the length-limiting rule (zlib's) must still give a
    complete code that zlib accepts."""
    from gpemsr_amd import png
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    vals = np.concatenate([np.full(c, 10 + i, np.uint8) for i, c in enumerate(fib)])
    rng = np.random.default_rng(0); rng.shuffle(vals)
    w = 257; h = len(vals) // w
    img = vals[:h * w].reshape(h, w)
    files, sizes = png.encode_gray8_compressed(torch.from_numpy(img[None]).to(_dev()))
    data = files[0, :int(sizes[0])].cpu().numpy().tobytes()
    assert np.array_equal(np.array(Image.open(io.BytesIO(data))), img)
    assert np.array_equal(po.decode_gray8(data), img)
