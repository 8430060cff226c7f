"""The PNG oracle (oracle/png_oracle.py) against independent implementations of the published format: Python's zlib (inflate, CRC-32,
Adler-32) and Pillow (libpng-compatible reader and writer).  The reference writes / reads its slices with OpenCV
(R:output_GPEMSR.py:95, R:data/util.py:75-88), which is not installed in this image; the file format is the contract."""
import io
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import png_oracle as po      # noqa: E402

Image = pytest.importorskip("PIL.Image")


def _img(h, w, seed, kind="noise"):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    y, x = np.mgrid[0:h, 0:w]
    return ((np.sin(x / 7.0) + np.cos(y / 5.0) + 2) * 60 + rng.integers(0, 8, (h, w))).astype(np.uint8)      # smooth: the deflater finds matches


@pytest.mark.parametrize("h,w", [(1, 1), (5, 7), (128, 128), (255, 257), (300, 1024)])
def test_stored_png_is_read_by_pillow_and_zlib(h, w):
    img = _img(h, w, 1 + h)
    data = po.encode_gray8_stored(img)
    assert len(data) == 57 + 2 + 5 * ((h * (w + 1) + 65534) // 65535) + h * (w + 1) + 4
    got = np.array(Image.open(io.BytesIO(data)))
    assert got.dtype == np.uint8 and np.array_equal(got, img)
    pw, ph, depth, ctype, interlace, idat = po.parse(data)                      # verifies every chunk CRC with zlib.crc32
    assert (pw, ph, depth, ctype, interlace) == (w, h, 8, 0, 0)
    assert np.array_equal(po.unfilter_gray8(zlib.decompress(idat), h, w), img)


@pytest.mark.parametrize("types", [(0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4), (4, 3, 2, 1)])
def test_filters_round_trip_and_match_pillow(types):
    img = _img(37, 53, 11, "smooth")
    data = po.make_png(img, types=types)
    assert np.array_equal(np.array(Image.open(io.BytesIO(data))), img)           # Pillow undoes our filters
    assert np.array_equal(po.decode_gray8(data), img)


def test_oracle_reads_pillow_files():
    for kind in ("noise", "smooth"):
        img = _img(64, 96, 5, kind)
        for level in (0, 1, 6, 9):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="PNG", compress_level=level)
            assert np.array_equal(po.decode_gray8(buf.getvalue()), img)


def test_host_side_chunk_parser_and_routing():
    """gpemsr_amd.png.parse_chunks / device_decodable (host code of the device decoder): IDAT payloads are concatenated across split chunks, CRCs
    are verified, and only non-interlaced 8-bit grayscale files of one size are routed to the device -- everything else is read on the host,
    as the reference does (R:data/util.py:75-88)."""
    from gpemsr_amd import png
    img = _img(24, 40, 3, "smooth")
    whole, split = po.make_png(img), po.make_png(img, idat_split=50)
    assert png.parse_chunks(whole)[:5] == (40, 24, 8, 0, 0)
    assert png.parse_chunks(split)[5] == png.parse_chunks(whole)[5] == po.parse(whole)[5]
    assert np.array_equal(po.unfilter_gray8(zlib.decompress(png.parse_chunks(split)[5]), 24, 40), img)
    h, w, payloads = png.device_decodable([whole, split])
    assert (h, w, len(payloads)) == (24, 40, 2)
    buf = io.BytesIO(); Image.fromarray(img.astype(np.uint16) * 256).save(buf, format="PNG")            # 16-bit grayscale
    assert png.device_decodable([whole, buf.getvalue()]) is None
    buf = io.BytesIO(); Image.fromarray(np.stack([img] * 3, axis=-1)).save(buf, format="PNG")            # RGB
    assert png.device_decodable([buf.getvalue()]) is None
    assert png.device_decodable([whole, po.make_png(_img(24, 41, 4))]) is None                            # mixed sizes
    bad = bytearray(whole); bad[-20] ^= 0xFF
    with pytest.raises(ValueError):
        png.parse_chunks(bytes(bad))
    with pytest.raises(ValueError):
        png.parse_chunks(b"not a png at all")
