"""CPU: the PNG container of the file-level codec (gscodec_studio_amd/compression/png_compression.py).  The reader is checked
against scanlines filtered HERE, with an independent statement of the five PNG filter types (PNG specification 9.2-9.4), so
that files from other encoders (imageio / Pillow pick Average and Paeth freely) decode; the writer against the reader."""
import os
import struct
import zlib

import numpy as np
import pytest

from gscodec_studio_amd.compression import png_read, png_write

MAGIC = b"\x89PNG\r\n\x1a\n"
CTYPE = {1: 0, 2: 4, 3: 2, 4: 6}


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _filter_rows(img, types):
    """Forward filtering, one filter type per row (spec 9.2): returns the IDAT payload before compression."""
    h, w, c = img.shape
    rows = img.reshape(h, w * c).astype(np.int32)
    out = bytearray()
    for y in range(h):
        ft = types[y]
        out.append(ft)
        for x in range(w * c):
            a = rows[y, x - c] if x >= c else 0
            b = rows[y - 1, x] if y else 0
            cc = rows[y - 1, x - c] if (y and x >= c) else 0
            pred = [0, a, b, (a + b) // 2, _paeth(a, b, cc)][ft]
            out.append((rows[y, x] - pred) & 0xFF)
    return bytes(out)


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_reader_undoes_all_five_filter_types(tmp_path, c):
    rs = np.random.RandomState(c)
    h, w = 23, 19
    img = rs.randint(0, 256, size=(h, w, c)).astype(np.uint8)
    img[:, :, 0] = (np.arange(w)[None, :] * 7 + np.arange(h)[:, None] * 3) % 256  # a smooth channel: large predictions
    types = [y % 5 for y in range(h)]
    rs.shuffle(types)
    body = _filter_rows(img, types)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, CTYPE[c], 0, 0, 0)
    # two IDAT chunks and an ancillary chunk in between, as real files have
    z = zlib.compress(body, 9)
    data = MAGIC + _chunk(b"IHDR", ihdr) + _chunk(b"tEXt", b"Comment\x00x") + _chunk(b"IDAT", z[:11]) + _chunk(b"IDAT", z[11:]) + _chunk(b"IEND", b"")
    p = tmp_path / "f.png"
    p.write_bytes(data)
    out = png_read(str(p))
    assert np.array_equal(out, img[:, :, 0] if c == 1 else img)


@pytest.mark.parametrize("shape", [(1, 1), (7, 5), (40, 33, 1), (16, 16, 2), (65, 31, 3), (12, 50, 4)])
def test_writer_round_trip_and_filter_choice(tmp_path, shape):
    rs = np.random.RandomState(sum(shape))
    img = rs.randint(0, 256, size=shape).astype(np.uint8)
    if len(shape) == 3 and shape[0] > 4:
        img[::2] = img[1::2][: img[::2].shape[0]] if img[1::2].shape[0] == img[::2].shape[0] else img[::2]  # rows equal to a neighbour: Up
        img[3, :] = np.arange(shape[1] * shape[2]).reshape(shape[1], shape[2]) % 256  # a ramp: Sub
    p = str(tmp_path / "w.png")
    png_write(p, img)
    out = png_read(p)
    want = img[:, :, 0] if (img.ndim == 3 and img.shape[2] == 1) else img
    assert out.dtype == np.uint8 and np.array_equal(out, want)


def test_reader_rejects_what_it_does_not_support(tmp_path):
    img = np.zeros((4, 4), np.uint8)
    p = str(tmp_path / "a.png")
    png_write(p, img)
    raw = bytearray(open(p, "rb").read())
    raw[-20] ^= 0xFF  # inside the IDAT payload
    (tmp_path / "bad.png").write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        png_read(str(tmp_path / "bad.png"))
    ihdr = struct.pack(">IIBBBBB", 4, 4, 16, 0, 0, 0, 0)  # 16-bit samples
    (tmp_path / "d16.png").write_bytes(MAGIC + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(bytes(4 * 9))) + _chunk(b"IEND", b""))
    with pytest.raises(ValueError):
        png_read(str(tmp_path / "d16.png"))
    (tmp_path / "n.png").write_bytes(b"not a png")
    with pytest.raises(ValueError):
        png_read(str(tmp_path / "n.png"))
