"""Asset decoders / writers behind Texture2D (csrc/image_io.cpp; reference: texture_2d.cpp:22-44 via stb_image).
PNG decoding is checked against PIL, Radiance RGBE against files encoded here (flat and new-style RLE) with the
expected floats computed by stb's rule rgb * 2^(e - 136)."""
import io
import os

import numpy as np
import pytest
from PIL import Image


def _rgbe_encode(rgb):
    rgb = np.asarray(rgb, np.float64)
    m = rgb.max(axis=-1)
    e = np.zeros(m.shape, np.int64)
    nz = m > 1e-32
    e[nz] = np.floor(np.log2(m[nz])).astype(np.int64) + 1
    scale = np.where(nz, 256.0 / np.exp2(e), 0.0)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(nz, e + 128, 0)
    return out


def _rle_row(row):
    """new-style RLE: 2 2 hi lo, then each channel separately (runs > 128, literals <= 128)."""
    w = row.shape[0]
    out = bytearray([2, 2, w >> 8, w & 255])
    for k in range(4):
        ch = row[:, k]
        i = 0
        while i < w:
            run = 1
            while i + run < w and run < 127 and ch[i + run] == ch[i]:
                run += 1
            if run >= 4:
                out += bytes([128 + run, int(ch[i])]); i += run
            else:
                j = i
                while j < w and j - i < 128:
                    r = 1
                    while j + r < w and r < 4 and ch[j + r] == ch[j]:
                        r += 1
                    if r >= 4:
                        break
                    j += 1
                out += bytes([j - i]) + bytes(int(v) for v in ch[i:j]); i = j
    return bytes(out)


def _hdr_file(rgbe, rle):
    h, w = rgbe.shape[:2]
    head = b"#?RADIANCE\n# made by test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n" + f"-Y {h} +X {w}\n".encode()
    body = b"".join(_rle_row(rgbe[y]) for y in range(h)) if rle else rgbe.tobytes()
    return head + body


def _expected_float(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    f = np.where(e != 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    out = np.ones(rgbe.shape[:2] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f[..., None]
    return out


@pytest.mark.parametrize("w,h,rle", [(40, 7, True), (40, 7, False), (5, 3, False), (300, 4, True)])
def test_hdr_decode(vrt, tmp_path, w, h, rle):
    rng = np.random.default_rng(w * 100 + h)
    rgb = rng.random((h, w, 3)) * np.exp2(rng.integers(-6, 10, (h, w, 1)))
    rgb[0, :w // 2] = rgb[0, 0]                       # long runs for the RLE path
    rgb[1, 1] = 0.0                                   # e = 0 pixel
    rgbe = _rgbe_encode(rgb)
    p = tmp_path / "sky.hdr"
    p.write_bytes(_hdr_file(rgbe, rle))
    got = vrt.load_image(p)
    assert got.dtype == np.float32 and got.shape == (h, w, 4)
    assert (got == _expected_float(rgbe)).all()


@pytest.mark.parametrize("mode,bits", [("RGBA", 8), ("RGB", 8), ("L", 8), ("LA", 8), ("P", 8), ("I;16", 16), ("1", 1)])
def test_png_decode_matches_pil(vrt, tmp_path, mode, bits):
    rng = np.random.default_rng(7)
    w, h = 37, 23
    if mode == "P":
        im = Image.fromarray(rng.integers(0, 200, (h, w), dtype=np.uint8), "P")
        im.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tobytes())
    elif mode == "I;16":
        im = Image.fromarray(rng.integers(0, 65536, (h, w)).astype(np.uint16))
    elif mode == "1":
        im = Image.fromarray((rng.random((h, w)) > 0.5)).convert("1")
    else:
        ch = {"RGBA": 4, "RGB": 3, "L": 1, "LA": 2}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        im = Image.fromarray(a[..., 0] if ch == 1 else a, mode)
    p = tmp_path / "img.png"
    im.save(p)
    got = vrt.load_image(p)
    assert got.dtype == np.uint8 and got.shape == (h, w, 4)
    if mode == "I;16":
        g = (np.array(Image.open(p)).astype(np.uint16) >> 8).astype(np.uint8)        # stb keeps the high byte
        exp = np.dstack([g, g, g, np.full_like(g, 255)])
    else:
        exp = np.array(Image.open(p).convert("RGBA"))
    assert (got == exp).all()


def test_writers_round_trip(vrt, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (19, 31, 4), dtype=np.uint8)
    vrt.write_image(tmp_path / "o.png", img)
    assert (np.array(Image.open(tmp_path / "o.png").convert("RGBA")) == img).all()
    assert (vrt.load_image(tmp_path / "o.png") == img).all()
    vrt.write_image(tmp_path / "o.ppm", img)
    assert (np.array(Image.open(tmp_path / "o.ppm")) == img[..., :3]).all()
    f = rng.normal(size=(5, 6, 3)).astype(np.float32)
    vrt.write_image(tmp_path / "o.pfm", f)
    raw = (tmp_path / "o.pfm").read_bytes()
    assert raw.startswith(b"PF\n6 5\n-1.0\n")
    back = np.frombuffer(raw[len(b"PF\n6 5\n-1.0\n"):], np.float32).reshape(5, 6, 3)[::-1]
    assert (back == f).all()


def test_load_failures(vrt, tmp_path):
    with pytest.raises(RuntimeError, match="Could not load image"):
        vrt.load_image(tmp_path / "missing.png")
    bad = tmp_path / "bad.png"; bad.write_bytes(b"\x89PNG\r\n\x1a\n" + b"\x00" * 40)
    with pytest.raises(RuntimeError, match="Could not load image"):
        vrt.load_image(bad)
    trunc = tmp_path / "t.hdr"; trunc.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 16\n\x02\x02\x00\x10")
    with pytest.raises(RuntimeError, match="Could not load image"):
        vrt.load_image(trunc)
