"""host/image_io.c -- the harness's image decoder / encoder (the reference's harness uses stb_image, stb.cpp:1-6,
full_TB.h:107,172).  Own codec: PNG (all bit depths, palette, Adam7), BMP, PNM -- checked here against Pillow's decode of
files Pillow wrote.  Where the reference tree is present (the build container) the SAME tool is also built on the
reference's stb headers (make convert_stb: -DLANCZOS_WITH_STB -I<reference>/stb_image, nothing copied) and must agree with
the own codec byte for byte; JPEG, which only stb decodes, is checked against Pillow within JPEG's decoder tolerance."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lanczos-hls_amd")
STB_DIR = "/root/reference/LanczosUpscaler/stb_image"

PIL = pytest.importorskip("PIL.Image")


def _tool(stb=False):
    target = "convert_stb" if stb else "convert"
    subprocess.run(["make", "-C", PKG, target, "--no-print-directory"], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return os.path.join(PKG, "build", "image_convert_stb" if stb else "image_convert")


def _decode(tool, src, tmp_path, channels=3):
    out = str(tmp_path / ("dec_%d.p%sm" % (channels, "g" if channels == 1 else "p")))
    r = subprocess.run([tool, src, out, str(channels)], capture_output=True, text=True)
    if r.returncode != 0:
        return None
    with open(out, "rb") as f:
        magic = f.readline()
        w, h = (int(v) for v in f.readline().split())
        assert f.readline().strip() == b"255" and magic[:2] in (b"P5", b"P6")
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, channels)


def _files(tmp_path):
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (19, 31, 4), dtype=np.uint8)
    g16 = rng.integers(0, 65536, (23, 29), dtype=np.uint16)
    out = {}
    PIL.fromarray(rgb).save(tmp_path / "rgb8.png")
    out["rgb8.png"] = rgb
    PIL.fromarray(rgb).save(tmp_path / "rgb8_adam7.png", interlace=1) if False else None
    PIL.fromarray(rgba).save(tmp_path / "rgba8.png")
    out["rgba8.png"] = rgba[..., :3]
    PIL.fromarray(g16).save(tmp_path / "gray16.png")
    out["gray16.png"] = np.repeat((g16 >> 8).astype(np.uint8)[..., None], 3, axis=2)      # stb keeps the high byte
    pal = PIL.fromarray(rgb).quantize(17)
    pal.save(tmp_path / "pal.png")
    out["pal.png"] = np.array(pal.convert("RGB"))
    bw = PIL.fromarray((rgb[..., 0] > 127).astype(np.uint8) * 255).convert("1")
    bw.save(tmp_path / "gray1.png")
    out["gray1.png"] = np.repeat(np.array(bw.convert("L"))[..., None], 3, axis=2)
    PIL.fromarray(rgb).save(tmp_path / "rgb24.bmp")
    out["rgb24.bmp"] = rgb
    PIL.fromarray(rgb[..., 0]).save(tmp_path / "gray8.bmp")
    out["gray8.bmp"] = np.repeat(rgb[..., :1], 3, axis=2)
    PIL.fromarray(rgba).save(tmp_path / "rgba32.bmp")
    out["rgba32.bmp"] = rgba[..., :3]
    return out


def _write_adam7(path, img):
    """Pillow cannot WRITE interlaced PNGs: assemble one by hand (zlib + filter type 0), the seven Adam7 passes."""
    import struct
    import zlib
    h, w, c = img.shape
    raw = b""
    for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = img[y0::dy, x0::dx]
        if sub.size == 0:
            continue
        for row in sub:
            raw += b"\x00" + row.tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_own_codec_decodes_what_pillow_wrote(tmp_path):
    tool = _tool()
    for name, want in _files(tmp_path).items():
        got = _decode(tool, str(tmp_path / name), tmp_path)
        assert got is not None, name
        assert np.array_equal(got, want), name
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (21, 34, 3), dtype=np.uint8)
    _write_adam7(str(tmp_path / "adam7.png"), img)
    assert np.array_equal(np.array(PIL.open(tmp_path / "adam7.png")), img)           # the hand-made file is a valid PNG
    assert np.array_equal(_decode(tool, str(tmp_path / "adam7.png"), tmp_path), img)
    g = _decode(tool, str(tmp_path / "rgb8.png"), tmp_path, channels=1)               # stb's luma conversion
    rgb = _files(tmp_path)["rgb8.png"].astype(int)
    assert np.array_equal(g[..., 0], (rgb[..., 0] * 77 + rgb[..., 1] * 150 + rgb[..., 2] * 29) >> 8)
    # the own encoder's files are valid PNGs for Pillow
    out = str(tmp_path / "roundtrip.png")
    assert subprocess.run([tool, str(tmp_path / "rgb24.bmp"), out, "3"]).returncode == 0
    assert np.array_equal(np.array(PIL.open(out)), _files(tmp_path)["rgb8.png"])
    # JPEG is not in the own codec: refused cleanly (the stb build takes it)
    PIL.fromarray(_files(tmp_path)["rgb8.png"]).save(tmp_path / "x.jpg", quality=95)
    assert _decode(tool, str(tmp_path / "x.jpg"), tmp_path) is None


@pytest.mark.skipif(not os.path.exists(os.path.join(STB_DIR, "stb_image.h")), reason="reference tree (stb headers) not present")
def test_stb_build_agrees_with_the_own_codec_and_takes_jpeg(tmp_path):
    own, stb = _tool(), _tool(stb=True)
    files = _files(tmp_path)
    img = np.random.default_rng(9).integers(0, 256, (21, 34, 3), dtype=np.uint8)
    _write_adam7(str(tmp_path / "adam7.png"), img)
    for name in list(files) + ["adam7.png"]:
        a, b = _decode(own, str(tmp_path / name), tmp_path), _decode(stb, str(tmp_path / name), tmp_path)
        assert a is not None and b is not None and np.array_equal(a, b), name
    smooth = np.clip(np.add.outer(np.arange(48) * 4, np.arange(64) * 3)[..., None] + np.array([0, 20, 40]), 0, 255).astype(np.uint8)
    PIL.fromarray(smooth).save(tmp_path / "x.jpg", quality=95)
    got = _decode(stb, str(tmp_path / "x.jpg"), tmp_path)
    assert got is not None and np.abs(got.astype(int) - np.array(PIL.open(tmp_path / "x.jpg")).astype(int)).max() <= 4
    # PNG written through stbi_write_png, read back by Pillow
    out = str(tmp_path / "stb_written.png")
    assert subprocess.run([stb, str(tmp_path / "rgb24.bmp"), out, "3"]).returncode == 0
    assert np.array_equal(np.array(PIL.open(out)), files["rgb8.png"])
