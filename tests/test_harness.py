"""The C harness (host/main.c) and the reference-shaped testbench (host/sim_tb_example.cpp): BASELINE config 1,
256x256 -> 512x512 RGB8 2x Lanczos-2, image file in -> image file out."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import patterns as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lanczos-hls_amd")


def _write_ppm(path, img):
    h, w, c = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(img).tobytes())


def _read_png(path):
    from PIL import Image
    return np.array(Image.open(path))


def _build():
    subprocess.run(["make", "-C", PKG, "--no-print-directory"], check=True, stdout=subprocess.DEVNULL)


def test_harness_without_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _build()
    src = tmp_path / "in.ppm"
    _write_ppm(str(src), P.gradient_noise(32, 32, 3))
    r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(src), str(tmp_path / "o.png")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Cannot use the GPU" in r.stdout
    # a missing input reproduces the reference's message (full_TB.h:110-113)
    r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(tmp_path / "nope.png"), str(tmp_path / "o.png")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Image was not loaded successfully." in r.stdout


@pytest.mark.gpu
def test_harness_config1_end_to_end(tmp_path):
    _build()
    img = P.gradient_noise(256, 256, 3)
    cfg = O.cfg(256, 256, 512, 512, 3, 2, 2, 1)
    want = O.expected_hwc_u8(cfg, img)
    src = tmp_path / "in.ppm"
    _write_ppm(str(src), img)
    # the plain C harness
    dst = tmp_path / "lanczos_upscale.png"
    r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(src), str(dst), "--scale", "2", "--a", "2", "--exact"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Scale:2/1, WIDTHS 256 -> 512" in r.stdout                # full_TB.h:124
    assert np.array_equal(_read_png(str(dst)), want)
    # the testbench-shaped caller: lanczos(stream_in, stream_out), "RMS err", expected + observed PNGs named as in
    # full_TB.h:166-177
    outdir = tmp_path / "img"
    outdir.mkdir()
    r = subprocess.run([os.path.join(PKG, "sim_tb_example"), str(src), str(outdir)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Scale:2/1, WIDTHS 256 -> 512" in r.stdout and "RMS err: " in r.stdout
    rms = float(r.stdout.split("RMS err: ")[1].split()[0])
    base = "256x256->512x512_2|1_2-"
    expected = _read_png(str(outdir / (base + "expected.png")))
    observed = _read_png(str(outdir / (base + "observed.png")))
    assert np.array_equal(expected, want)                             # the bit-exact mode IS lanczos_expected()
    diff = np.abs(observed.astype(int) - want.astype(int))
    assert observed.shape == want.shape and diff.max() <= 1
    assert abs(rms - float(np.sqrt((diff.astype(float) ** 2).mean()))) < 1e-3
    # the kernel.h call surface of hls_compat.hpp (kernel.h:2-8): lanczos_kernel(in, out, scale) and raw_lanczos_kernel(x) give the
    # software twin's double (full_TB.h:51-53), bit for bit
    import re
    m = re.search(r"lanczos_kernel\(0, 1, 2\) = (\S+) raw_lanczos_kernel\(0\.5\) = (\S+) raw_lanczos_kernel\(1\) = (\S+)", r.stdout)
    assert m, r.stdout
    want_k = O.lib().oracle_lanczos_kernel(0.5, 2)
    assert float(m.group(1)) == want_k and float(m.group(2)) == want_k
    assert 0 < abs(float(m.group(3))) < 1e-15          # sin(pi) in double is not 0


@pytest.mark.gpu
def test_harness_multi_device_options(tmp_path):
    """`--devices` path of the plain-C harness (lanczos_resample_multi_host): two contexts on device 0 stand in for two GPUs."""
    _build()
    img = P.gradient_noise(96, 128, 3)
    want = O.expected_hwc_u8(O.cfg(128, 96, 256, 192, 3, 3, 2, 1), img)
    src = tmp_path / "in.ppm"
    _write_ppm(str(src), img)
    for split in ("frames", "rows"):
        dst = tmp_path / f"multi_{split}.png"
        r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(src), str(dst), "--exact", "--devices", "0,0",
                            "--frames", "5", "--split", split], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"5 frames over 2 device(s), split by {split}" in r.stdout
        assert np.array_equal(_read_png(str(dst)), want)
    # --root: the batch resident on the first device, lanczos_resample_multi_root (one device: no exchange partner; with more
    # devices the RCCL scatter / gather runs -- unmeasured on this pool's one-GPU boxes), both rates printed separately
    for split in ("frames", "rows"):
        dst = tmp_path / f"root_{split}.png"
        r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(src), str(dst), "--exact", "--devices", "0", "--root",
                            "--frames", "3", "--split", split], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "compute only" in r.stdout and "root scatter + gather" in r.stdout and "3 frames resident on device 0" in r.stdout
        assert np.array_equal(_read_png(str(dst)), want)
    dst = tmp_path / "hls.png"                                         # --hls: the HLS-semantics mode through the harness
    r = subprocess.run([os.path.join(PKG, "lanczos_upscale"), str(src), str(dst), "--hls"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(_read_png(str(dst)), O.hls_expected_hwc(O.cfg(128, 96, 256, 192, 3, 3, 2, 1), img))
