"""The CPU checker itself: our restatement (oracle/) against the reference's own outputs.

Pins, strongest first:
  1. tests/golden/golden_small.npz  -- raw outputs of the reference's compiled software path
     (oracle/_ref, full_TB.h:29-96) for small shapes x {noise, dark} inputs
  2. tests/golden/kat_digests.json  -- FNV-1a-64 digests of the reference output for medium/large
     shapes, including the five digests recorded in SURVEY.md 8(c)
  3. oracle/_ref live, when it has been built in this container (skipped on the GPU box)
"""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _parse(key):
    dims, out, sc, a, c = key.split("_")
    iw, ih = (int(v) for v in dims.split("x"))
    ow, oh = (int(v) for v in out.split("x"))
    sn, sd = (int(v) for v in sc.split("-"))
    return iw, ih, ow, oh, sn, sd, int(a[1:]), int(c[1:])


def _small():
    z = np.load(os.path.join(GOLD, "golden_small.npz"))
    keys = sorted({k.rsplit(":", 1)[0] for k in z.files})
    return z, keys


def test_small_fixtures_bit_exact():
    z, keys = _small()
    assert len(keys) >= 16
    for k in keys:
        shape, pname = k.split(":")
        iw, ih, ow, oh, sn, sd, a, c = _parse(shape)
        cfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
        got = O.expected_planar_u8(cfg, z[k + ":in"])
        assert np.array_equal(got, z[k + ":out"]), k
        # the interleaved (stb layout) entry computes the same numbers
        hwc = O.expected_hwc_u8(cfg, np.ascontiguousarray(z[k + ":in"].transpose(1, 2, 0)))
        assert np.array_equal(hwc.transpose(2, 0, 1), z[k + ":out"]), k
        # so does the threaded variant used as the multi-core CPU baseline
        thr = O.expected_planar_u8(cfg, z[k + ":in"], threads=3)
        assert np.array_equal(thr, z[k + ":out"]), k


def _digest_cases():
    with open(os.path.join(GOLD, "kat_digests.json")) as f:
        d = json.load(f)
    return d


@pytest.mark.parametrize("pname", ["noise", "dark"])
def test_medium_digests(pname):
    d = _digest_cases()["digests"]
    seen = 0
    for k, want in d.items():
        shape, p = k.split(":")
        iw, ih, ow, oh, sn, sd, a, c = _parse(shape)
        if p != pname or ow * oh > 1000 * 600:
            continue
        cfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
        n = c * ih * iw
        img = (O.lcg_u8(n, 12345) if p == "noise"
               else (O.lcg_u8(n, 777).astype(np.uint16) * 72 // 256).astype(np.uint8)).reshape(c, ih, iw)
        got = O.fnv1a64(O.expected_planar_u8(cfg, img, threads=4))
        assert f"{got:016x}" == want, k
        seen += 1
    assert seen >= 10


def test_survey_kat_full_size():
    """The 1080p->4K and 720p->4K known answers of SURVEY.md 8(c) (BASELINE configs 2 and 3)."""
    d = _digest_cases()
    for shape in ("1920x1080_3840x2160_2-1_a3_c3", "1280x720_3840x2160_3-1_a3_c3"):
        iw, ih, ow, oh, sn, sd, a, c = _parse(shape)
        cfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
        img = O.lcg_u8(c * ih * iw, 12345).reshape(c, ih, iw)
        got = f"{O.fnv1a64(O.expected_planar_u8(cfg, img, threads=8)):016x}"
        assert got == d["survey_8c"][shape] == d["digests"][shape + ":noise"]


def test_against_live_reference_build():
    cases = [r for r in O.ref_configs() if r[2] * r[3] <= 1000 * 600]
    built = [r for r in cases if os.path.exists(O.ref_so_path(*r))]
    if not built:
        pytest.skip("oracle/_ref not built here (make -C oracle ref needs /root/reference)")
    rng = np.random.default_rng(7)
    for (iw, ih, ow, oh, sn, sd, a, c) in built:
        cfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
        for img in (rng.integers(0, 256, (c, ih, iw), dtype=np.uint8),
                    rng.integers(0, 40, (c, ih, iw), dtype=np.uint8),
                    np.full((c, ih, iw), 255, np.uint8),
                    ((np.add.outer(np.arange(ih), np.arange(iw)) // 4 % 2) * 255).astype(np.uint8)[None]
                    .repeat(c, 0)):
            ref = O.ref_expected_planar_u8(cfg, img)
            got = O.expected_planar_u8(cfg, img)
            assert np.array_equal(got, ref), (iw, ih, ow, oh, sn, sd, a, c)


def test_double_to_uint8_truncates():
    L = O.lib()
    # full_TB.h:29-37: clamp, then C cast (truncate toward zero) -- not rounding
    for x, want in [(0.0, 0), (0.999, 0), (1.0, 1), (254.9999, 254), (255.0, 255), (255.5, 255),
                    (300.0, 255), (-0.5, 0), (-7.0, 0), (127.5, 127), (16.99999999, 16)]:
        assert L.oracle_double_to_uint8(x) == want, x


def test_kernel_formula():
    L = O.lib()
    for a in (2, 3, 4):
        assert L.oracle_lanczos_kernel(0.0, a) == 1.0  # sinc(0) = 1 (full_TB.h:40-42)
        for x in (0.5, -0.5, 1.5, 2.5, 0.3333333333333333, 1.25):
            want = (math.sin(math.pi * x) / (math.pi * x)) * (math.sin(math.pi * x / a) / (math.pi * x / a))
            assert L.oracle_lanczos_kernel(x, a) == want
        # integer offsets are NOT exactly zero in double and there is no |x|<a window
        assert 0 < abs(L.oracle_lanczos_kernel(1.0, a)) < 1e-16
        assert abs(L.oracle_lanczos_kernel(float(-a), a)) < 1e-30


def test_inplace_quirk_rows():
    """full_TB.h:67-77 overwrites rows it later reads: only rows < K differ from a clean V pass."""
    rng = np.random.default_rng(3)
    for (iw, ih, sn, sd, a, wantK) in [(32, 24, 2, 1, 2, 3), (32, 24, 2, 1, 3, 5), (32, 24, 3, 1, 3, 4),
                                        (32, 24, 2, 1, 4, 7), (36, 30, 4, 3, 3, 9)]:
        ow, oh = iw * sn // sd, ih * sn // sd
        cfg = O.cfg(iw, ih, ow, oh, 3, a, sn, sd)
        assert O.inplace_rows(cfg) == wantK
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        inp = O.expected_hwc_u8(cfg, img)
        oop = O.outofplace_hwc_u8(cfg, img)
        assert np.array_equal(inp[wantK:], oop[wantK:])
        assert not np.array_equal(inp[:wantK], oop[:wantK])


def test_u16_generalisation_matches_u8_semantics():
    """parity unpinned by the reference (no 16-bit path exists): the u16 entry is the same loop
    with clamp 65535; on inputs < 256 whose sums never reach 255 it must reproduce the u8 result."""
    rng = np.random.default_rng(5)
    cfg = O.cfg(20, 14, 40, 28, 4, 4, 2, 1)
    img = rng.integers(0, 200, (14, 20, 4), dtype=np.uint8)
    a8 = O.expected_hwc_u8(cfg, img)
    a16 = O.expected_hwc_u16(cfg, img.astype(np.uint16))
    ok = a8 < 255
    assert np.array_equal(a16[ok], a8[ok].astype(np.uint16))
    assert (a16[~ok] >= 255).all()
