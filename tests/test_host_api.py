"""CPU-only checks of the product's host side: the C ABI loads and exports what include/lanczos_hip.h
declares, descriptor validation mirrors the harness's EXIT_FAILURE checks (full_TB.h:110-123), and the
host tap tables hold exactly the software model's weights (full_TB.h:51-60)."""
import ctypes
import os
import re

import numpy as np
import pytest

import lanczos_hls_amd as L
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lanczos_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lanczos_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.ABI_SYMBOLS), declared ^ set(L.ABI_SYMBOLS)
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in L._lib().lanczos_version()


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(L.LanczosError) as e:
        L.Context(0)
    assert e.value.code == L.ERR_NO_DEVICE
    with pytest.raises(L.LanczosError):
        L.lanczos(np.zeros((8, 8, 3), np.uint8), 2, 1, 3)


def test_validation_codes():
    lib = L._lib()
    d = L.make_desc(1920, 1080, 3, 2, 1, 3)
    assert (d.out_w, d.out_h) == (3840, 2160)
    d = L.make_desc(300, 200, 3, 4, 3, 3)
    assert (d.out_w, d.out_h) == (400, 266)      # OUT = IN*N/D, integer division
    d = L.make_desc(100, 50, 3, 6, 4, 3)           # reduced by gcd like lanczos.h:110
    assert (d.scale_n, d.scale_d) == (3, 2)

    def code(**kw):
        args = dict(in_w=64, in_h=48, channels=3, bps=1, sn=2, sd=1, a=3)
        args.update(kw)
        dd = L.Desc()
        return lib.lanczos_desc_init(ctypes.byref(dd), args["in_w"], args["in_h"], args["channels"],
                                     args["bps"], args["sn"], args["sd"], args["a"])

    assert code() == L.OK
    assert code(channels=2) == L.ERR_BAD_ARG      # full_TB.h:120-123
    assert code(channels=5) == L.ERR_BAD_ARG
    assert code(a=1) == L.ERR_BAD_ARG
    assert code(a=5) == L.ERR_BAD_ARG
    assert code(in_w=0) == L.ERR_BAD_ARG          # full_TB.h:115-118
    assert code(bps=3) == L.ERR_BAD_ARG
    assert code(sn=0) == L.ERR_BAD_ARG
    assert code(sn=1, sd=1) == L.OK               # S == 1: one in-place recurrence per column, any frame height
    assert code(sn=2, sd=3) == L.ERR_UNSUPPORTED  # S < 1: the reference itself writes out of bounds (full_TB.h:85)
    assert code(sn=1, sd=2) == L.ERR_UNSUPPORTED
    # a tampered descriptor (wrong output size) is rejected like a wrong-size image
    d = L.make_desc(64, 48, 3, 2, 1, 3)
    d.out_w += 1
    assert lib.lanczos_validate(ctypes.byref(d)) == L.ERR_BAD_ARG
    d = L.make_desc(64, 48, 3, 2, 1, 3)
    d.out_row0, d.out_rows = 90, 10
    assert lib.lanczos_validate(ctypes.byref(d)) == L.ERR_BAD_ARG
    for i in (1, 2):                              # reserved[1..2] must be zero (hand-built descriptors: zero-initialise)
        d = L.make_desc(64, 48, 3, 2, 1, 3)
        d.reserved[i] = 1
        assert lib.lanczos_validate(ctypes.byref(d)) == L.ERR_BAD_ARG
    assert lib.lanczos_validate(None) == L.ERR_BAD_ARG
    assert b"bad argument" in lib.lanczos_strerror(L.ERR_BAD_ARG)
    assert L.ERR_RCCL == 6 and b"RCCL" in lib.lanczos_strerror(L.ERR_RCCL)


def test_kernel_twins_match_software_model():
    OL = O.lib()
    for a in (2, 3, 4):
        for x in (0.0, 0.5, -0.5, 1.0, 2.0, -3.0, 1 / 3, 2.6666666666666665, 1000.3333333333334 - 1000):
            assert L.lanczos_kernel(x, a) == OL.oracle_lanczos_kernel(x, a)
    # index form: x = out/SCALE - in  (full_TB.h:57,60)
    for (sn, sd) in ((2, 1), (3, 1), (4, 3)):
        for out_idx in (0, 1, 7, 1001):
            x = out_idx / (sn / sd)
            for in_idx in range(int(np.floor(x)) - 2, int(np.floor(x)) + 4):
                assert L.lanczos_kernel_idx(in_idx, out_idx, sn, sd, 3) == OL.oracle_lanczos_kernel(x - in_idx, 3)


@pytest.mark.parametrize("shape", [(1920, 1080, 2, 1, 3), (1280, 720, 3, 1, 3), (300, 200, 4, 3, 3),
                                   (480, 270, 2, 1, 4), (256, 256, 2, 1, 2)])
def test_tap_tables_are_the_reference_loop(shape):
    iw, ih, sn, sd, a = shape
    d = L.make_desc(iw, ih, 3, sn, sd, a)
    OL = O.lib()
    SCALE = sn / sd
    for axis, in_n, out_n in ((0, iw, d.out_w), (1, ih, d.out_h)):
        first, w = L.taps_host(d, axis)
        rng = np.random.default_rng(axis)
        for o in list(range(0, 12)) + list(range(out_n - 12, out_n)) + list(rng.integers(0, out_n, 200)):
            x = o / SCALE
            fl = int(np.floor(x))
            assert first[o] == fl - a + 1
            lo, hi = max(0, fl - a + 1), min(in_n - 1, fl + a)   # full_TB.h:59
            for k in range(2 * a):
                i = first[o] + k
                want = OL.oracle_lanczos_kernel(x - i, a) if lo <= i <= hi else 0.0
                assert w[o, k] == want, (axis, o, k)


def test_inplace_rows_and_strips():
    for (iw, ih, sn, sd, a) in [(64, 48, 2, 1, 2), (64, 48, 2, 1, 3), (64, 48, 3, 1, 3), (64, 48, 2, 1, 4),
                                (60, 48, 4, 3, 3), (64, 48, 3, 2, 3)]:
        d = L.make_desc(iw, ih, 3, sn, sd, a)
        oc = O.cfg(iw, ih, d.out_w, d.out_h, 3, a, sn, sd)
        assert L.inplace_rows(d) == O.inplace_rows(oc)
    d = L.make_desc(3840, 2160, 4, 2, 1, 4, bytes_per_sample=2)   # BASELINE config 5
    assert (d.out_w, d.out_h) == (7680, 4320)
    rows = 4320 // 8
    for g in range(8):
        r0, n = L.strip_input_rows(d, g * rows, rows)
        assert r0 == max(0, g * rows // 2 - 3)
        assert r0 + n - 1 == min(2159, (g * rows + rows - 1) // 2 + 4)
        assert n <= rows // 2 + 8
