"""ctypes access to the CPU checker (oracle/) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
(lanczos-hls_amd/).  Two libraries:
  * oracle/liblanczos_oracle.so  -- our restatement of full_TB.h:29-96 (travels to the GPU box)
  * oracle/_ref/ref_*.so         -- the reference's own lines, one build per compile-time shape
                                    (oracle/build_ref.sh; present when built in the container)
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblanczos_oracle.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


class OracleCfg(ctypes.Structure):
    _fields_ = [
        ("in_w", ctypes.c_int), ("in_h", ctypes.c_int),
        ("out_w", ctypes.c_int), ("out_h", ctypes.c_int),
        ("channels", ctypes.c_int), ("a", ctypes.c_int),
        ("scale_n", ctypes.c_int), ("scale_d", ctypes.c_int),
    ]


_lib = None


def build_oracle():
    """Compile oracle/liblanczos_oracle.so (gcc, seconds)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, n) for n in ("lanczos_oracle.c", "lanczos_hls_model.c", "lanczos_hls_model.h")]
        if (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
            build_oracle()
        L = ctypes.CDLL(ORACLE_SO)
        P = ctypes.POINTER(OracleCfg)
        L.oracle_lanczos_kernel.restype = ctypes.c_double
        L.oracle_lanczos_kernel.argtypes = [ctypes.c_double, ctypes.c_int]
        L.oracle_double_to_uint8.restype = ctypes.c_uint8
        L.oracle_double_to_uint8.argtypes = [ctypes.c_double]
        for name in ("oracle_expected_planar_u8", "oracle_expected_hwc_u8", "oracle_expected_hwc_u16"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [P, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_outofplace_hwc_u8.restype = ctypes.c_int
        L.oracle_outofplace_hwc_u8.argtypes = [P, ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_inplace_rows.restype = ctypes.c_int
        L.oracle_inplace_rows.argtypes = [P]
        L.oracle_fnv1a64.restype = ctypes.c_uint64
        L.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.oracle_lcg_fill_u8.restype = None
        L.oracle_lcg_fill_u8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        for name in ("oracle_hls_expected_hwc_u8", "oracle_hls_expected_hwc_u16"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [P, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_hls_rom.restype = ctypes.c_double
        L.oracle_hls_rom.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.oracle_hls_weight.restype = ctypes.c_double
        L.oracle_hls_weight.argtypes = [ctypes.c_int] * 5
        L.oracle_lcg_fill_u16.restype = None
        L.oracle_lcg_fill_u16.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
        _lib = L
    return _lib


def cfg(in_w, in_h, out_w, out_h, channels, a, scale_n, scale_d):
    return OracleCfg(in_w, in_h, out_w, out_h, channels, a, scale_n, scale_d)


def lcg_u8(n, seed=12345):
    buf = np.empty(n, dtype=np.uint8)
    lib().oracle_lcg_fill_u8(buf.ctypes.data, n, seed)
    return buf


def lcg_u16(n, seed=12345):
    buf = np.empty(n, dtype=np.uint16)
    lib().oracle_lcg_fill_u16(buf.ctypes.data, n, seed)
    return buf


def fnv1a64(arr):
    arr = np.ascontiguousarray(arr)
    return int(lib().oracle_fnv1a64(arr.ctypes.data, arr.nbytes))


def expected_planar_u8(c, img, threads=1):
    """img: uint8 [C][IN_H][IN_W] -> uint8 [C][OUT_H][OUT_W]  (full_TB.h:79-96)"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.shape == (c.channels, c.in_h, c.in_w)
    out = np.empty((c.channels, c.out_h, c.out_w), dtype=np.uint8)
    rc = lib().oracle_expected_planar_u8(ctypes.byref(c), img.ctypes.data, out.ctypes.data, threads)
    assert rc == 0, rc
    return out


def expected_hwc_u8(c, img, threads=1):
    """img: uint8 [IN_H][IN_W][C] (stb layout) -> uint8 [OUT_H][OUT_W][C]"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.shape == (c.in_h, c.in_w, c.channels), (img.shape, (c.in_h, c.in_w, c.channels))
    out = np.empty((c.out_h, c.out_w, c.channels), dtype=np.uint8)
    rc = lib().oracle_expected_hwc_u8(ctypes.byref(c), img.ctypes.data, out.ctypes.data, threads)
    assert rc == 0, rc
    return out


def expected_hwc_u16(c, img, threads=1):
    img = np.ascontiguousarray(img, dtype=np.uint16)
    assert img.shape == (c.in_h, c.in_w, c.channels)
    out = np.empty((c.out_h, c.out_w, c.channels), dtype=np.uint16)
    rc = lib().oracle_expected_hwc_u16(ctypes.byref(c), img.ctypes.data, out.ctypes.data, threads)
    assert rc == 0, rc
    return out


def hls_expected_hwc(c, img, threads=1, bit_precision=0):
    """oracle/lanczos_hls_model.c: the HLS pipeline's semantics (V-then-H, ROM weights, de-ring clamps, zero / repeat
    borders) in ideal arithmetic, or (bit_precision > 0, 8-bit samples) with the ap_fixed quantisation of lanczos.h:74-81.
    PARITY UNPINNED by the reference."""
    img = np.ascontiguousarray(img)
    assert img.shape == (c.in_h, c.in_w, c.channels) and img.dtype in (np.uint8, np.uint16)
    out = np.empty((c.out_h, c.out_w, c.channels), dtype=img.dtype)
    if bit_precision:
        assert img.dtype == np.uint8
        fx = lib().oracle_hls_expected_hwc_u8_fx
        fx.restype = ctypes.c_int
        fx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        rc = fx(ctypes.cast(ctypes.byref(c), ctypes.c_void_p), img.ctypes.data, out.ctypes.data, threads, bit_precision)
        assert rc == 0, rc
        return out
    fn = lib().oracle_hls_expected_hwc_u8 if img.dtype == np.uint8 else lib().oracle_hls_expected_hwc_u16
    rc = fn(ctypes.byref(c), img.ctypes.data, out.ctypes.data, threads)
    assert rc == 0, rc
    return out


def hls_rom(k, a, scale_n):
    return float(lib().oracle_hls_rom(k, a, scale_n))


def hls_weight(i, o, a, scale_n, scale_d):
    return float(lib().oracle_hls_weight(i, o, a, scale_n, scale_d))


def outofplace_hwc_u8(c, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((c.out_h, c.out_w, c.channels), dtype=np.uint8)
    rc = lib().oracle_outofplace_hwc_u8(ctypes.byref(c), img.ctypes.data, out.ctypes.data)
    assert rc == 0, rc
    return out


def inplace_rows(c):
    return int(lib().oracle_inplace_rows(ctypes.byref(c)))


# ---------------------------------------------------------------- reference builds (oracle/_ref)
def ref_so_path(in_w, in_h, out_w, out_h, scale_n, scale_d, a, channels):
    return os.path.join(
        REF_DIR, f"ref_{in_w}x{in_h}_{out_w}x{out_h}_{scale_n}-{scale_d}_a{a}_c{channels}.so")


def ref_configs():
    rows = []
    with open(os.path.join(ORACLE_DIR, "ref_configs.txt")) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            rows.append(tuple(int(t) for t in line.split()))
    return rows  # (iw, ih, ow, oh, sn, sd, a, c)


def ref_expected_planar_u8(c, img):
    """Run the reference's own compiled lanczos_expected (planar). Returns None if not built."""
    path = ref_so_path(c.in_w, c.in_h, c.out_w, c.out_h, c.scale_n, c.scale_d, c.a, c.channels)
    if not os.path.exists(path):
        return None
    R = ctypes.CDLL(path)
    R.ref_lanczos_expected.restype = None
    R.ref_lanczos_expected.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.shape == (c.channels, c.in_h, c.in_w)
    out = np.zeros((c.channels, c.out_h, c.out_w), dtype=np.uint8)  # static storage starts zeroed
    R.ref_lanczos_expected(img.ctypes.data, out.ctypes.data)
    return out
