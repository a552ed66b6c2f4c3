"""lanczos_hls_amd -- Python plumbing over the C ABI (include/lanczos_hip.h).

The product is liblanczos_hip.so (HIP kernels for gfx950 + the extern "C" boundary).  This module
only loads it with ctypes and mirrors the reference's call surface so tests and the benchmark read like
the reference's testbench:

    lanczos(img_hwc, scale_n, scale_d, a)   <- lanczos(stream_in, stream_out)   lanczos.h:121-126
    lanczos_kernel(x, a)                    <- double lanczos_kernel(double)    full_TB.h:51-53

There is NO CPU fallback: if the shared library is missing or no GPU is present the calls raise.
Nothing here imports anything under oracle/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LANCZOS_LIB: load another build of the same C ABI (A/B experiments); default is the in-tree product build
LIB_PATH = os.environ.get("LANCZOS_LIB") or os.path.join(_HERE, "liblanczos_hip.so")

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_RCCL = range(7)   # include/lanczos_hip.h:43-49
MODE_LSB1, MODE_EXACT, MODE_HLS = 0, 1, 2
KERNEL_NONE, KERNEL_GENERIC, KERNEL_FAST, KERNEL_HLS = 0, 1, 2, 3

# every symbol include/lanczos_hip.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = [
    "lanczos_desc_init", "lanczos_validate", "lanczos_inplace_rows", "lanczos_strip_input_rows",
    "lanczos_in_frame_bytes", "lanczos_out_frame_bytes", "lanczos_kernel", "lanczos_kernel_idx",
    "lanczos_taps_host", "lanczos_create", "lanczos_destroy", "lanczos_host_alloc", "lanczos_host_free",
    "lanczos_resample_host",
    "lanczos_resample_device", "lanczos_planar_to_interleaved_device", "lanczos_interleaved_to_planar_device",
    "lanczos_resample_planar_device", "lanczos_u8", "lanczos_timing_enable", "lanczos_timing_read",
    "lanczos_last_kernel", "lanczos_last_hip_error", "lanczos_force_kernel", "lanczos_strerror",
    "lanczos_version",
    "lanczos_partition_frames", "lanczos_partition_rows", "lanczos_multi_create", "lanczos_multi_destroy",
    "lanczos_multi_devices", "lanczos_resample_multi_host", "lanczos_resample_multi_root",
    "lanczos_multi_last_error", "lanczos_multi_exchange_plan", "lanczos_multi_exchange_selftest", "lanczos_device_alloc", "lanczos_device_free",
    "lanczos_device_copy",
]
SPLIT_FRAMES, SPLIT_ROWS = 0, 1


class LanczosError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = _lib().lanczos_strerror(code).decode() if _LIB is not None else str(code)
        super().__init__(f"{what}: {msg} (code {code})")


class Desc(ctypes.Structure):
    """lanczos_desc -- the run-time form of params.h (lanczos.h:9-31)."""
    _fields_ = [
        ("in_w", ctypes.c_int32), ("in_h", ctypes.c_int32),
        ("out_w", ctypes.c_int32), ("out_h", ctypes.c_int32),
        ("channels", ctypes.c_int32), ("bytes_per_sample", ctypes.c_int32),
        ("scale_n", ctypes.c_int32), ("scale_d", ctypes.c_int32),
        ("a", ctypes.c_int32), ("mode", ctypes.c_int32),
        ("out_row0", ctypes.c_int32), ("out_rows", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3),
    ]


_LIB = None


def build(verbose=False):
    """Compile liblanczos_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH) and not os.environ.get("LANCZOS_LIB"):
            try:  # a fresh checkout: compile the product (hipcc, ~40 s).  This is a build step, not a fallback.
                build()
            except Exception as e:
                raise RuntimeError(
                    f"{LIB_PATH} is missing and `make -C lanczos-hls_amd` failed ({e}); "
                    "there is no CPU fallback") from e
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C lanczos-hls_amd` "
                "(or __graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        PD = ctypes.POINTER(Desc)
        c_int, c_void_p, c_size_t, c_double = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double
        PI = ctypes.POINTER(c_int)
        L.lanczos_desc_init.argtypes = [PD] + [c_int] * 7
        L.lanczos_validate.argtypes = [PD]
        L.lanczos_inplace_rows.argtypes = [PD]
        L.lanczos_strip_input_rows.argtypes = [PD, c_int, c_int, PI, PI]
        L.lanczos_in_frame_bytes.argtypes = [PD]
        L.lanczos_in_frame_bytes.restype = c_size_t
        L.lanczos_out_frame_bytes.argtypes = [PD]
        L.lanczos_out_frame_bytes.restype = c_size_t
        L.lanczos_kernel.argtypes = [c_double, c_int]
        L.lanczos_kernel.restype = c_double
        L.lanczos_kernel_idx.argtypes = [c_int] * 5
        L.lanczos_kernel_idx.restype = c_double
        L.lanczos_taps_host.argtypes = [PD, c_int, c_void_p, c_void_p]
        L.lanczos_create.argtypes = [ctypes.POINTER(c_void_p), c_int]
        L.lanczos_destroy.argtypes = [c_void_p]
        L.lanczos_host_alloc.argtypes = [ctypes.POINTER(c_void_p), c_size_t]
        L.lanczos_host_free.argtypes = [c_void_p]
        L.lanczos_resample_host.argtypes = [c_void_p, PD, c_void_p, c_void_p, c_int]
        L.lanczos_resample_device.argtypes = [c_void_p, PD, c_void_p, c_void_p, c_int, c_size_t, c_size_t,
                                              c_void_p]
        L.lanczos_planar_to_interleaved_device.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]
        L.lanczos_interleaved_to_planar_device.argtypes = [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]
        L.lanczos_resample_planar_device.argtypes = [c_void_p, PD, c_void_p, c_void_p, c_int, c_void_p]
        L.lanczos_u8.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int]
        L.lanczos_timing_enable.argtypes = [c_void_p, c_int]
        L.lanczos_timing_read.argtypes = [c_void_p, PI, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]
        L.lanczos_last_kernel.argtypes = [c_void_p]
        L.lanczos_last_hip_error.argtypes = [c_void_p]
        L.lanczos_force_kernel.argtypes = [c_void_p, c_int]
        L.lanczos_strerror.argtypes = [c_int]
        L.lanczos_partition_frames.argtypes = [c_int, c_int, c_int, PI, PI]
        L.lanczos_partition_rows.argtypes = [PD, c_int, c_int, PI, PI, PI, PI]
        L.lanczos_multi_create.argtypes = [ctypes.POINTER(c_void_p), PI, c_int]
        L.lanczos_multi_destroy.argtypes = [c_void_p]
        L.lanczos_multi_devices.argtypes = [c_void_p]
        L.lanczos_resample_multi_host.argtypes = [c_void_p, PD, c_void_p, c_void_p, c_int, c_int]
        L.lanczos_resample_multi_root.argtypes = [c_void_p, PD, c_void_p, c_void_p, c_int, c_int,
                                                  ctypes.POINTER(c_double), ctypes.POINTER(c_double)]
        L.lanczos_strerror.restype = ctypes.c_char_p
        L.lanczos_version.argtypes = []
        L.lanczos_version.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc != OK:
        raise LanczosError(rc, what)


def make_desc(in_w, in_h, channels, scale_n, scale_d, a, bytes_per_sample=1, mode=MODE_LSB1,
              out_row0=0, out_rows=0, bit_precision=0):
    """bit_precision: BIT_PRECISION of the HLS mode's fixed-point emulation (lanczos_desc.reserved[0]; 0 = ideal arithmetic)."""
    d = Desc()
    _check(_lib().lanczos_desc_init(ctypes.byref(d), in_w, in_h, channels, bytes_per_sample,
                                    scale_n, scale_d, a), "lanczos_desc_init")
    d.mode = mode
    d.out_row0, d.out_rows = out_row0, out_rows
    d.reserved[0] = bit_precision
    _check(_lib().lanczos_validate(ctypes.byref(d)), "lanczos_validate")
    return d


def lanczos_kernel(x, a):
    """double lanczos_kernel(double x) of full_TB.h:51-53, LANCZOS_A as an argument."""
    return _lib().lanczos_kernel(float(x), int(a))


def lanczos_kernel_idx(in_idx, out_idx, scale_n, scale_d, a):
    """kernel_t lanczos_kernel(input_idx_t, output_idx_t, scale_t) of kernel.h:6 (double result)."""
    return _lib().lanczos_kernel_idx(in_idx, out_idx, scale_n, scale_d, a)


def inplace_rows(desc):
    return _lib().lanczos_inplace_rows(ctypes.byref(desc))


def strip_input_rows(desc, out_row0, out_rows):
    r0, n = ctypes.c_int(), ctypes.c_int()
    _check(_lib().lanczos_strip_input_rows(ctypes.byref(desc), out_row0, out_rows, ctypes.byref(r0),
                                           ctypes.byref(n)), "lanczos_strip_input_rows")
    return r0.value, n.value


def taps_host(desc, axis):
    n = desc.out_w if axis == 0 else desc.out_h
    first = np.empty(n, dtype=np.int32)
    w = np.empty((n, 2 * desc.a), dtype=np.float64)
    _check(_lib().lanczos_taps_host(ctypes.byref(desc), axis, first.ctypes.data, w.ctypes.data),
           "lanczos_taps_host")
    return first, w


class PinnedArray:
    """A numpy array over page-locked host memory (lanczos_host_alloc) -- lets lanczos_resample_host overlap
    its PCIe copies.  Keep the object alive while the array is in use."""

    def __init__(self, shape, dtype=np.uint8):
        self.shape = tuple(int(v) for v in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = ctypes.c_void_p()
        _check(_lib().lanczos_host_alloc(ctypes.byref(self._p), self.nbytes), "lanczos_host_alloc")
        buf = (ctypes.c_uint8 * self.nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def close(self):
        if self._p:
            self.array = None
            _lib().lanczos_host_free(self._p)
            self._p = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One lanczos_ctx: a device, a stream, resident tap tables.  Not shared between threads."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _check(_lib().lanczos_create(ctypes.byref(self._h), device), "lanczos_create")

    def close(self):
        if self._h:
            _lib().lanczos_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host buffers: the drop-in for lanczos(stream_in, stream_out) at full_TB.h:140
    def resample(self, img, scale_n, scale_d, a, mode=MODE_LSB1, out=None, bit_precision=0):
        """img: [H][W][C] (or [F][H][W][C]) uint8/uint16, stb interleaved layout -> scaled image(s).
        `out`: optional preallocated result (e.g. a PinnedArray's .array).  bit_precision: MODE_HLS only (make_desc)."""
        img = np.ascontiguousarray(img)
        batched = img.ndim == 4
        x = img if batched else img[None]
        if x.ndim != 4 or x.dtype not in (np.uint8, np.uint16):
            raise LanczosError(ERR_BAD_ARG, "resample: expected [H][W][C] uint8/uint16")
        f, h, w, c = x.shape
        d = make_desc(w, h, c, scale_n, scale_d, a, x.dtype.itemsize, mode, bit_precision=bit_precision)
        if out is None:
            out = np.empty((f, d.out_h, d.out_w, c), dtype=x.dtype)
        else:
            if out.dtype != x.dtype or out.size != f * d.out_h * d.out_w * c or not out.flags["C_CONTIGUOUS"]:
                raise LanczosError(ERR_BAD_ARG, "resample: `out` has the wrong size/dtype/layout")
            out = out.reshape((f, d.out_h, d.out_w, c))
        _check(_lib().lanczos_resample_host(self._h, ctypes.byref(d), x.ctypes.data, out.ctypes.data, f),
               "lanczos_resample_host")
        return out if batched else out[0]

    def resample_strip(self, strip_in, desc):
        """One row strip (desc.out_row0/out_rows set); strip_in holds the rows strip_input_rows() names."""
        strip_in = np.ascontiguousarray(strip_in)
        _, need = strip_input_rows(desc, desc.out_row0, desc.out_rows)
        if (strip_in.ndim != 3 or strip_in.shape[0] != need or strip_in.shape[1] != desc.in_w
                or strip_in.shape[2] != desc.channels or strip_in.dtype.itemsize != desc.bytes_per_sample):
            raise LanczosError(ERR_BAD_ARG, f"resample_strip: expected [{need}][{desc.in_w}][{desc.channels}] samples of "
                                            f"{desc.bytes_per_sample} byte(s), got {strip_in.shape} {strip_in.dtype}")
        out = np.empty((desc.out_rows, desc.out_w, desc.channels), dtype=strip_in.dtype)
        _check(_lib().lanczos_resample_host(self._h, ctypes.byref(desc), strip_in.ctypes.data,
                                            out.ctypes.data, 1), "lanczos_resample_host")
        return out

    def u8(self, img, out_w, out_h, a):
        """lanczos_u8: the reference's call shape (sizes as ints, scale = out_w/in_w reduced)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        out = np.empty((out_h, out_w, c), dtype=np.uint8)
        _check(_lib().lanczos_u8(self._h, img.ctypes.data, w, h, c, out.ctypes.data, out_w, out_h, a),
               "lanczos_u8")
        return out

    # -- device buffers (raw pointers, e.g. torch tensors' data_ptr()), asynchronous
    def resample_device(self, desc, d_in, d_out, frames, in_frame_stride=0, out_frame_stride=0, stream=None):
        _check(_lib().lanczos_resample_device(self._h, ctypes.byref(desc), d_in, d_out, frames,
                                              in_frame_stride, out_frame_stride, stream),
               "lanczos_resample_device")

    # -- planar frames [C][H][W] (the reference's img_in / img_out_ex arrays, full_TB.h:20-21), device pointers
    def planar_to_interleaved_device(self, d_planar, d_inter, w, h, channels, bytes_per_sample, frames, stream=None):
        _check(_lib().lanczos_planar_to_interleaved_device(self._h, d_planar, d_inter, w, h, channels,
                                                           bytes_per_sample, frames, stream),
               "lanczos_planar_to_interleaved_device")

    def interleaved_to_planar_device(self, d_inter, d_planar, w, h, channels, bytes_per_sample, frames, stream=None):
        _check(_lib().lanczos_interleaved_to_planar_device(self._h, d_inter, d_planar, w, h, channels,
                                                           bytes_per_sample, frames, stream),
               "lanczos_interleaved_to_planar_device")

    def resample_planar_device(self, desc, d_in_planar, d_out_planar, frames, stream=None):
        _check(_lib().lanczos_resample_planar_device(self._h, ctypes.byref(desc), d_in_planar, d_out_planar, frames,
                                                     stream), "lanczos_resample_planar_device")

    def timing_enable(self, on=True):
        _check(_lib().lanczos_timing_enable(self._h, 1 if on else 0), "lanczos_timing_enable")

    def timing_read(self):
        n, a, b = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _check(_lib().lanczos_timing_read(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b)),
               "lanczos_timing_read")
        return n.value, a.value, b.value

    def last_kernel(self):
        return _lib().lanczos_last_kernel(self._h)

    def force_kernel(self, family):
        _check(_lib().lanczos_force_kernel(self._h, family), "lanczos_force_kernel")


def partition_frames(frames, parts, part):
    f0, cnt = ctypes.c_int(), ctypes.c_int()
    _check(_lib().lanczos_partition_frames(frames, parts, part, ctypes.byref(f0), ctypes.byref(cnt)), "lanczos_partition_frames")
    return f0.value, cnt.value


def partition_rows(desc, parts, part):
    v = [ctypes.c_int() for _ in range(4)]
    _check(_lib().lanczos_partition_rows(ctypes.byref(desc), parts, part, *[ctypes.byref(x) for x in v]), "lanczos_partition_rows")
    return tuple(x.value for x in v)   # out_row0, out_rows, in_row0, in_rows


class Xfer(ctypes.Structure):
    """lanczos_xfer -- one message of the root exchange (lanczos_multi_exchange_plan)."""
    _fields_ = [("src", ctypes.c_int), ("dst", ctypes.c_int), ("src_off", ctypes.c_size_t), ("dst_off", ctypes.c_size_t),
                ("bytes", ctypes.c_size_t)]


def exchange_plan(desc, frames, split, n_devices, phase):
    """The scatter (phase 0) / gather (phase 1) of lanczos_resample_multi_root as a list of (src, dst, src_off, dst_off, bytes).
    Host only: no GPU, no RCCL."""
    fn = _lib().lanczos_multi_exchange_plan
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.POINTER(Desc), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(Xfer), ctypes.c_int]
    n = fn(ctypes.byref(desc), frames, split, n_devices, phase, None, 0)
    if n < 0:
        raise LanczosError(-n, "lanczos_multi_exchange_plan")
    arr = (Xfer * max(n, 1))()
    n2 = fn(ctypes.byref(desc), frames, split, n_devices, phase, arr, n)
    assert n2 == n
    return [(x.src, x.dst, x.src_off, x.dst_off, x.bytes) for x in arr[:n]]


class MultiContext:
    """lanczos_multi: one context per device of one node (frames or row strips split over them, no data-path collective)."""

    def __init__(self, devices):
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_int * len(devices))(*devices)
        _check(_lib().lanczos_multi_create(ctypes.byref(self._h), arr, len(devices)), "lanczos_multi_create")

    def close(self):
        if self._h:
            _lib().lanczos_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resample(self, frames_hwc, scale_n, scale_d, a, mode=MODE_LSB1, split=SPLIT_FRAMES):
        x = np.ascontiguousarray(frames_hwc)
        f, h, w, c = x.shape
        d = make_desc(w, h, c, scale_n, scale_d, a, x.dtype.itemsize, mode)
        out = np.empty((f, d.out_h, d.out_w, c), dtype=x.dtype)
        _check(_lib().lanczos_resample_multi_host(self._h, ctypes.byref(d), x.ctypes.data, out.ctypes.data, f, split),
               "lanczos_resample_multi_host")
        return out

    def resample_root(self, desc, d_in_root, d_out_root, frames, split=SPLIT_FRAMES):
        """Device pointers on devices[0]; returns (compute_ms, total_ms)."""
        cm, tm = ctypes.c_double(), ctypes.c_double()
        _check(_lib().lanczos_resample_multi_root(self._h, ctypes.byref(desc), d_in_root, d_out_root, frames, split,
                                                  ctypes.byref(cm), ctypes.byref(tm)), "lanczos_resample_multi_root")
        return cm.value, tm.value


    def exchange_selftest(self, messages=4, nbytes=1 << 20, fail_at=-1):
        """The RCCL exchange machinery on one rank (lanczos_multi_exchange_selftest): returns the C status code and, for
        ERR_RCCL, (rccl_error, message index) from lanczos_multi_last_error."""
        lib = _lib()
        lib.lanczos_multi_exchange_selftest.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
        rc = lib.lanczos_multi_exchange_selftest(self._h, messages, nbytes, fail_at)
        he, re_, at = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.lanczos_multi_last_error(self._h, ctypes.byref(he), ctypes.byref(re_), ctypes.byref(at))
        return rc, re_.value, at.value


_default_ctx = None


def lanczos(img, scale_n, scale_d=1, a=3, mode=MODE_LSB1):
    """Module-level convenience mirroring the reference entry point: whole frame in, whole frame out."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx.resample(img, scale_n, scale_d, a, mode)
