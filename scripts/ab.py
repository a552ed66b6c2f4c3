#!/usr/bin/env python3
"""scripts/ab.py -- interleaved A/B of several builds of liblanczos_hip.so in ONE process on ONE device
(cdna_hip_programming.md 5.4 rule 24: perf deltas come from interleaved rounds in one process).

    python3 scripts/ab.py [--config c2] [--patterns gradient,noise,blocks] [--rounds 7] [--steps 20] [--frames F]
                          [--mode lsb1|exact] [--env NAME=V ...] lib1.so lib2.so ...

Every library is loaded with its own ctypes handle (same C ABI), gets its own context, and runs `steps` whole steps per
round on torch's current stream, timed with torch.cuda events (device time of the whole step: every kernel + gaps).
Prints per library / pattern: median, min and all rounds in us, and the HBM-roofline fraction of the median.
Since round 3 the library reads its environment switches ONCE per process (first lanczos_create, csrc/lanczos_env.hpp): set them
for the whole run (`LANCZOS_X=1 python3 scripts/ab.py ...`), one process per setting.  A "path.so@NAME=V" library spec (per-library
environment overrides, rounds 1-2) is refused: it would report two identical runs as two variants.
"""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--patterns", default="gradient,noise,blocks")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--mode", default="lsb1")
    ap.add_argument("--check", action="store_true", help="compare every library's output with the first one's")
    ap.add_argument("--rotate", type=int, default=1, help="cycle through this many distinct input batches (and output buffers) step by step")
    ap.add_argument("--rotate-what", default="both", choices=["both", "in", "out"], help="which side --rotate cycles")
    args = ap.parse_args()

    import torch
    import bench
    import lanczos_hls_amd as L

    cfg = bench.CONFIGS[args.config]
    iw, ih, c, bps, sn, sd, a, _ = cfg
    frames = args.frames or (4 if args.config == "c5" else 16)
    dev = torch.device("cuda", 0)
    mode = L.MODE_EXACT if args.mode == "exact" else L.MODE_LSB1
    d = L.make_desc(iw, ih, c, sn, sd, a, bps, mode)   # the product library's host helper (same struct for every build)

    handles = []
    for spec in args.libs:
        path, _, envs = spec.partition("@")
        if envs:
            sys.exit(f"ab.py: '{spec}': per-library environment overrides no longer select behaviour (the library reads its "
                     f"environment once per process) -- set {envs} for the whole run, one process per setting")
        if len(set(os.path.abspath(s) for s in args.libs)) != len(args.libs):
            sys.exit("ab.py: the same library twice is not an A/B")
        env = {}
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.lanczos_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        lib.lanczos_destroy.argtypes = [ctypes.c_void_p]
        lib.lanczos_resample_device.argtypes = [ctypes.c_void_p, ctypes.POINTER(L.Desc), ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        h = ctypes.c_void_p()
        assert lib.lanczos_create(ctypes.byref(h), 0) == 0
        handles.append((spec, lib, h, env))

    y = torch.empty((frames, d.out_h, d.out_w, c), device=dev, dtype=torch.uint8 if bps == 1 else torch.int16)
    stream = torch.cuda.current_stream().cuda_stream
    alg = frames * (iw * ih + d.out_w * d.out_h) * c * bps

    all_keys = {k for _s, _l, _h, env in handles for k in env}

    ys = [y] + [torch.empty_like(y) for _ in range(args.rotate - 1)]
    rot = [0]

    def run(lib, h, env, x, n):
        for k in all_keys:      # a library sees only its own overrides (some are read once, at the first call)
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = v
        for _ in range(n):
            i = rot[0] % args.rotate
            rot[0] += 1
            ii = i if args.rotate_what in ("both", "in") else 0
            io = i if args.rotate_what in ("both", "out") else 0
            rc = lib.lanczos_resample_device(h, ctypes.byref(d), x[ii].data_ptr(), ys[io].data_ptr(), frames, 0, 0, stream)
            assert rc == 0, rc

    for pat in args.patterns.split(","):
        x = [bench.make_frames(torch, pat, frames, ih, iw, c, bps, dev, 1234 + i) for i in range(args.rotate)]
        res = {spec: [] for spec, *_ in handles}
        ref_out = None
        for spec, lib, h, env in handles:   # warm-up (+ optional output comparison)
            run(lib, h, env, x, 30)
            torch.cuda.synchronize()
            if args.check:
                if ref_out is None:
                    ref_out = y.clone()
                else:
                    diff = (y.to(torch.int32) - ref_out.to(torch.int32)).abs()
                    print(f"check {pat} {spec}: max |diff| vs first = {int(diff.max())}, differing = {int((diff != 0).sum())}")
        for r in range(args.rounds):
            order = handles if r % 2 == 0 else handles[::-1]
            for spec, lib, h, env in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                run(lib, h, env, x, 3)
                e0.record()
                run(lib, h, env, x, args.steps)
                e1.record()
                torch.cuda.synchronize()
                res[spec].append(e0.elapsed_time(e1) / args.steps * 1e3)
        for spec, *_ in handles:
            v = res[spec]
            med = statistics.median(v)
            print(f"{args.config} {pat:9s} {spec:60s} median {med:7.2f} us  min {min(v):7.2f}  frac {alg / med / 1e3 / 8000:.4f}  "
                  f"all {[round(t, 1) for t in v]}", flush=True)
    for spec, lib, h, env in handles:
        lib.lanczos_destroy(h)


if __name__ == "__main__":
    main()
