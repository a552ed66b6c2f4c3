#!/usr/bin/env python3
"""scripts/instance_speed.py -- whole-step device time of the integer-scale kernel instances at 4K-class output sizes
(4x and RGBA8 3x next to the BASELINE shapes), fast kernel vs the f64 generic kernel, inputs cycled."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import lanczos_hls_amd as L

CASES = [  # (in_w, in_h, C, bps, S, a, frames)
    (1920, 1080, 3, 1, 2, 3, 16), (1280, 720, 3, 1, 3, 3, 16), (960, 540, 3, 1, 4, 3, 16), (960, 540, 4, 1, 4, 3, 16),
    (1280, 720, 4, 1, 3, 3, 16), (1920, 1080, 4, 1, 2, 3, 16), (1920, 1080, 1, 1, 2, 3, 16), (960, 540, 1, 1, 4, 2, 16),
    (1280, 720, 3, 1, 3, 2, 16), (1280, 720, 3, 1, 3, 4, 16), (1920, 1080, 3, 1, 2, 2, 16), (1920, 1080, 3, 1, 2, 4, 16),
    (1280, 720, 3, 2, 3, 3, 8), (1280, 720, 4, 2, 3, 4, 8), (1920, 1080, 4, 2, 2, 3, 8),
]
dev = torch.device("cuda", 0)
ctx = L.Context(0)
stream = torch.cuda.current_stream().cuda_stream
for (iw, ih, c, bps, s, a, frames) in CASES:
    d = L.make_desc(iw, ih, c, s, 1, a, bps)
    rot = 3
    xs = [bench.make_frames(torch, "gradient", frames, ih, iw, c, bps, dev, 10 + i) for i in range(rot)]
    ys = [torch.empty((frames, d.out_h, d.out_w, c), device=dev, dtype=xs[0].dtype) for _ in range(rot)]
    alg = frames * (iw * ih + d.out_w * d.out_h) * c * bps
    res = {}
    for fam, name in ((L.KERNEL_NONE, "fast"), (L.KERNEL_GENERIC, "generic")):
        ctx.force_kernel(fam)
        n = 20 if name == "fast" else 3
        for i in range(3):
            ctx.resample_device(d, xs[i % rot].data_ptr(), ys[i % rot].data_ptr(), frames, 0, 0, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            ctx.resample_device(d, xs[i % rot].data_ptr(), ys[i % rot].data_ptr(), frames, 0, 0, stream)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n * 1e3
        fam_used = ctx.last_kernel()
    ctx.force_kernel(L.KERNEL_NONE)
    print(f"{'u8 ' if bps == 1 else 'u16'} C{c} {s}x a={a} {iw}x{ih}->{d.out_w}x{d.out_h} x{frames}: fast {res['fast']:8.1f} us "
          f"({alg / res['fast'] / 1e3 / 8000:.3f} of 8 TB/s, {frames * d.out_w * d.out_h / res['fast']:.0f} Mpix/s)   generic {res['generic']:9.1f} us   "
          f"x{res['generic'] / res['fast']:.1f}", flush=True)
ctx.close()
