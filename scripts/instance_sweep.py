#!/usr/bin/env python3
"""scripts/instance_sweep.py [lsb1|exact] -- whole-step device time of EVERY marching-kernel instance (lanczos_fast.hpp LZ_FAST_CONFIGS)
at a 4K-class output, gradient input, 3 batches cycled.  One line per instance; run it once per library (LANCZOS_LIB) and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import lanczos_hls_amd as L
mode = L.MODE_EXACT if len(sys.argv) > 1 and sys.argv[1] == "exact" else L.MODE_LSB1
cases = [(1, c, s, a) for s in (2, 3, 4) for c in (3, 4, 1) for a in (2, 3, 4)] + [(2, c, s, a) for c in (3, 4) for s in (2, 3) for a in (3, 4)]
dev = torch.device("cuda", 0)
ctx = L.Context(0)
stream = torch.cuda.current_stream().cuda_stream
for (bps, c, s, a) in cases:
    iw, ih = {2: (1920, 1080), 3: (1280, 720), 4: (960, 540)}[s]
    frames = 16 if bps == 1 else 8
    d = L.make_desc(iw, ih, c, s, 1, a, bps, mode)
    xs = [bench.make_frames(torch, "gradient", frames, ih, iw, c, bps, dev, 10 + i) for i in range(3)]
    ys = [torch.empty((frames, d.out_h, d.out_w, c), device=dev, dtype=xs[0].dtype) for _ in range(3)]
    alg = frames * (iw * ih + d.out_w * d.out_h) * c * bps
    best = 1e30
    for rnd in range(3):
        for i in range(4):
            ctx.resample_device(d, xs[i % 3].data_ptr(), ys[i % 3].data_ptr(), frames, 0, 0, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 12
        for i in range(n):
            ctx.resample_device(d, xs[i % 3].data_ptr(), ys[i % 3].data_ptr(), frames, 0, 0, stream)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    assert ctx.last_kernel() == L.KERNEL_FAST
    print(f"{'u8 ' if bps == 1 else 'u16'} C{c} {s}x a={a} x{frames}: {best:8.1f} us  {alg / best / 1e3 / 8000:.3f}", flush=True)
    del xs, ys
ctx.close()
