#!/usr/bin/env python3
"""scripts/instance_one.py in_w in_h C bps S a frames [mode] -- whole-step device time of one kernel instance (gradient input, 3 batches cycled)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import lanczos_hls_amd as L
iw, ih, c, bps, s, a, frames = [int(x) for x in sys.argv[1:8]]
mode = L.MODE_EXACT if len(sys.argv) > 8 and sys.argv[8] == "exact" else L.MODE_LSB1
dev = torch.device("cuda", 0)
ctx = L.Context(0)
stream = torch.cuda.current_stream().cuda_stream
d = L.make_desc(iw, ih, c, s, 1, a, bps, mode)
xs = [bench.make_frames(torch, "gradient", frames, ih, iw, c, bps, dev, 10 + i) for i in range(3)]
ys = [torch.empty((frames, d.out_h, d.out_w, c), device=dev, dtype=xs[0].dtype) for _ in range(3)]
alg = frames * (iw * ih + d.out_w * d.out_h) * c * bps
for rnd in range(4):
    for i in range(5):
        ctx.resample_device(d, xs[i % 3].data_ptr(), ys[i % 3].data_ptr(), frames, 0, 0, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        ctx.resample_device(d, xs[i % 3].data_ptr(), ys[i % 3].data_ptr(), frames, 0, 0, stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
print(f"{os.environ.get('LANCZOS_LIB', 'product')}: {'u8' if bps == 1 else 'u16'} C{c} {s}x a={a} x{frames} {'exact' if mode == L.MODE_EXACT else 'lsb1'}: {us:.1f} us ({alg / us / 1e3 / 8000:.3f} of 8 TB/s)")
