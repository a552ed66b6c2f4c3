#!/usr/bin/env python3
"""scripts/rat_speed.py lib.so [lib2.so ...] -- rational-scale kernels: time 4 x 1080p frames at 4/3 and 3/2 per library
(LANCZOS_LIB is read at import: one subprocess per library), fast family vs the f64 generic kernel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import lanczos_hls_amd as L, patterns as P
ctx = L.Context(0)
x = torch.from_numpy(np.stack([P.gradient_noise(1080, 1920, 3, seed=40 + i) for i in range(4)])).cuda()
for (sn, sd) in ((4, 3), (3, 2), (5, 2)):
    d = L.make_desc(1920, 1080, 3, sn, sd, 3)
    y = torch.empty((4, d.out_h, d.out_w, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for fam in (L.KERNEL_GENERIC, L.KERNEL_NONE):
        ctx.force_kernel(fam)
        for _ in range(5): ctx.resample_device(d, x.data_ptr(), y.data_ptr(), 4, 0, 0, st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ctx.resample_device(d, x.data_ptr(), y.data_ptr(), 4, 0, 0, st)
        torch.cuda.synchronize(); res[fam] = (time.perf_counter() - t0) / 20 * 1e6
    print(f"  {sn}/{sd}: generic {res[1]:.1f} us  fast {res[0]:.1f} us  ratio {res[1] / res[0]:.2f}", flush=True)
''' % (ROOT, ROOT)
for lib in sys.argv[1:]:
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LANCZOS_LIB=os.path.abspath(lib)), check=False)
