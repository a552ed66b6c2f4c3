#!/usr/bin/env python3
"""tests/instances_full_size_gpu.py -- every marching-kernel instance (lanczos_fast.hpp LZ_FAST_CONFIGS: 35 sample-type / channel /
scale / a combinations) at a 4K-class output size, both parity modes, against the CPU oracle (test infrastructure, not collected
by pytest: minutes of CPU time).  The pytest suite covers the same instances at 160 x 45; this run exercises them with full
workgroup tables (every table mode, hundreds of ticks per chunk, the register budgets of round 4).  Two frames per call, so
that a launch holds more than one (strip, frame) pair per XCD.  Exit code 1 on any mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lanczos_hls_amd as L  # noqa: E402
import oracle_lib as O  # noqa: E402
import patterns as P  # noqa: E402


def main():
    ctx = L.Context(0)
    cases = [(np.uint8, c, s, a) for s in (2, 3, 4) for c in (3, 4, 1) for a in (2, 3, 4)]
    cases += [(np.uint16, c, s, a) for c in (3, 4) for s in (2, 3) for a in (3, 4)]
    bad = n = 0
    t0 = time.time()
    for (dt, c, s, a) in cases:
        w, h = {2: (1920, 1080), 3: (1280, 720), 4: (960, 540)}[s]
        frames = []
        for seed, kind in ((11, "noise"), (12, "ramp")):
            if dt == np.uint8:
                img = P.noise(h, w, c, seed=seed) if kind == "noise" else P.gradient_noise(h, w, c, seed=seed)
            else:
                img = P.noise(h, w, c, seed=seed, dtype=np.uint16)
                if kind == "ramp":
                    yy, xx = np.mgrid[0:h, 0:w]
                    img = np.clip((yy * 50 + xx * 20)[..., None] + (img >> 9), 0, 65535).astype(np.uint16)
            frames.append(np.ascontiguousarray(img))
        cfg = O.cfg(w, h, w * s, h * s, c, a, s, 1)
        want = [O.expected_hwc_u8(cfg, f, threads=16) if dt == np.uint8 else O.expected_hwc_u16(cfg, f, threads=16) for f in frames]
        batch = np.stack(frames)
        for mode in (L.MODE_EXACT, L.MODE_LSB1):
            got = ctx.resample(batch, s, 1, a, mode)
            assert ctx.last_kernel() == L.KERNEL_FAST
            for i in range(2):
                d = np.abs(got[i].astype(np.int32) - want[i].astype(np.int32))
                ok = d.max() == 0 if mode == L.MODE_EXACT else d.max() <= 1
                n += 1
                if not ok:
                    bad += 1
                    print(f"MISMATCH {dt.__name__} C{c} {s}x a={a} mode {mode} frame {i}: max |diff| {d.max()}, {np.count_nonzero(d if mode == L.MODE_EXACT else d > 1)} samples", flush=True)
        print(f"{dt.__name__} C{c} {s}x a={a} {w}x{h}->{w * s}x{h * s}: ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"instances at full size: {len(cases)} instances x 2 modes x 2 frames = {n} comparisons, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
