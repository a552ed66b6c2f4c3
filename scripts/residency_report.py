import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import lanczos_hls_amd as L
ctx = L.Context(0)
for (w, h, c, s, a, dt) in [(1920, 64, 3, 2, 3, np.uint8), (1280, 64, 3, 3, 3, np.uint8), (3840, 64, 4, 2, 4, np.uint16), (1920, 64, 1, 3, 3, np.uint8), (1280, 64, 3, 3, 4, np.uint8)]:
    img = np.zeros((h, w, c), dt)
    for mode in (L.MODE_LSB1, L.MODE_EXACT):
        ctx.resample(img, s, 1, a, mode)
