"""Synthetic frame generators of SURVEY.md 8(d): interleaved HWC, fixed seeds."""
import numpy as np

import oracle_lib as O


def noise(h, w, c, seed=12345, dtype=np.uint8):
    """LCG noise (worst case for the integer-phase double-rounding quirk)."""
    n = h * w * c
    if dtype == np.uint8:
        return O.lcg_u8(n, seed).reshape(h, w, c)
    return O.lcg_u16(n, seed).reshape(h, w, c)


def dark_noise(h, w, c, seed=777):
    return (O.lcg_u8(h * w * c, seed).astype(np.uint16) * 72 // 256).astype(np.uint8).reshape(h, w, c)


def gradient_noise(h, w, c, seed=99):
    """(x*255/W + y*255/H)/2 + 2-bit noise: natural-image-like."""
    y, x = np.mgrid[0:h, 0:w]
    base = ((x * 255 // max(w, 1) + y * 255 // max(h, 1)) // 2).astype(np.int32)
    nz = (O.lcg_u8(h * w * c, seed).reshape(h, w, c) >> 6).astype(np.int32)
    return np.clip(base[..., None] + nz, 0, 255).astype(np.uint8)


def blocks(h, w, c):
    """((x/16 + y/16 + c) % 5) * 60: flat regions, worst for truncation ties."""
    y, x = np.mgrid[0:h, 0:w]
    ch = np.arange(c)
    return ((((x // 16 + y // 16)[..., None] + ch) % 5) * 60).astype(np.uint8)


ALL_U8 = {"noise": noise, "dark": dark_noise, "gradient": gradient_noise, "blocks": blocks}
