#!/usr/bin/env python3
"""Generate tests/golden/kat_digests_u16.json: known-answer digests for 16-bit samples.

PARITY UNPINNED BY THE REFERENCE: its software path is 8-bit only (`typedef ap_uint<8> byte`, clamp UINT8_MAX,
full_TB.h:18,30), so there is no reference build to run.  The generator is the CPU restatement of full_TB.h:29-96
(oracle/lanczos_oracle.c) templated on the sample type -- the same code that is bit-identical to the reference's own
compiled lines at 8 bit (tests/test_oracle.py) -- with clamp 65535.  Run anywhere (no /root/reference needed):
    python tests/golden/make_golden_u16.py
Input: interleaved [H][W][C] uint16, LCG seed 12345 (s = s*1664525 + 1013904223; v = s >> 16).
Digest: FNV-1a-64 over the interleaved [OUT_H][OUT_W][C] little-endian uint16 output bytes.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SHAPES = [
    # (in_w, in_h, channels, scale_n, scale_d, a)
    (3840, 2160, 4, 2, 1, 4),   # BASELINE config 5
    (480, 270, 4, 2, 1, 4),
    (320, 180, 3, 2, 1, 3),
    (150, 100, 3, 4, 3, 3),
]


def main():
    out = {}
    for (iw, ih, c, sn, sd, a) in SHAPES:
        ow, oh = iw * sn // sd, ih * sn // sd
        img = O.lcg_u16(ih * iw * c, 12345).reshape(ih, iw, c)
        res = O.expected_hwc_u16(O.cfg(iw, ih, ow, oh, c, a, sn, sd), img, os.cpu_count() or 1)
        out[f"{iw}x{ih}_{ow}x{oh}_{sn}-{sd}_a{a}_c{c}"] = f"{O.fnv1a64(res):016x}"
        print(iw, ih, out[f"{iw}x{ih}_{ow}x{oh}_{sn}-{sd}_a{a}_c{c}"], flush=True)
    with open(os.path.join(HERE, "kat_digests_u16.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden_u16.py (oracle/ restatement templated on uint16)",
                   "pin": "PARITY UNPINNED BY THE REFERENCE (no 16-bit path exists there)",
                   "input": "interleaved [H][W][C] uint16, LCG seed 12345, v = s >> 16",
                   "digest": "FNV-1a-64 over interleaved [OUT_H][OUT_W][C] little-endian uint16",
                   "digests": out}, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
