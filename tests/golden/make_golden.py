#!/usr/bin/env python3
"""Generate tests/golden/ from the REFERENCE's own compiled software path (oracle/_ref).

Run in the build container (where /root/reference exists) after `make -C oracle ref`:
    python tests/golden/make_golden.py
Writes
  * golden_small.npz   -- for every small shape of oracle/ref_configs.txt and two input
                          patterns: planar input bytes + the reference's planar output bytes
  * kat_digests.json   -- FNV-1a-64 digests of the reference output for the large shapes
                          (LCG input seed 12345, planar CHW order; cross-checked against the
                          five digests recorded in SURVEY.md 8(c))
Fixtures are data only (inputs and expected outputs); no reference source is stored.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

SURVEY_KAT = {
    "256x256_512x512_2-1_a2_c3": "d620a2ac5725fb2b",
    "1920x1080_3840x2160_2-1_a3_c3": "9ccefeb71ba19851",
    "1280x720_3840x2160_3-1_a3_c3": "a9b89004c5a2864a",
    "480x270_960x540_2-1_a4_c4": "ff9cd4a91feddc66",
    "300x200_400x266_4-3_a3_c3": "1d0d4ed16b6c2d4e",
}
SMALL_MAX_OUT = 64 * 64 * 4  # shapes with at most this many output samples are stored raw


def key(iw, ih, ow, oh, sn, sd, a, c):
    return f"{iw}x{ih}_{ow}x{oh}_{sn}-{sd}_a{a}_c{c}"


def patterns(c, ih, iw):
    n = c * ih * iw
    noise = O.lcg_u8(n, 12345).reshape(c, ih, iw)
    # dark noise: values 0..71 -- the regime where the double sum at integer phases lands one ulp
    # below the centre sample and truncates to v0-1
    dark = (O.lcg_u8(n, 777).astype(np.uint16) * 72 // 256).astype(np.uint8).reshape(c, ih, iw)
    return {"noise": noise, "dark": dark}


def main():
    small = {}
    digests = {}
    for (iw, ih, ow, oh, sn, sd, a, c) in O.ref_configs():
        cfg = O.cfg(iw, ih, ow, oh, c, a, sn, sd)
        k = key(iw, ih, ow, oh, sn, sd, a, c)
        for pname, img in patterns(c, ih, iw).items():
            ref = O.ref_expected_planar_u8(cfg, img)
            if ref is None:
                raise SystemExit(f"reference build missing for {k}: run `make -C oracle ref`")
            digests[f"{k}:{pname}"] = f"{O.fnv1a64(ref):016x}"
            if ow * oh * c <= SMALL_MAX_OUT:
                small[f"{k}:{pname}:in"] = img
                small[f"{k}:{pname}:out"] = ref
        if k in SURVEY_KAT:
            assert digests[f"{k}:noise"] == SURVEY_KAT[k], (k, digests[f"{k}:noise"])
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **small)
    with open(os.path.join(HERE, "kat_digests.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/_ref = reference full_TB.h:29-96)",
                   "input": "planar [C][H][W]; noise = LCG seed 12345 (s*1664525+1013904223, s>>24); "
                            "dark = (LCG seed 777) * 72 // 256",
                   "digest": "FNV-1a-64 over planar [C][OUT_H][OUT_W]",
                   "survey_8c": SURVEY_KAT,
                   "digests": digests}, f, indent=1, sort_keys=True)
    print(f"wrote {len(small) // 2} raw fixtures, {len(digests)} digests")


if __name__ == "__main__":
    main()
