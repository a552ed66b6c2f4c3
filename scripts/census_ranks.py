#!/usr/bin/env python3
"""scripts/census_ranks.py <census dump> [nb] -- lifetimes of k_march's workgroups by slot class (rank on the CU x CU full / one
short), from a LANCZOS_STAMP=1 LANCZOS_CENSUS_DUMP=<file> run; prints the relative shares that would equalise them."""
import collections
import sys

import numpy as np

d = np.loadtxt(sys.argv[1])
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
idx, st, en = d[:, 0].astype(int), d[:, 1] / 100.0, d[:, 2] / 100.0
n = len(idx)
life = en - st
j = idx // 8
x = idx % 8
n_x = np.array([(n - xx + 7) // 8 for xx in range(8)])[x]
rank, c = j // 32, j % 32
on_cu = (n_x - c + 31) // 32
print(f"{n} workgroups, span {en.max():.1f} us, lifetime min/p50/max {life.min():.1f}/{np.median(life):.1f}/{life.max():.1f}, end min/p10/p50/max "
      f"{en.min():.1f}/{np.percentile(en, 10):.1f}/{np.median(en):.1f}/{en.max():.1f}")
mean = life.mean()
for full in (True, False):
    for r in range(nb):
        m = (rank == r) & ((on_cu >= nb) if full else (on_cu == nb - 1))
        if m.sum():
            print(f"  {'full ' if full else 'short'} rank {r}: n={m.sum():4d} lifetime {life[m].mean():7.1f} us  -> share x {mean / life[m].mean():.3f}")
