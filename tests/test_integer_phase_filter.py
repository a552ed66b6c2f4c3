"""The marching kernel's second-stage integer-phase filter (lanczos_march.hpp, lanczos_taps.cpp: integer_phase_tight2),
checked EXHAUSTIVELY on the CPU against the reference's double chain (full_TB.h:58-63 at x == floor(x): separate multiply
and add, ascending taps), with the weights of the checker's own lanczos_kernel (oracle/, libm sin):

    a sample v0 (1 <= v0 <= 64) whose double sum truncates to v0 - 1 always satisfies
        n(-2) > 2 n(-1) + 3 v0    or    ( n(+2) > 3 v0  and  n(+2) + 4 v0 + 2 > 2 n(+1) )

The sum is  ((((t + n(-2) w2) + n(-1) w1) + v0) + n(+1) w1) + n(+2) w2 (+ taps at |d| >= 3).  Every step is monotone in the
running sum, so the state after `+ v0` is all that the samples in front pass on: for every v0 the 65 536 (n(-2), n(-1)) pairs
that the first clause does NOT flag are folded into their distinct states (a handful), and each state is pushed through
all 65 536 (n(+1), n(+2)) pairs that the second clause does not flag -- none may end below v0.  The taps at |d| >= 3 are
non-negative or negligible (asserted here as on the host): they can only raise the sum, so they are taken as absent."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def test_filter_is_specific_to_lanczos_3():
    """Lanczos-4's +-1 taps are 1.41 x its +-2 taps, not twice: the host must not enable the filter there
    (integer_phase_tight2 checks |w(1) / |w(2)| - 2| < 1e-6)."""
    OL = O.lib()
    r3 = OL.oracle_lanczos_kernel(1.0, 3) / -OL.oracle_lanczos_kernel(2.0, 3)
    r4 = OL.oracle_lanczos_kernel(1.0, 4) / -OL.oracle_lanczos_kernel(2.0, 4)
    assert abs(r3 - 2.0) < 1e-12 and abs(r4 - 2.0) > 0.5


@pytest.mark.parametrize("a", [3])
def test_second_stage_filter_never_drops_a_flip(a):
    OL = O.lib()
    w = {d: OL.oracle_lanczos_kernel(float(d), a) for d in range(-a, a)}   # tap of x - i = d, x integer
    assert w[0] == 1.0 and w[2] < 0 and w[-2] == w[2] and w[1] > 0 and w[-1] == w[1]
    for d in w:
        if abs(d) >= 3:
            assert w[d] >= 0 or abs(w[d]) * 255 < 2.0 ** -70
    n = np.arange(256, dtype=np.float64)
    ni = np.arange(256, dtype=np.int64)
    flagged = 0.0
    for v0 in range(1, 65):
        # ---- in front of v0 (ascending taps: n(-2) w2 first, then + n(-1) w1, then + v0)
        s2 = ((n[:, None] * w[2]) + (n[None, :] * w[1])) + float(v0)          # [n(-2), n(-1)]
        a_flag = ni[:, None] > 2 * ni[None, :] + 3 * v0                         # first clause
        states = np.unique(s2[~a_flag])                                         # what unflagged fronts can hand on
        assert states.min() >= v0, f"v0={v0}: an unflagged front already left v0"
        # ---- behind v0
        b_flag = (ni[None, :] > 3 * v0) & (ni[None, :] + 4 * v0 + 2 > 2 * ni[:, None])   # [n(+1), n(+2)]: second clause
        p1 = n[:, None] * w[-1]
        p2 = n[None, :] * w[-2]
        for s in states:
            bad = (((s + p1) + p2) < v0) & ~b_flag
            assert not bad.any(), f"v0={v0} state {s!r}: flip missed at (n+1, n+2) = {np.argwhere(bad)[:3].tolist()}"
        flagged += 1 - (1 - a_flag.mean()) * (1 - b_flag.mean())
    # on uniform noise about a third of the samples with 1 <= v0 <= 64 (a quarter of all samples) stay flagged: ~8 % of all
    assert 0.2 < flagged / 64 < 0.45
