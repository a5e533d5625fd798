"""The k-draw "reservoir by tickets" closed form has EXACTLY the output distribution of the reference's
reservoir loop (src/utils/sampling.rs:6-26, including its 0..i quirk).

Both laws are enumerated exhaustively with rational arithmetic (no RNG, no C code): the reference loop
over all index vectors (j_k, ..., j_{n-1}), the closed form over all of its bounded draws.  Then the C
oracle's two philox algorithms are checked against that law empirically."""
from collections import Counter
from fractions import Fraction
from itertools import product

import numpy as np
import pytest

import orc


def law_reference_loop(n, k):
    """dst after `for i in k..n: j = U[0,i); if j < k: dst[j] = i` for every equally likely j-vector."""
    law = Counter()
    ranges = [range(i) for i in range(k, n)]
    total = 1
    for r in ranges:
        total *= len(r)
    for js in product(*ranges):
        dst = list(range(k))
        for i, j in zip(range(k, n), js):
            if j < k:
                dst[j] = i
        law[tuple(dst)] += Fraction(1, total)
    return law


def law_tickets(n, k):
    """Enumerates the closed form: k ordered tickets without replacement from an urn of (n-k) position
    tickets and (k-1) blanks, drawn by partial Fisher-Yates with r_s ~ U[0, n-1-s)."""
    law = Counter()
    ranges = [range(n - 1 - s) for s in range(k)]
    total = 1
    for r in ranges:
        total *= len(r)
    for rs in product(*ranges):
        urn = list(range(n - 1))
        dst = []
        for s, r in enumerate(rs):
            last = n - 2 - s
            t = urn[r]
            urn[r] = urn[last]
            dst.append(k + t if t < n - k else s)
        law[tuple(dst)] += Fraction(1, total)
    return law


@pytest.mark.parametrize("n,k", [(2, 1), (3, 1), (5, 1), (3, 2), (4, 2), (6, 2), (7, 3), (6, 4), (8, 3), (7, 5)])
def test_tickets_law_equals_reference_loop_law_exactly(n, k):
    a, b = law_reference_loop(n, k), law_tickets(n, k)
    assert sum(a.values()) == 1 and sum(b.values()) == 1
    assert a == b


def test_quirk_item_k_never_leaves_unless_evicted():
    # sampling.rs:19 draws from 0..i: item k always enters; e.g. n = k+1 always contains item k
    law = law_reference_loop(4, 3)
    assert all(3 in dst for dst in law)


@pytest.mark.parametrize("algo", [orc.RES_TICKETS, orc.RES_LITERAL])
@pytest.mark.parametrize("n,k", [(6, 2), (7, 3), (9, 4)])
def test_c_oracle_philox_algorithms_follow_the_law(algo, n, k):
    law = law_reference_loop(n, k)
    trials = 40000
    obs = Counter()
    for c in range(trials):
        obs[tuple(orc.reservoir_positions(orc.rng_philox(0xABCDEF, c), n, k, algo=algo).tolist())] += 1
    assert set(obs) <= set(law)
    chi2 = sum((obs[o] - trials * float(p)) ** 2 / (trials * float(p)) for o, p in law.items())
    dof = len(law) - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof), (chi2, dof)      # ~6 sigma


def test_c_oracle_ref_mode_follows_the_law():
    n, k = 7, 3
    law = law_reference_loop(n, k)
    rng = orc.rng_ref()
    trials = 40000
    obs = Counter(tuple(orc.reservoir_positions(rng, n, k).tolist()) for _ in range(trials))
    chi2 = sum((obs[o] - trials * float(p)) ** 2 / (trials * float(p)) for o, p in law.items())
    dof = len(law) - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof)


def test_small_n_takes_everything_in_order():
    for algo in (orc.RES_TICKETS, orc.RES_LITERAL):
        assert orc.reservoir_positions(orc.rng_philox(1), 3, 5, algo=algo).tolist() == [0, 1, 2]
        assert orc.reservoir_positions(orc.rng_philox(1), 5, 5, algo=algo).tolist() == [0, 1, 2, 3, 4]
        assert orc.reservoir_positions(orc.rng_philox(1), 0, 5, algo=algo).tolist() == []
