"""Shared helpers for the tests: seeded random Pauli operators / programs."""
import numpy as np

from openvqe_amd.operators import Hamiltonian, Term


def random_string(rng, n, min_weight=1, max_weight=None, alphabet="XYZ"):
    max_weight = max_weight or n
    w = int(rng.integers(min_weight, max_weight + 1))
    qs = sorted(rng.choice(n, w, replace=False).tolist())
    op = "".join(rng.choice(list(alphabet), w))
    return op, qs


def random_hamiltonian(rng, n, nterms, constant=None):
    seen = set()
    terms = []
    while len(terms) < nterms:
        op, qs = random_string(rng, n)
        key = (op, tuple(qs))
        if key in seen:
            continue
        seen.add(key)
        terms.append(Term(float(rng.normal()), op, qs))
    return Hamiltonian(n, terms, float(rng.normal()) if constant is None else constant)


def random_generators(rng, n, k, max_terms=4, same_support_prob=0.5):
    """k Hermitian generators with real coefficients; some share an x-mask inside (fusable runs)."""
    gens = []
    for _ in range(k):
        nt = int(rng.integers(1, max_terms + 1))
        terms = []
        if rng.random() < same_support_prob:
            # strings on one support differing in X<->Y only: identical x masks
            w = int(rng.integers(1, min(n, 4) + 1))
            qs = sorted(rng.choice(n, w, replace=False).tolist())
            for _ in range(nt):
                op = "".join(rng.choice(list("XY"), w))
                terms.append(Term(float(rng.normal()), op, qs))
        else:
            for _ in range(nt):
                op, qs = random_string(rng, n)
                terms.append(Term(float(rng.normal()), op, qs))
        gens.append(Hamiltonian(n, terms, do_clean_up=False))
    return gens


def random_state(rng, n):
    psi = rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)
    return psi / np.linalg.norm(psi)
