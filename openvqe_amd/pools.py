"""Fermionic operator pools emitting packed Pauli sums directly.

``spin_complement_gsd`` follows the enumeration ORDER of ref:openvqe/common_files/generator_excitations.py:83-156
(singles over even p <= q, then for every (p, q) the doubles over (r, s) in three spin couplings a/b/c),
because pool INDICES are what the ADAPT traces record.  The raw enumeration has 69 entries for 3 orbitals and 175
for 4 — the pool sizes pinned by ref:tests/test_main_fermionic_adapt.py:11,15.  "Cc"[i, j] = a+_i a_j, "CcCc"[i, j, k, l] =
a+_i a_j a+_k a_l, spin orbitals interleaved (alpha even / beta odd), un-normalised +-1 coefficients.
"""
from __future__ import annotations

from . import fermion


def _jw_sum(nqbits, terms):
    """terms: [(coeff, [(orbital, dagger), ...])] -> Pauli-sum dict"""
    total = {}
    for coeff, ladder in terms:
        total = fermion.psum_add(total, fermion.jw_product(ladder), coeff)
    return total


def _cc(i, j):
    return [(i, True), (j, False)]


def _cccc(i, j, k, l):
    return [(i, True), (j, False), (k, True), (l, False)]


def spin_complement_gsd(n_elec, orbital_number, transform="JW"):
    """-> (pool_size, cluster_ops_sp): anti-Hermitian spin-complemented generalised singles and doubles"""
    if transform != "JW":
        raise NotImplementedError("only the Jordan-Wigner mapping is restated")
    n = 2 * orbital_number
    singles, doubles = [], []
    for p in range(0, n, 2):
        for q in range(p, n, 2):
            singles.append([(1, _cc(p, q)), (-1, _cc(q, p)), (1, _cc(p + 1, q + 1)), (-1, _cc(q + 1, p + 1))])
            for r in range(p, n, 2):
                for s in range(q if r == p else r, n, 2):
                    term_a = [(1, _cccc(r, p, s, q)), (-1, _cccc(q, s, p, r)),
                              (1, _cccc(r + 1, p + 1, s + 1, q + 1)), (-1, _cccc(q + 1, s + 1, p + 1, r + 1))]
                    term_b = [(1, _cccc(r, p, s + 1, q + 1)), (-1, _cccc(q + 1, s + 1, p, r)),
                              (1, _cccc(r + 1, p + 1, s, q)), (-1, _cccc(q, s, p + 1, r + 1))]
                    term_c = [(1, _cccc(r, p + 1, s + 1, q)), (-1, _cccc(q, s + 1, p + 1, r)),
                              (1, _cccc(r + 1, p, s, q + 1)), (-1, _cccc(q + 1, s, p, r + 1))]
                    doubles.extend([term_a, term_b, term_c])
    # NB no operator is dropped: with myQLM every entry of the raw enumeration survives `_apply_transforms`
    # (the pinned sizes 69 / 175 ARE the raw loop counts), identically-zero ones included —
    # they simply carry no Pauli terms here and rank with gradient 0.
    pool = [fermion.psum_to_hamiltonian(n, _jw_sum(n, terms), tol=1e-13) for terms in singles + doubles]
    return len(pool), pool


def singlet_upccgsd(n_orb, transform="JW", perm=0):
    """-> (pool_size, cluster_ops_sp): spin-adapted generalised singles + paired doubles, the list repeated
    ``perm`` extra times (k-UpCCGSD) — enumeration of ref:openvqe/common_files/generator_excitations.py:403-466
    (pool size 36 for H2/6-31G with perm = 2, ref:tests/test_main_ucc.py:15)."""
    if transform != "JW":
        raise NotImplementedError("only the Jordan-Wigner mapping is restated")
    n = 2 * n_orb
    singles, doubles = [], []
    for p in range(0, n, 2):
        for q in range(0, p, 2):
            singles.append([(1, _cc(q, p)), (-1, _cc(p, q)), (1, _cc(q + 1, p + 1)), (-1, _cc(p + 1, q + 1))])
    import itertools
    for p, q in itertools.combinations(range(0, n, 2), 2):
        doubles.append([(1.0, _cccc(q, p, q + 1, p + 1)), (-1.0, _cccc(p + 1, q + 1, p, q))])
    pool = [fermion.psum_to_hamiltonian(n, _jw_sum(n, terms), tol=1e-13) for terms in singles + doubles]
    pool = pool + pool * perm
    return len(pool), pool


# ---------------------------------------------------------------------------------------------- qubit pools
def qubit_pool(kind, nbqbits, rng=None):
    """Single-Pauli-string pools of qubit-ADAPT (ref:openvqe/common_files/qubit_pool.py:278-465, 1184-1268):
    'YXXX' | 'XYXX' | 'XXYX' | 'XXXY' — "YX" on every pair (a, b) with a + b even, then the 4-letter string on every
    quadruple with an even number of odd indices, each as Hamiltonian(n, [Term(-1.0, string, qubits)]);
    'random' — position by position one of the four pools (the reference draws with an unseeded
    np.random.randint; pass ``rng`` for a reproducible draw).  50 operators at 8 qubits
    (ref:tests/test_main_qubit_adapt.py:14)."""
    import itertools

    from .operators import Hamiltonian, Term

    def family(word):
        out = []
        for a, b in itertools.combinations(range(nbqbits), 2):
            if (a + b) % 2 == 0:
                out.append(Hamiltonian(nbqbits, [Term(-1.0, "YX", [a, b])], do_clean_up=False))
        for q in itertools.combinations(range(nbqbits), 4):
            if sum(k % 2 for k in q) % 2 == 0:
                out.append(Hamiltonian(nbqbits, [Term(-1.0, word, list(q))], do_clean_up=False))
        return out

    if kind in ("YXXX", "XYXX", "XXYX", "XXXY"):
        pool = family(kind)
    elif kind == "random":
        import numpy as np
        rng = rng or np.random.default_rng()
        fams = [family(w) for w in ("YXXX", "XYXX", "XXYX", "XXXY")]
        pool = [fams[int(rng.integers(0, 4))][i] for i in range(len(fams[3]))]
    else:
        raise KeyError(kind)
    return len(pool), pool
