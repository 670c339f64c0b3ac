"""Fermionic operator pools emitting packed Pauli sums directly.

``spin_complement_gsd`` follows the enumeration ORDER of ref:openvqe/common_files/generator_excitations.py:83-156
(singles over even p <= q, then for every (p, q) the doubles over (r, s) in three spin couplings a/b/c),
because pool INDICES are what the ADAPT traces record.  The raw enumeration has 69 entries for 3 orbitals and 175
for 4 — the pool sizes pinned by ref:tests/test_main_fermionic_adapt.py:11,15.  "Cc"[i, j] = a+_i a_j, "CcCc"[i, j, k, l] =
a+_i a_j a+_k a_l, spin orbitals interleaved (alpha even / beta odd), un-normalised +-1 coefficients.
"""
from __future__ import annotations

from . import fermion


def _jw_sum(nqbits, terms, transform="JW"):
    """terms: [(coeff, [(orbital, dagger), ...])] -> Pauli-sum dict"""
    total = {}
    for coeff, ladder in terms:
        total = fermion.psum_add(total, fermion.encoded_product(nqbits, ladder, transform), coeff)
    return total


def _cc(i, j):
    return [(i, True), (j, False)]


def _cccc(i, j, k, l):
    return [(i, True), (j, False), (k, True), (l, False)]


def spin_complement_gsd(n_elec, orbital_number, transform="JW"):
    """-> (pool_size, cluster_ops_sp): anti-Hermitian spin-complemented generalised singles and doubles"""
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
    pool = [fermion.psum_to_hamiltonian(n, _jw_sum(n, terms, transform), tol=1e-13) for terms in singles + doubles]
    return len(pool), pool


def singlet_upccgsd(n_orb, transform="JW", perm=0, with_fermionic=False):
    """-> (pool_size, cluster_ops_sp) [or (pool_size, cluster_ops, cluster_ops_sp) ``with_fermionic``]: spin-adapted
    generalised singles + paired doubles, the list repeated ``perm`` extra times (k-UpCCGSD) — enumeration of
    ref:openvqe/common_files/generator_excitations.py:403-466 (pool size 36 for H2/6-31G with perm = 2,
    ref:tests/test_main_ucc.py:15)."""
    import itertools
    n = 2 * n_orb
    ops = []
    for p in range(0, n, 2):
        for q in range(0, p, 2):
            ops.append(_pool_op(n, [(1, "Cc", [q, p]), (1, "Cc", [q + 1, p + 1])], ordered=False))
    for p, q in itertools.combinations(range(0, n, 2), 2):
        ops.append(_pool_op(n, [(1.0, "CcCc", [q, p, q + 1, p + 1])]))
    size, fermi, spin = _finish(ops, transform, perm)
    return (size, fermi, spin) if with_fermionic else (size, spin)


# ---------------------------------------------------------------------------------------------- pools through the
# fermionic layer (normal ordering + merge + JW), the route of ref:…generator_excitations.py:159-609
def _pool_op(n, base, hc_after_all=False, ordered=True, normalise=False):
    """anti-Hermitian operator  sum_k (t_k - t_k^dagger)  from base terms [(coeff, op, orbitals)].  Emission order:
    t_0, -t_0^dagger, t_1, -t_1^dagger, ... (or all t_k first, then all -t_k^dagger); every term is normal-ordered
    before the merge when ``ordered``.  ``normalise``: divide by the 2-norm of the merged coefficients and return None
    for an operator that vanishes (ref:…generator_excitations.py:353-357, 546-550)."""
    from .fermionic import FermionHamiltonian, Term, normal_ordered_terms
    flip = {"C": "c", "c": "C"}
    fwd = [Term(c, op, list(q)) for c, op, q in base]
    bwd = [Term(-c, "".join(flip[o] for o in reversed(op)), list(reversed(q))) for c, op, q in base]
    seq = fwd + bwd if hc_after_all else [t for pair in zip(fwd, bwd) for t in pair]
    if ordered:
        seq = [u for t in seq for u in normal_ordered_terms(t)]
    op = FermionHamiltonian(n, seq)
    if normalise:
        norm = sum(abs(t.coeff) ** 2 for t in op.terms) ** 0.5
        if not norm > 0:
            return None
        op = op / norm
    return op


def _finish(fermi_ops, transform, perm=0):
    """``_apply_transforms`` (ref:…generator_excitations.py:16-36): JW image of every operator, the list repeated
    ``perm`` extra times -> (pool_size, cluster_ops, cluster_ops_sp).  The reference drops operators whose image has
    ``terms == []``; with myQLM that never fires for these pools — the pinned sizes 69 / 175 / 70
    (ref:tests/test_main_fermionic_adapt.py:11,15, ref:tests/test_main_qubit_adapt.py:11) count the identically-zero
    operators of the raw enumeration (p == q singles, ...) — so nothing is dropped here either: such operators carry no
    Pauli strings and rank with gradient 0."""
    from .fermionic import spin_operator
    keep_f = list(fermi_ops)
    keep_s = [spin_operator(op, transform) for op in keep_f]
    keep_f = keep_f + keep_f * perm
    keep_s = keep_s + keep_s * perm
    return len(keep_s), keep_f, keep_s


def _spin_flip(q):
    return [p + 1 for p in q]


def singlet_sd(n_elec, orbital_number, transform="JW"):
    """occupied -> virtual singles and doubles in singlet spin coupling (ref:…generator_excitations.py:274-359):
    per (i <= j occupied, a <= b virtual) the two singlet couplings A (coefficients 2,2,1,1,1,1 / sqrt 12) and B
    (+-1/2), each normalised, vanishing ones skipped."""
    n = 2 * orbital_number
    nocc = -(-n_elec // 2)
    r12 = 12 ** 0.5
    singles, doubles = [], []
    for i in range(0, 2 * nocc, 2):
        for j in range(i, 2 * nocc, 2):
            for a in range(2 * nocc, n, 2):
                if j == i:
                    singles.append(_pool_op(n, [(0.5, "Cc", [a, i]), (0.5, "Cc", [a + 1, i + 1])], hc_after_all=True,
                                            ordered=False))
                for b in range(a, n, 2):
                    mixed = [[a, b + 1, i, j + 1], [a + 1, b, i + 1, j], [a, b + 1, i + 1, j], [a + 1, b, i, j + 1]]
                    coupling_a = [(2 / r12, "CCcc", [a, b, i, j]), (2 / r12, "CCcc", [a + 1, b + 1, i + 1, j + 1])] + \
                                 [(1 / r12, "CCcc", q) for q in mixed]
                    coupling_b = [(s, "CCcc", q) for s, q in zip((0.5, 0.5, -0.5, -0.5), mixed)]
                    for base in (coupling_a, coupling_b):
                        op = _pool_op(n, base, normalise=True)
                        if op is not None:
                            doubles.append(op)
    return _finish(singles + doubles, transform)


def singlet_gsd(n_elec, orbital_number, transform="JW"):
    """generalised singles and doubles in singlet coupling (ref:…generator_excitations.py:468-552); H2/6-31G: 70
    operators (ref:tests/test_main_qubit_adapt.py:11)."""
    n = 2 * orbital_number
    r12 = 12 ** 0.5
    singles, doubles = [], []
    for p in range(0, n, 2):
        for q in range(p, n, 2):
            singles.append(_pool_op(n, [(0.5, "Cc", [p, q]), (0.5, "Cc", [p + 1, q + 1])], ordered=False))
            for r in range(p, n, 2):
                for s in range(q if r == p else r, n, 2):
                    same = [[r, p, s, q], [r + 1, p + 1, s + 1, q + 1]]
                    mixed = [[r, p, s + 1, q + 1], [r + 1, p + 1, s, q], [r, p + 1, s + 1, q], [r + 1, p, s, q + 1]]
                    coupling_a = [(2 / r12, "CcCc", t) for t in same] + [(1 / r12, "CcCc", t) for t in mixed]
                    coupling_b = [(c, "CcCc", t) for c, t in zip((0.5, 0.5, -0.5, -0.5), mixed)]
                    for base in (coupling_a, coupling_b):
                        op = _pool_op(n, base, normalise=True)
                        if op is not None:
                            doubles.append(op)
    return _finish(singles + doubles, transform)


def uccgsd(n_elec, orbital_number, transform="JW"):
    """generalised singles and doubles over SPIN orbitals, no spin adaptation (ref:…generator_excitations.py:555-609)"""
    n = 2 * orbital_number
    singles, doubles = [], []
    for p in range(n):
        for q in range(p, n):
            singles.append(_pool_op(n, [(1, "Cc", [p, q])], ordered=False))
            for r in range(p, n):
                for s in range(q if r == p else r, n):
                    doubles.append(_pool_op(n, [(1, "CCcc", [p, q, r, s])]))
    return _finish(singles + doubles, transform)


def spin_complement_gsd_twin(n_elec, orbital_number, transform="JW"):
    """the "twin" enumeration of the spin-complemented generalised pool (ref:…generator_excitations.py:159-271):
    singles over alpha pairs p < q; same-spin doubles over ordered pair-of-pairs (counter pq >= rs); opposite-spin
    doubles over (alpha, beta) pairs with their spin-swapped partner."""
    n = 2 * orbital_number
    alpha = range(0, n, 2)
    beta = range(1, n, 2)
    ops = []
    for p in alpha:
        for q in alpha:
            if p < q:
                ops.append(_pool_op(n, [(1, "Cc", [q, p]), (1, "Cc", [q + 1, p + 1])], ordered=False))
    pq = 0
    for p in alpha:
        for q in alpha:
            if p > q:
                continue
            rs = 0
            for r in alpha:
                for s in alpha:
                    if r > s:
                        continue
                    if pq >= rs:   # (the reference tests pq < rs BEFORE advancing rs: rs only counts accepted pairs)
                        t = [r, p, s, q]
                        ops.append(_pool_op(n, [(1, "CcCc", t), (1, "CcCc", _spin_flip(t))]))
                        rs += 1
            pq += 1
    pq = 0
    for p in alpha:
        for q in beta:
            rs = 0
            for r in alpha:
                for s in beta:
                    if pq < rs or p > q:
                        continue
                    ops.append(_pool_op(n, [(1, "CcCc", [r, p, s, q]), (1, "CcCc", [s - 1, q - 1, r + 1, p + 1])],
                                        hc_after_all=True))
                    rs += 1
            pq += 1
    return _finish(ops, transform)


# ---------------------------------------------------------------------------------------------- qubit pools derived
# from cluster operators (ref:openvqe/common_files/qubit_pool.py:29-274, 1270-1316)
def generate_pool_from_cluster(pool_condition, cluster_ops, nbqbits):
    """'full': every distinct Pauli string of the JW images of the cluster operators, first-appearance order;
    'full_without_Z': the same strings with their Z factors removed, duplicates dropped;
    'reduced_without_Z': per qubit support of the Z-stripped strings only the first string seen.
    Every pool operator is Hamiltonian(n, [Term(-1.0, string, qubits)]) -> (pool_size, pool).  The reference works
    on a text rendering of the strings ("[X0 Z1 Y2]"); here the same selection runs on (letters, qubits) tuples."""
    from .fermionic import spin_operator
    from .operators import Hamiltonian, Term
    print("The current pool is", pool_condition)
    seen, strings = set(), []
    for op in cluster_ops:
        sp = op if not hasattr(op, "to_spin") else spin_operator(op)
        for t in sp.terms:
            key = (t.op, tuple(t.qbits))
            if key not in seen:
                seen.add(key)
                strings.append(key)

    def strip_z(key):
        kept = [(c, q) for c, q in zip(*key) if c != "Z"]
        return "".join(c for c, _ in kept), tuple(q for _, q in kept)

    if pool_condition == "full":
        chosen = strings
    elif pool_condition == "full_without_Z":
        chosen = list(dict.fromkeys(strip_z(k) for k in strings))
    elif pool_condition == "reduced_without_Z":
        first = {}
        for k in strings:
            letters, qs = strip_z(k)
            first.setdefault(qs, (letters, qs))
        chosen = list(first.values())
    else:
        return None, None
    pool = [Hamiltonian(nbqbits, [Term(-1.0, letters, list(qs))], do_clean_up=False) for letters, qs in chosen]
    return len(pool), pool


# ---------------------------------------------------------------------------------------------- qubit pools
def qubit_pool(kind, nbqbits, rng=None, source_pool=None, molecule_symbol=None):
    """Single-Pauli-string pools of qubit-ADAPT (ref:openvqe/common_files/qubit_pool.py:278-465, 1184-1268):
    'YXXX' | 'XYXX' | 'XXYX' | 'XXXY' — "YX" on every pair (a, b) with a + b even, then the 4-letter string on every
    quadruple with an even number of odd indices, each as Hamiltonian(n, [Term(-1.0, string, qubits)]);
    'random' — position by position one of the four pools (the reference draws with an unseeded
    np.random.randint; pass ``rng`` for a reproducible draw).  50 operators at 8 qubits
    (ref:tests/test_main_qubit_adapt.py:14).
    The other kinds of the reference's dispatcher (ref:openvqe/common_files/qubit_pool.py:1249-1266): 'two' / 'four' — the
    XXYX family times projectors (1 -+ Z..Z) on its qubits (sums of two / four strings, ref:...:470-697); 'minimal' — the 2n - 2
    operators V of the qubit-ADAPT article (ref:...:906-958); 'pure_with_symmetry' — the eleven H4 strings of ref:...:961-1040
    (``molecule_symbol="H4"``); 'eight' / 'without_Z_from_generator' — ``source_pool`` with every Z dropped from its strings
    (ref:...:790-903; 'eight' also drops repeated operators)."""
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
    elif kind in ("two", "four"):
        pool = _projected_family(nbqbits, kind)
    elif kind == "minimal":
        pool = _minimal_family(nbqbits)
    elif kind == "pure_with_symmetry":
        pool = _pure_with_symmetry(nbqbits, molecule_symbol)
    elif kind in ("eight", "without_Z_from_generator"):
        if source_pool is None:
            raise ValueError(f"qubit pool {kind!r} is derived from another pool: pass source_pool")
        pool = _z_stripped(nbqbits, source_pool, unique=(kind == "eight"))
    elif kind == "random":
        import numpy as np
        rng = rng or np.random.default_rng()
        fams = [family(w) for w in ("YXXX", "XYXX", "XXYX", "XXXY")]
        pool = [fams[int(rng.integers(0, 4))][i] for i in range(len(fams[3]))]
    else:
        raise KeyError(kind)
    return len(pool), pool


def _projected_family(nbqbits, kind):
    """'two' / 'four' pools, written as data: the base string -YX (pairs a + b even) or -XXYX (quadruples with an even number of
    odd indices) times factors (c_I + c_Z Z..Z(qubits)).  'two': one factor on all qubits of the string; 'four': (-1 - ZZZZ) and
    one ZZ factor on the two qubits the spin pattern picks — the same-spin quadruples contribute three operators (ZZ on (c,d),
    (b,d), (a,d)).  Factors are multiplied out in the order (term x Z-part) first, then (term x constant), so the strings
    come out in the order the reference's Hamiltonian products give them."""
    import itertools

    from .operators import Hamiltonian, Term

    def expand(word, qubits, factors):
        terms = [Term(-1.0, word, list(qubits))]
        for c_i, c_z, zq in factors:
            zterm = Term(c_z, "Z" * len(zq), list(zq))
            terms = [t * zterm for t in terms] + [Term(t.coeff * c_i, t.op, list(t.qbits)) for t in terms]
        return Hamiltonian(nbqbits, terms)

    out = []
    for a, b in itertools.combinations(range(nbqbits), 2):
        if (a + b) % 2 == 0:
            out.append(expand("YX", (a, b), [(1.0, -1.0, (a, b))] if kind == "two" else [(-1.0, 1.0, (a, b))]))
    for q in itertools.combinations(range(nbqbits), 4):
        a, b, c, d = q
        if sum(k % 2 for k in q) % 2:
            continue
        if kind == "two":
            out.append(expand("XXYX", q, [(1.0, 1.0, q)]))
            continue
        first = (-1.0, -1.0, q)
        if a % 2 == b % 2 == c % 2 == d % 2:
            picks = [(c, d), (b, d), (a, d)]
        elif a % 2 == b % 2:
            picks = [(c, d)]
        elif a % 2 == c % 2:
            picks = [(b, d)]
        else:
            picks = [(a, d)]
        for pair in picks:
            out.append(expand("XXYX", q, [first, (-1.0, 1.0, pair)]))
    return out


def _minimal_family(nbqbits):
    """V_i = -Y_{k-i} Z_{k-i+1} .. Z_k (i = 0 .. n - 1, k = n - 1) and, for 0 < i < n - 1, the same string without its first Z"""
    from .operators import Hamiltonian, Term
    k = nbqbits - 1
    out = []
    for i in range(nbqbits):
        for skip in ((0,) if i in (0, nbqbits - 1) else (0, 1)):
            zs = list(range(k - i + 1 + skip, k + 1))
            out.append(Hamiltonian(nbqbits, [Term(-1, "Y" + "Z" * len(zs), [k - i] + zs)]))
    return out


#: the symmetry-adapted H4 pool of Shkolnikov et al. (arXiv:2109.05340) as the reference lists it (sign, string over qubits 0..7)
_PURE_H4 = ((-1.0, "YIXIYIYI"), (-1.0, "ZYXIYIZY"), (-1.0, "YIZYXIZY"), (-1.0, "ZZYXYYII"), (1.0, "XXIZIIXY"), (-1.0, "YIZYZXYI"),
            (-1.0, "XIYZYZYI"), (1.0, "XZIIYZII"), (1.0, "ZXXZZXYI"), (1.0, "XXIIIIXY"), (-1.0, "IYYZXIZY"))


def _pure_with_symmetry(nbqbits, molecule_symbol):
    from .operators import Hamiltonian, Term
    if molecule_symbol != "H4":
        return []            # the reference supports H4 only and returns an empty pool otherwise
    if nbqbits != 8:
        raise ValueError("the H4 pool lives on 8 qubits")
    return [Hamiltonian(8, [Term(float(c), word, list(range(8)))]) for c, word in _PURE_H4]


def _z_stripped(nbqbits, source_pool, unique):
    """every operator of ``source_pool`` with the Z letters dropped from its strings and the sign of every coefficient
    flipped (imaginary coefficients contribute their imaginary part: the pools the reference feeds in are i x Hermitian);
    ``unique``: an operator equal to an earlier one, or to its negative, is left out (compared as {string: coefficient})"""
    from .operators import Hamiltonian, Term
    out, seen = [], []
    for op in source_pool:
        if not op.terms:
            continue
        terms = []
        for t in op.terms:
            c = complex(t.coeff)
            value = c.imag if c.real == 0 else c.real
            kept = [(q, letter) for q, letter in zip(t.qbits, t.op) if letter != "Z"]
            terms.append(Term(-1 * value, "".join(letter for _, letter in kept), [q for q, _ in kept]))
        new = Hamiltonian(nbqbits, terms)
        if unique:
            key = {(t.op, tuple(t.qbits)): complex(t.coeff) for t in new.terms}
            neg = {k: -v for k, v in key.items()}
            if any(k == key or k == neg for k in seen):
                continue
            seen.append(key)
        out.append(new)
    return out
