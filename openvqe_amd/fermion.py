"""Fermion -> qubit front-end producing the packed Pauli inputs of the hot path.

The reference obtains these objects from myQLM (``qat.fermion``): the JW-transformed molecular
Hamiltonian (ref:openvqe/common_files/molecule_factory.py:336-350), the UCCSD cluster operators
(ref:openvqe/common_files/generator_excitations.py:40-80, ref:openvqe/algorithms/ucc.py:24-31) and the
HF integer (ref:openvqe/common_files/molecule_factory_with_sparse.py:487-491).  This module restates the
published Jordan-Wigner mapping so that workloads can be built where myQLM / PySCF are absent (the
GPU box): a_p = Z_0..Z_{p-1} (X_p + iY_p)/2, qubit p <-> spin orbital p, spin orbitals interleaved
(alpha even, beta odd — SURVEY.md Appendix A).

Internally a Pauli string is (x, z) bit masks in QUBIT space (bit q <-> qubit q), P = i^{|x&z|} X^x Z^z.
"""
from __future__ import annotations

import itertools

import numpy as np

from .operators import Hamiltonian, Term


# ---------------------------------------------------------------- Pauli-sum algebra on masks
def _mul_string(a, b):
    """(x1,z1)*(x2,z2) -> (phase, (x,z)) with P1 P2 = phase * P12."""
    x1, z1 = a
    x2, z2 = b
    x, z = x1 ^ x2, z1 ^ z2
    k = (bin(x1 & z1).count("1") + bin(x2 & z2).count("1") - bin(x & z).count("1")) % 4
    phase = (1, 1j, -1, -1j)[k]
    if bin(z1 & x2).count("1") & 1:
        phase = -phase
    return phase, (x, z)


def psum_mul(A, B):
    out = {}
    for ka, ca in A.items():
        for kb, cb in B.items():
            ph, k = _mul_string(ka, kb)
            out[k] = out.get(k, 0) + ph * ca * cb
    return out


def psum_add(A, B, scale=1.0):
    out = dict(A)
    for k, c in B.items():
        out[k] = out.get(k, 0) + scale * c
    return out


def psum_iadd(A, B, scale=1.0):
    """A += scale * B in place (same insertion order and arithmetic as psum_add, without the copy)."""
    for k, c in B.items():
        A[k] = A.get(k, 0) + scale * c
    return A


def jw_ladder(p, dagger):
    """a_p (dagger=False) or a_p^dagger as a Pauli sum."""
    zchain = (1 << p) - 1
    xs = (1 << p, zchain)             # Z..Z X_p
    ys = (1 << p, zchain | (1 << p))  # Z..Z Y_p
    return {xs: 0.5, ys: (-0.5j if dagger else 0.5j)}


def jw_product(ladder_ops):
    """JW image of a product of ladder operators [(p, dagger), ...] (left to right)."""
    out = {(0, 0): 1.0}
    for p, dag in ladder_ops:
        out = psum_mul(out, jw_ladder(p, dag))
    return out


# ---------------------------------------------------------------- linear fermion -> qubit encodings
# qubit occupation vector q = beta f (mod 2), f = orbital occupations.  Jordan-Wigner: beta = 1; parity basis: q_i =
# f_0 + ... + f_i; Bravyi-Kitaev: the binary-tree (Fenwick) matrix.  The transforms the reference offers next to JW
# (ref:openvqe/common_files/molecule_factory.py:349-356, ref:…generator_excitations.py:17-22).
def encoding_matrix(nqbits, transform):
    if transform == "JW":
        return np.eye(nqbits, dtype=np.uint8)
    if transform == "parity_basis":
        return np.tril(np.ones((nqbits, nqbits), dtype=np.uint8))
    if transform == "Bravyi-Kitaev":
        size = 1
        beta = np.ones((1, 1), dtype=np.uint8)
        while size < nqbits:                      # beta_{2m} = [[beta_m, 0], [0 with last row 1, beta_m]]
            big = np.zeros((2 * size, 2 * size), dtype=np.uint8)
            big[:size, :size] = beta
            big[size:, size:] = beta
            big[2 * size - 1, :size] = 1
            beta, size = big, 2 * size
        return beta[:nqbits, :nqbits].copy()
    raise ValueError(f"unknown transform '{transform}'")


def _gf2_inverse(m):
    n = m.shape[0]
    a = np.concatenate([m.copy() % 2, np.eye(n, dtype=np.uint8)], axis=1)
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r, c])
        a[[c, piv]] = a[[piv, c]]
        for r in range(n):
            if r != c and a[r, c]:
                a[r] ^= a[c]
    return a[:, n:]


_LADDERS = {}


def encoded_ladder(nqbits, p, dagger, transform):
    """a_p / a+_p as a Pauli sum under a linear encoding: with c = X_{column p of beta} Z_{parity set of p} (the operator that
    flips occupation p with the sign of the orbitals below it) and n_p read from the flip set F(p) (row p of beta^-1),
    a_p = c (1 - Z_F) / 2 and a+_p = c (1 + Z_F) / 2."""
    key = (nqbits, transform)
    if key not in _LADDERS:
        beta = encoding_matrix(nqbits, transform)
        inv = _gf2_inverse(beta)
        cols = [sum(1 << i for i in range(nqbits) if beta[i, j]) for j in range(nqbits)]
        flips = [sum(1 << i for i in range(nqbits) if inv[j, i]) for j in range(nqbits)]
        below = np.cumsum(np.vstack([np.zeros((1, nqbits), dtype=np.int64), inv[:-1].astype(np.int64)]), axis=0) % 2
        pars = [sum(1 << i for i in range(nqbits) if below[j, i]) for j in range(nqbits)]
        _LADDERS[key] = (cols, pars, flips)
    cols, pars, flips = _LADDERS[key]
    x, zp, zf = cols[p], pars[p], pars[p] ^ flips[p]

    def string(z):   # the operator product X^x Z^z written as coefficient * (Hermitian string (x, z))
        return (x, z), (-1j) ** (bin(x & z).count("1") % 4)
    (k1, c1), (k2, c2) = string(zp), string(zf)
    out = {k1: 0.5 * c1}
    out[k2] = out.get(k2, 0) + (0.5 if dagger else -0.5) * c2
    return out


def encoded_product(nqbits, ladder_ops, transform):
    if transform == "JW":
        return jw_product(ladder_ops)
    out = {(0, 0): 1.0}
    for p, dag in ladder_ops:
        out = psum_mul(out, encoded_ladder(nqbits, p, dag, transform))
    return out


def recode_occupation(integer, nqbits, transform):
    """occupation integer (orbital 0 = most significant bit) -> basis index of the encoded qubit register"""
    if transform == "JW":
        return int(integer)
    beta = encoding_matrix(nqbits, transform)
    f = np.array([(int(integer) >> (nqbits - 1 - q)) & 1 for q in range(nqbits)], dtype=np.int64)
    q = (beta.astype(np.int64) @ f) % 2
    return int(sum(int(b) << (nqbits - 1 - i) for i, b in enumerate(q)))


def psum_to_hamiltonian(nqbits, psum, constant=0.0, tol=1e-13, real=False):
    """dict[(x,z)] -> Hamiltonian (insertion order kept); the identity string goes to the constant."""
    terms = []
    const = constant
    for (x, z), c in psum.items():
        if abs(c) <= tol:
            continue
        if x == 0 and z == 0:
            const = const + c
            continue
        qs = [q for q in range(nqbits) if ((x | z) >> q) & 1]
        op = "".join("Y" if ((x >> q) & 1 and (z >> q) & 1) else "X" if (x >> q) & 1 else "Z" for q in qs)
        if real:
            if abs(complex(c).imag) > 1e-10:
                raise ValueError("non-real coefficient in a Hermitian operator")
            c = complex(c).real
        terms.append(Term(c, op, qs))
    if real:
        const = complex(const).real
    return Hamiltonian(nqbits, terms, const, do_clean_up=False)


# ---------------------------------------------------------------- molecular Hamiltonian
def spin_orbital_integrals(h1_spatial, eri_chem):
    """Spatial integrals -> spin-orbital (interleaved) h_pq and physicist-ordered h_pqrs with
    H = sum h_pq a+_p a_q + 1/2 sum h_pqrs a+_p a+_q a_r a_s  (h_pqrs = (ps|qr) in chemist notation)."""
    m = h1_spatial.shape[0]
    n = 2 * m
    hpq = np.zeros((n, n))
    hpqrs = np.zeros((n, n, n, n))
    for p in range(n):
        for q in range(n):
            if p % 2 == q % 2:
                hpq[p, q] = h1_spatial[p // 2, q // 2]
    for p, q, r, s in itertools.product(range(n), repeat=4):
        if p % 2 == s % 2 and q % 2 == r % 2:
            hpqrs[p, q, r, s] = eri_chem[p // 2, s // 2, q // 2, r // 2]
    return hpq, hpqrs


def jw_molecular_hamiltonian(hpq, hpqrs, constant=0.0, tol=1e-12, transform="JW"):
    """qubit image of the electronic-structure Hamiltonian (spin-orbital integrals); Jordan-Wigner by default,
    "Bravyi-Kitaev" / "parity_basis" through the linear-encoding ladder operators"""
    n = hpq.shape[0]
    if transform == "JW":
        ladders_c = [jw_ladder(p, True) for p in range(n)]
        ladders_a = [jw_ladder(p, False) for p in range(n)]
    else:
        ladders_c = [encoded_ladder(n, p, True, transform) for p in range(n)]
        ladders_a = [encoded_ladder(n, p, False, transform) for p in range(n)]
    total = {}
    for p in range(n):
        for q in range(n):
            if abs(hpq[p, q]) > tol:
                psum_iadd(total, psum_mul(ladders_c[p], ladders_a[q]), hpq[p, q])
    # cache a+_p a+_q and a_r a_s
    cc = {}
    aa = {}
    for p in range(n):
        for q in range(n):
            if p != q:
                cc[(p, q)] = psum_mul(ladders_c[p], ladders_c[q])
                aa[(p, q)] = psum_mul(ladders_a[p], ladders_a[q])
    for p, q in cc:
        for r, s in aa:
            v = hpqrs[p, q, r, s]
            if abs(v) > tol:
                psum_iadd(total, psum_mul(cc[(p, q)], aa[(r, s)]), 0.5 * v)
    return psum_to_hamiltonian(n, total, constant, tol=tol, real=True)


def hf_integer(nqbits, n_electrons):
    """HF determinant: lowest n_electrons spin orbitals occupied; qubit q <-> bit (n-1-q)."""
    return sum(1 << (nqbits - 1 - q) for q in range(n_electrons))


# ---------------------------------------------------------------- UCCSD generators
def _excitation_generator(nqbits, creators, annihilators):
    """Hermitian generator i(T - T^+) for T = prod a+_creators prod a_annihilators, as a Hamiltonian
    with real coefficients (what ucc_action receives after the `*1j` of ref:openvqe/algorithms/ucc.py:31)."""
    t = jw_product([(p, True) for p in creators] + [(p, False) for p in annihilators])
    tdag = jw_product([(p, True) for p in reversed(annihilators)] + [(p, False) for p in reversed(creators)])
    anti = psum_add(t, tdag, -1.0)
    herm = {k: 1j * c for k, c in anti.items()}
    return psum_to_hamiltonian(nqbits, herm, tol=1e-13, real=True)


def _anti_excitation(nqbits, creators, annihilators):
    """Anti-Hermitian T - T^+ (the un-multiplied pool operator whose sparse matrix the ADAPT screen uses)."""
    t = jw_product([(p, True) for p in creators] + [(p, False) for p in annihilators])
    tdag = jw_product([(p, True) for p in reversed(annihilators)] + [(p, False) for p in reversed(creators)])
    return psum_to_hamiltonian(nqbits, psum_add(t, tdag, -1.0), tol=1e-13)


def uccsd_excitations(n_spatial, n_occ_spatial):
    """Spin-conserving single and double excitations (occupied -> virtual), spin orbitals interleaved.
    Counts: singles 2*o*v, doubles 2*C(o,2)*C(v,2) + (o*v)^2."""
    nso = 2 * n_spatial
    nocc = 2 * n_occ_spatial
    occ = list(range(nocc))
    virt = list(range(nocc, nso))
    singles = [(i, a) for i in occ for a in virt if i % 2 == a % 2]
    doubles = []
    for i, j in itertools.combinations(occ, 2):
        for a, b in itertools.combinations(virt, 2):
            # spin conservation: multiset of spins equal
            if sorted((i % 2, j % 2)) == sorted((a % 2, b % 2)):
                doubles.append((i, j, a, b))
    return singles, doubles


def uccsd_generators(n_spatial, n_occ_spatial):
    """Hermitian JW generators of UCCSD: singles then doubles (2 resp. 8 Pauli strings each)."""
    nq = 2 * n_spatial
    singles, doubles = uccsd_excitations(n_spatial, n_occ_spatial)
    gens = [_excitation_generator(nq, [a], [i]) for i, a in singles]
    gens += [_excitation_generator(nq, [b, a], [i, j]) for i, j, a, b in doubles]
    return gens


def uccsd_pool_antihermitian(n_spatial, n_occ_spatial):
    nq = 2 * n_spatial
    singles, doubles = uccsd_excitations(n_spatial, n_occ_spatial)
    pool = [_anti_excitation(nq, [a], [i]) for i, a in singles]
    pool += [_anti_excitation(nq, [b, a], [i, j]) for i, j, a, b in doubles]
    return pool


# ---------------------------------------------------------------- synthetic molecule-shaped workloads
def synthetic_integrals(n_spatial, seed, h_scale=0.5, eri_scale=0.1):
    """Random symmetric h_pq ~ N(0, h_scale^2) and 8-fold-symmetric (pq|rs) ~ N(0, eri_scale^2)
    (generic spin-conserving integrals without point-group symmetry)."""
    rng = np.random.default_rng(seed)
    m = n_spatial
    h = rng.normal(0.0, h_scale, (m, m))
    h = 0.5 * (h + h.T)
    g = rng.normal(0.0, eri_scale, (m, m, m, m))
    g = g + g.transpose(1, 0, 2, 3)
    g = g + g.transpose(0, 1, 3, 2)
    g = g + g.transpose(2, 3, 0, 1)
    return h, g / 8.0


def synthetic_molecule(n_spatial, n_occ_spatial, seed):
    """(hamiltonian, uccsd generators, hf integer) of a molecule-shaped synthetic problem."""
    h, g = synthetic_integrals(n_spatial, seed)
    hpq, hpqrs = spin_orbital_integrals(h, g)
    ham = jw_molecular_hamiltonian(hpq, hpqrs, 0.0)
    gens = uccsd_generators(n_spatial, n_occ_spatial)
    return ham, gens, hf_integer(2 * n_spatial, 2 * n_occ_spatial)


# ---------------------------------------------------------------- UCCSD in the reference's operator order + MP2 guess
def cluster_excitations(nqbits, n_elec, order="auto"):
    """Excitation tuples in the ORDER and FORM in which ``qat.fermion…get_cluster_ops_and_init_guess`` hands the
    cluster operators to the reference (ref:openvqe/common_files/generator_excitations.py:75-79) — third-party code that
    is not in the reference tree.  Form (both stored runs): singles (a, i), doubles (a, b, i, j) with a < b virtual,
    i < j occupied, same total spin; ``op.terms[0].qbits`` of cluster operator k is exactly tuple k
    (ref:openvqe/ucc_family/get_energy_qucc.py:46-49).  Order, as pinned by the two stored QUCCSD runs
    (tests/test_reference_traces.py::test_k5_*, energies to < 1e-8, CNOT counts 292 / 70):
      "descending" — H4/STO-3G full space, 8 qubits / 4 electrons (ref:notebooks/demo_quccsd.ipynb): singles by
          (A, I) spatial descending, beta before alpha; doubles in decreasing lexicographic order of (a, b, i, j);
      "ascending"  — H4 active space, 6 qubits / 2 electrons (ref:notebooks/demo_quccsd_active_space.ipynb): singles by
          A ascending, beta before alpha; doubles in increasing lexicographic order of (b, a, j, i) — the mirror image.
    No single enumeration rule reproducing both was found (per-orbital sort keys cannot: the stored first-order
    gradients show the LOWEST virtual orbital first in one run and the HIGHEST in the other), so "auto" keeps the
    observed split: descending when two or more spatial orbitals are occupied, ascending for a single occupied orbital."""
    if order == "auto":
        order = "descending" if n_elec >= 4 else "ascending"
    occ = list(range(n_elec))
    virt = list(range(n_elec, nqbits))
    no, nv = n_elec // 2, (nqbits - n_elec) // 2
    doubles = [(a, b, i, j) for a, b in itertools.combinations(virt, 2) for i, j in itertools.combinations(occ, 2)
               if (a % 2) + (b % 2) == (i % 2) + (j % 2)]
    if order == "descending":
        singles = [(2 * (no + A) + s, 2 * I + s) for A in reversed(range(nv)) for I in reversed(range(no)) for s in (1, 0)]
        doubles.sort(reverse=True)
    elif order == "ascending":
        singles = [(2 * (no + A) + s, 2 * I + s) for A in range(nv) for I in range(no) for s in (1, 0)]
        doubles.sort(key=lambda t: (t[1], t[0], t[3], t[2]))
    else:
        raise ValueError(order)
    return singles, doubles


def cluster_ops_and_mp2_guess(n_elec, orb_energies_full, hpqrs):
    """(cluster_ops, theta_MP2, hf_init) — the triple of ``get_cluster_ops_and_init_guess(n_elec, noons, orbital
    energies, hpqrs)`` (call sites ref:…generator_excitations.py:75-79, ref:openvqe/common_files/molecule_factory.py:
    487-491).  cluster_ops: Hermitian fermionic generators i(T - T^+) ("we must skip 1j", ref:openvqe/algorithms/ucc.py:
    26), T = c+_a c_i resp. c+_a c+_b c_i c_j on the tuples of ``cluster_excitations``;
    theta_MP2: 0 for singles, (h_abij - h_abji) / (e_i + e_j - e_a - e_b) for doubles, with H = sum h_pq c+_p c_q +
    1/2 sum h_pqrs c+_p c+_q c_r c_s (the Moller-Plesset amplitude of that T; E(theta_MP2) of the stored run is
    reproduced to 3e-9); hf_init: the first n_elec spin orbitals occupied, orbital 0 = most significant bit."""
    from .fermionic import FermionHamiltonian, Term
    nq = len(orb_energies_full)
    singles, doubles = cluster_excitations(nq, n_elec)
    e = orb_energies_full
    ops, theta = [], []
    for a, i in singles:
        ops.append(FermionHamiltonian(nq, [Term(1j, "Cc", [a, i]), Term(-1j, "Cc", [i, a])], do_clean_up=False))
        theta.append(0.0)
    for a, b, i, j in doubles:
        ops.append(FermionHamiltonian(nq, [Term(1j, "CCcc", [a, b, i, j]), Term(-1j, "CCcc", [j, i, b, a])],
                                      do_clean_up=False))
        theta.append(float((hpqrs[a, b, i, j] - hpqrs[a, b, j, i]) / (e[i] + e[j] - e[a] - e[b])))
    return ops, theta, hf_integer(nq, n_elec)
