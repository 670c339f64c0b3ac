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


def quccsd_like_gates(rng, n, n_single, n_double, extra_random=0, disjoint_ladders=False):
    """literal gate list of the reference's fermionic QUCCSD templates on random excitations (traced through
    openvqe_amd.common_files.circuit with symbolic angles) + optional random literal gates.
    -> (gates [(name, qubits, scale, const, pidx)], K)"""
    from openvqe_amd.common_files.circuit import efficient_fermionic_ansatz
    from openvqe_amd.qat_compat import AffineParam, Program, lower_circuit
    exci = []
    for _ in range(n_single):
        a, b = sorted(rng.choice(n, 2, replace=False).tolist())
        exci.append([a, b])
    for _ in range(n_double):
        q = rng.choice(n, 4, replace=False).tolist()
        # occupied pair below the virtual pair (the UCCSD index order) keeps the two CNOT ladders apart; interleaved
        # ladders, which the reference's ladder code does not undo exactly, otherwise
        exci.append(sorted(q) if disjoint_ladders else sorted(q[:2]) + sorted(q[2:]))
    order = rng.permutation(len(exci))
    exci = [exci[i] for i in order]
    K = len(exci)
    prog = Program()
    reg = prog.qalloc(n)
    efficient_fermionic_ansatz(reg, prog, exci, [AffineParam(k) for k in range(K)])
    _, kind, gates = lower_circuit(prog.to_circ())
    assert kind == "gates"
    gates = list(gates)
    for _ in range(extra_random):
        name = str(rng.choice(["X", "H", "RX", "RY", "RZ", "CNOT"]))
        pos = int(rng.integers(0, len(gates) + 1))
        if name == "CNOT":
            c, t = rng.choice(n, 2, replace=False).tolist()
            g = (name, [c, t], 0.0, 0.0, -1)
        elif name in ("X", "H"):
            g = (name, [int(rng.integers(0, n))], 0.0, 0.0, -1)
        else:
            g = (name, [int(rng.integers(0, n))], float(rng.choice([1.0, -1.0, -2.0])), float(rng.uniform(-1, 1)),
                 int(rng.integers(-1, K)))
        gates.insert(pos, g)
    return gates, K
