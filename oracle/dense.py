"""TEST INFRASTRUCTURE ONLY — dense/Kronecker CPU oracle (numpy + scipy).

Restates, with explicit 2^n x 2^n matrices, what the reference's hot path
computes.  This is deliberately a *different algorithm* from the bit-mask
kernels of the product (and of ``oracle/masks.py`` / ``oracle/c``): operators
are built by Kronecker products exactly the way the reference's own scipy
code does it, and exponentials are taken with ``scipy.linalg.expm``.

Reference lines followed (``ref:`` = /root/reference/):
  * operator -> matrix, qubit 0 is the LEFTMOST Kronecker factor:
    ref:openvqe/adapt/qubit_adapt_vqe.py:81-123 (term_to_matrix_sparse)
  * exp(-i theta P) |psi>:  ref:openvqe/adapt/qubit_adapt_vqe.py:20-55
  * exp(theta A) |psi> (anti-Hermitian A): ref:openvqe/adapt/fermionic_adapt_vqe.py:12-38
  * <psi|H|psi>: ref:openvqe/adapt/qubit_adapt_vqe.py:424-426
  * fermionic ADAPT gradient 2 Re(sig^+ A psi): ref:openvqe/adapt/fermionic_adapt_vqe.py:41-122
  * qubit ADAPT gradient 2 |psi^+ H P psi|: ref:openvqe/adapt/qubit_adapt_vqe.py:126-150
  * energy of the UCC circuit  prod_k prod_j exp(-i theta_k c_kj P_kj)|HF>:
    ref:openvqe/ucc_family/get_energy_ucc.py:35-50 (loop + OBS job); the inner
    product form is myQLM's one-step Trotterisation in ``terms`` order.
  * HF integer -> basis index (bit n-1-q <-> qubit q):
    ref:openvqe/ucc_family/get_energy_qucc.py:40-45,
    ref:openvqe/common_files/molecule_factory_with_sparse.py:622-642
  * gate set of ref:openvqe/common_files/circuit.py:1 with myQLM conventions
    RX(a)=exp(-iaX/2), RY(a)=exp(-iaY/2), RZ(a)=diag(e^{-ia/2}, e^{ia/2}),
    apply(CNOT, control, target).

Operators are duck-typed: anything with ``nbqbits``, ``terms`` (each with
``coeff``, ``op``, ``qbits``) and optionally ``constant_coeff``.
"""
from collections import namedtuple

import numpy as np
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

OTerm = namedtuple("OTerm", "coeff op qbits")


class OHam:
    """Minimal operator container for oracle-side tests."""

    def __init__(self, nbqbits, terms, constant_coeff=0.0):
        self.nbqbits = int(nbqbits)
        self.terms = [OTerm(complex(c), str(o), list(q)) for (c, o, q) in terms]
        self.constant_coeff = constant_coeff


_PAULI = {
    "I": np.array([[1, 0], [0, 1]], dtype=complex),
    "X": np.array([[0, 1], [1, 0]], dtype=complex),
    "Y": np.array([[0, -1j], [1j, 0]], dtype=complex),
    "Z": np.array([[1, 0], [0, -1]], dtype=complex),
}


def pauli_string_matrix(nbqbits, op, qbits, sparse=True):
    """Kronecker product with qubit 0 leftmost (qubit_adapt_vqe.py:103-120)."""
    factors = ["I"] * nbqbits
    for ch, q in zip(op, qbits):
        factors[q] = ch
    mat = None
    for ch in factors:
        f = scipy.sparse.csr_matrix(_PAULI[ch])
        mat = f if mat is None else scipy.sparse.kron(mat, f, format="csr")
    return mat if sparse else mat.toarray()


def operator_matrix(operator, sparse=True, with_constant=True):
    """Sum_t coeff_t * kron(...) (+ constant * I)."""
    n = operator.nbqbits
    dim = 1 << n
    total = scipy.sparse.csr_matrix((dim, dim), dtype=complex)
    for term in operator.terms:
        total = total + complex(term.coeff) * pauli_string_matrix(n, term.op, term.qbits)
    const = getattr(operator, "constant_coeff", 0.0) or 0.0
    if with_constant and const != 0:
        total = total + complex(const) * scipy.sparse.identity(dim, dtype=complex, format="csr")
    total = scipy.sparse.csr_matrix(total)
    return total if sparse else total.toarray()


def basis_state(nbqbits, hf_init):
    """|HF>: X on qubit q iff bit (n-1-q) of hf_init is set == unit vector at index hf_init."""
    psi = np.zeros(1 << nbqbits, dtype=complex)
    psi[int(hf_init)] = 1.0
    return psi


def pauli_rotation(psi, nbqbits, op, qbits, phi):
    """exp(-i phi P) psi through scipy's matrix exponential (qubit_adapt_vqe.py:49-54)."""
    pmat = pauli_string_matrix(nbqbits, op, qbits).toarray()
    return scipy.linalg.expm(-1j * phi * pmat) @ psi


def ucc_state(nbqbits, hf_init, generators, thetas):
    """prod_k prod_{j in terms order} exp(-i theta_k c_kj P_kj) |HF>.

    ``zip`` truncation to min(len(generators), len(thetas)) as in
    get_energy_ucc.py:42.  Generator coefficients must be real (Hermitian
    generators).
    """
    psi = basis_state(nbqbits, hf_init)
    for gen, theta in zip(generators, thetas):
        for term in gen.terms:
            c = complex(term.coeff)
            if abs(c.imag) > 1e-12:
                raise ValueError("generator with complex Pauli coefficient")
            psi = pauli_rotation(psi, nbqbits, term.op, term.qbits, float(theta) * c.real)
    return psi


def expectation(operator, psi):
    """Re <psi|H|psi> including the constant term."""
    hmat = operator_matrix(operator, sparse=True)
    return float(np.real(np.vdot(psi, hmat @ psi)))


def ucc_energy(hamiltonian, generators, hf_init, thetas):
    psi = ucc_state(hamiltonian.nbqbits, hf_init, generators, thetas)
    return expectation(hamiltonian, psi)


# ---------------------------------------------------------------- gate level
def gate_matrix(name, angle=None):
    if name == "X":
        return _PAULI["X"]
    if name == "Y":
        return _PAULI["Y"]
    if name == "Z":
        return _PAULI["Z"]
    if name == "H":
        return np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2.0)
    if name == "RX":
        return scipy.linalg.expm(-0.5j * angle * _PAULI["X"])
    if name == "RY":
        return scipy.linalg.expm(-0.5j * angle * _PAULI["Y"])
    if name == "RZ":
        return scipy.linalg.expm(-0.5j * angle * _PAULI["Z"])
    raise ValueError(name)


def apply_gate(psi, nbqbits, name, qubits, angle=None):
    """Apply one gate through its full 2^n matrix (Kronecker embedding)."""
    dim = 1 << nbqbits
    if name == "CNOT":
        c, t = qubits
        p0 = np.array([[1, 0], [0, 0]], dtype=complex)
        p1 = np.array([[0, 0], [0, 1]], dtype=complex)

        def emb(mats):
            out = np.array([[1.0 + 0j]])
            for q in range(nbqbits):
                out = np.kron(out, mats.get(q, _PAULI["I"]))
            return out

        full = emb({c: p0}) + emb({c: p1, t: _PAULI["X"]})
    else:
        g = gate_matrix(name, angle)
        full = np.array([[1.0 + 0j]])
        for q in range(nbqbits):
            full = np.kron(full, g if q == qubits[0] else _PAULI["I"])
    assert full.shape == (dim, dim)
    return full @ psi


def gate_circuit_state(nbqbits, hf_init, gates):
    """gates: iterable of (name, qubits, angle_or_None) applied in order on |HF>."""
    psi = basis_state(nbqbits, hf_init)
    for name, qubits, angle in gates:
        psi = apply_gate(psi, nbqbits, name, list(qubits), angle)
    return psi


# ------------------------------------------------------------- ADAPT screens
def exact_exp_state(reference_ket, sparse_generators, parameters):
    """prod_k expm(theta_k A_k) ref  (fermionic_adapt_vqe.py:35-38)."""
    state = np.asarray(reference_ket, dtype=complex).reshape(-1) * 1.0
    for theta, amat in zip(parameters, sparse_generators):
        state = scipy.sparse.linalg.expm_multiply(theta * scipy.sparse.csc_matrix(amat), state)
    return state


def qubit_exp_state(reference_ket, operators, coefficients):
    """prod_k expm(-i theta_k * matrix(op_k)) ref  (qubit_adapt_vqe.py:42-54)."""
    state = np.asarray(reference_ket, dtype=complex).reshape(-1) * 1.0
    for theta, op in zip(coefficients, operators):
        mat = operator_matrix(op, sparse=False, with_constant=False)
        state = scipy.linalg.expm(-1j * theta * mat) @ state
    return state


def fermionic_pool_gradients(pool_sparse, hamiltonian_sparse, state):
    """g_i = 2 Re(sig^+ A_i psi), sig = H psi (fermionic_adapt_vqe.py:67-73,114)."""
    sig = hamiltonian_sparse @ state
    out = []
    for amat in pool_sparse:
        gi = 2.0 * np.vdot(sig, amat @ state)
        out.append(float(gi.real))
    return out


def qubit_pool_gradients(pool_ops, hamiltonian_sparse, state):
    """g_i = 2 |psi^+ H P_i psi| (qubit_adapt_vqe.py:147-150)."""
    out = []
    for op in pool_ops:
        pmat = operator_matrix(op, sparse=True, with_constant=False)
        val = np.vdot(state, hamiltonian_sparse @ (pmat @ state))
        out.append(2.0 * float(np.abs(val)))
    return out
