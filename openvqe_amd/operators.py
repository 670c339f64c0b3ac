"""Host-side Pauli-operator containers for the statevector backend.

These are the objects the reference's L1 code receives from ``qat.core`` /
``qat.fermion`` and only ever touches through a small protocol
(SURVEY.md §8b): ``.nbqbits``, ``.terms[i].{coeff,op,qbits}``,
``.constant_coeff``, scalar ``*`` / ``+`` / ``/``, ``get_matrix(sparse=...)``
(call sites: ref:openvqe/ucc_family/get_energy_ucc.py:40-47,
ref:openvqe/algorithms/ucc.py:30-31, ref:openvqe/adapt/fermionic_adapt_vqe.py:474,554,
ref:openvqe/adapt/qubit_adapt_vqe.py:98-122).  ``Term`` / ``Hamiltonian`` here
are duck-type compatible with them, so either kind can be handed to the
backend.

Packing for the device follows SURVEY.md Appendix A: reference qubit q maps to
basis-index bit (n-1-q); a string becomes two uint64 index-space masks (x, z)
with I=(0,0) X=(1,0) Z=(0,1) Y=(1,1).
"""
from __future__ import annotations

import numpy as np

_PAULI_MUL = {  # (a, b) -> (phase, c)  with a*b = phase * c
    ("I", "I"): (1, "I"), ("I", "X"): (1, "X"), ("I", "Y"): (1, "Y"), ("I", "Z"): (1, "Z"),
    ("X", "I"): (1, "X"), ("Y", "I"): (1, "Y"), ("Z", "I"): (1, "Z"),
    ("X", "X"): (1, "I"), ("Y", "Y"): (1, "I"), ("Z", "Z"): (1, "I"),
    ("X", "Y"): (1j, "Z"), ("Y", "X"): (-1j, "Z"),
    ("Y", "Z"): (1j, "X"), ("Z", "Y"): (-1j, "X"),
    ("Z", "X"): (1j, "Y"), ("X", "Z"): (-1j, "Y"),
}


class Term:
    """One Pauli string: ``coeff * op[0]_{qbits[0]} op[1]_{qbits[1]} ...``."""

    __slots__ = ("coeff", "op", "qbits")

    def __init__(self, coefficient, pauli_op, qbits):
        if len(pauli_op) != len(qbits):
            raise ValueError("Term: len(pauli_op) != len(qbits)")
        self.coeff = coefficient
        self.op = str(pauli_op)
        self.qbits = [int(q) for q in qbits]

    def copy(self):
        return Term(self.coeff, self.op, list(self.qbits))

    @property
    def _coeff(self):
        """myQLM keeps a term's coefficient as a serialisable record, and one reference helper reads it directly
        (``term._coeff.complex_p.re`` / ``.im``, ref:openvqe/common_files/qubit_pool.py:729-732): the same shape here"""
        import types
        c = complex(self.coeff)
        return types.SimpleNamespace(complex_p=types.SimpleNamespace(re=c.real, im=c.imag))

    def _canonical(self):
        """(sorted qubits, ops) with identities dropped; key for merging."""
        pairs = sorted((q, c) for q, c in zip(self.qbits, self.op) if c != "I")
        return tuple(pairs)

    def __mul__(self, other):
        if isinstance(other, Term):
            ops = {}
            phase = 1
            for q, c in zip(self.qbits, self.op):
                ops[q] = c
            for q, c in zip(other.qbits, other.op):
                ph, r = _PAULI_MUL[(ops.get(q, "I"), c)]
                phase *= ph
                ops[q] = r
            qs = sorted(q for q, c in ops.items() if c != "I")
            return Term(self.coeff * other.coeff * phase, "".join(ops[q] for q in qs), qs)
        return Term(self.coeff * other, self.op, list(self.qbits))

    __rmul__ = lambda self, other: Term(other * self.coeff, self.op, list(self.qbits))

    def __repr__(self):
        return f"{self.coeff} * ({self.op}|{self.qbits})"


class Hamiltonian:
    """Pauli-sum operator on ``nqbits`` qubits (spin representation)."""

    def __init__(self, nqbits, terms=(), constant_coeff=0.0, do_clean_up=True):
        self.nbqbits = int(nqbits)
        self.constant_coeff = constant_coeff
        self.terms = [t.copy() if isinstance(t, Term) else Term(t.coeff, t.op, t.qbits) for t in terms]
        for t in self.terms:
            if t.qbits and max(t.qbits) >= self.nbqbits:
                raise ValueError("Term acts outside the register")
        if do_clean_up:
            self._clean_up()

    # -- construction helpers ------------------------------------------------
    def _clean_up(self, threshold=0.0):
        """Merge equal strings (first-appearance order), fold identities into the constant."""
        merged = {}
        order = []
        const = self.constant_coeff
        for t in self.terms:
            key = t._canonical()
            if not key:
                const = const + t.coeff
                continue
            if key in merged:
                merged[key] = merged[key] + t.coeff
            else:
                merged[key] = t.coeff
                order.append(key)
        self.constant_coeff = const
        self.terms = [
            Term(merged[k], "".join(c for _, c in k), [q for q, _ in k])
            for k in order
            if abs(merged[k]) > threshold
        ]

    @classmethod
    def from_pauli_dict(cls, pauli_dict):
        """{'XIZY': coeff, ...}; string position q <-> qubit q
        (ref:openvqe/applications/quantum_batteries/utils.py:13-24)."""
        n = len(next(iter(pauli_dict)))
        terms = []
        for s, c in pauli_dict.items():
            qs = [q for q, ch in enumerate(s) if ch != "I"]
            terms.append(Term(c, "".join(s[q] for q in qs), qs))
        return cls(n, terms)

    def copy(self):
        return Hamiltonian(self.nbqbits, self.terms, self.constant_coeff, do_clean_up=False)

    # -- arithmetic used by the reference (ucc.py:31, generator_excitations.py:235,354) --
    def __mul__(self, other):
        if isinstance(other, Hamiltonian):
            if other.nbqbits != self.nbqbits:
                raise ValueError("qubit count mismatch")
            terms = [a * b for a in self.terms for b in other.terms]
            terms += [Term(b.coeff * self.constant_coeff, b.op, b.qbits) for b in other.terms
                      if self.constant_coeff != 0]
            terms += [Term(a.coeff * other.constant_coeff, a.op, a.qbits) for a in self.terms
                      if other.constant_coeff != 0]
            return Hamiltonian(self.nbqbits, terms, self.constant_coeff * other.constant_coeff)
        return Hamiltonian(self.nbqbits, [Term(t.coeff * other, t.op, t.qbits) for t in self.terms],
                           self.constant_coeff * other, do_clean_up=False)

    def __rmul__(self, other):
        return Hamiltonian(self.nbqbits, [Term(other * t.coeff, t.op, t.qbits) for t in self.terms],
                           other * self.constant_coeff, do_clean_up=False)

    def __truediv__(self, other):
        return self * (1.0 / other)

    def __neg__(self):
        return self * (-1.0)

    def __add__(self, other):
        if isinstance(other, Hamiltonian):
            if other.nbqbits != self.nbqbits:
                raise ValueError("qubit count mismatch")
            return Hamiltonian(self.nbqbits, list(self.terms) + list(other.terms),
                               self.constant_coeff + other.constant_coeff)
        return Hamiltonian(self.nbqbits, self.terms, self.constant_coeff + other, do_clean_up=False)

    __radd__ = __add__

    def __sub__(self, other):
        return self + (other * (-1.0))

    def dag(self):
        return Hamiltonian(self.nbqbits, [Term(np.conj(t.coeff), t.op, t.qbits) for t in self.terms],
                           np.conj(self.constant_coeff), do_clean_up=False)

    # -- packing ------------------------------------------------------------
    def packed(self):
        """(xmask[T] u64, zmask[T] u64, coeff[T] c128) in basis-index bit space."""
        return pack_terms(self.nbqbits, self.terms)

    def get_matrix(self, sparse=False):
        """2^n x 2^n matrix, qubit 0 = most significant index bit
        (ref:openvqe/adapt/qubit_adapt_vqe.py:103-120 ordering)."""
        import scipy.sparse

        n = self.nbqbits
        dim = 1 << n
        xs, zs, cs = self.packed()
        idx = np.arange(dim, dtype=np.uint64)
        rows, cols, vals = [], [], []
        for x, z, c in zip(xs, zs, cs):
            col = idx ^ x
            par = col & z
            for s in (32, 16, 8, 4, 2, 1):
                par ^= par >> np.uint64(s)
            sign = 1.0 - 2.0 * (par & np.uint64(1)).astype(np.float64)
            ny = bin(int(x & z)).count("1") % 4
            rows.append(idx)
            cols.append(col)
            vals.append(c * (1j ** ny) * sign)
        if self.constant_coeff != 0:
            rows.append(idx)
            cols.append(idx)
            vals.append(np.full(dim, self.constant_coeff, dtype=complex))
        if rows:
            mat = scipy.sparse.coo_matrix(
                (np.concatenate(vals), (np.concatenate(rows).astype(np.int64),
                                        np.concatenate(cols).astype(np.int64))),
                shape=(dim, dim), dtype=complex).tocsr()
        else:
            mat = scipy.sparse.csr_matrix((dim, dim), dtype=complex)
        return mat if sparse else mat.toarray()

    def __repr__(self):
        head = f"{self.constant_coeff} * I^{self.nbqbits}"
        return " +\n".join([head] + [repr(t) for t in self.terms])


# qat.fermion.SpinHamiltonian is the same protocol (qubit_pool.py builds pools with it)
SpinHamiltonian = Hamiltonian
Observable = Hamiltonian


def pack_string(nbqbits, op, qbits):
    x = 0
    z = 0
    for ch, q in zip(op, qbits):
        if not 0 <= q < nbqbits:
            raise ValueError("qubit index out of range")
        bit = 1 << (nbqbits - 1 - q)
        if ch == "X":
            x |= bit
        elif ch == "Y":
            x |= bit
            z |= bit
        elif ch == "Z":
            z |= bit
        elif ch != "I":
            raise ValueError(f"unknown Pauli '{ch}'")
    return x, z


def pack_terms(nbqbits, terms):
    """(x masks, z masks, coefficients) of a list of Pauli strings, whole list at once: the characters and qubits of all
    strings side by side in two flat arrays, one ``bitwise_or.reduceat`` per mask (13 300 strings of the N2 UCCSD generators:
    30 -> 9 ms against a Python loop over the characters)"""
    if nbqbits > 64:
        raise ValueError("at most 64 qubits fit the uint64 masks")
    T = len(terms)
    cs = np.fromiter((complex(t.coeff) for t in terms), np.complex128, T)
    xs = np.zeros(T, dtype=np.uint64)
    zs = np.zeros(T, dtype=np.uint64)
    lens = np.fromiter((len(t.op) for t in terms), np.int64, T)
    total = int(lens.sum())
    if total == 0:
        return xs, zs, cs
    qlens = np.fromiter((len(t.qbits) for t in terms), np.int64, T)
    if not np.array_equal(qlens, lens):   # (a flat array would shift every later term's qubits: checked before anything is flattened)
        k = int(np.argmax(qlens != lens))
        raise ValueError(f"term {k}: {int(lens[k])} Pauli characters on {int(qlens[k])} qubits")
    try:
        chars = np.frombuffer("".join([t.op for t in terms]).encode("latin-1"), dtype=np.uint8)
    except UnicodeEncodeError as exc:
        raise ValueError(f"unknown Pauli '{exc.object[exc.start]}'") from None
    qubits = np.fromiter((q for t in terms for q in t.qbits), np.int64, total)
    if qubits.min() < 0 or qubits.max() >= nbqbits:
        raise ValueError("qubit index out of range")
    is_x, is_y, is_z, is_i = chars == ord("X"), chars == ord("Y"), chars == ord("Z"), chars == ord("I")
    if not (is_x | is_y | is_z | is_i).all():
        bad = chr(int(chars[~(is_x | is_y | is_z | is_i)][0]))
        raise ValueError(f"unknown Pauli '{bad}'")
    bits = np.left_shift(np.uint64(1), (nbqbits - 1 - qubits).astype(np.uint64))
    zero = np.uint64(0)
    nonempty = lens > 0
    starts = (np.cumsum(lens) - lens)[nonempty]
    xs[nonempty] = np.bitwise_or.reduceat(np.where(is_x | is_y, bits, zero), starts)
    zs[nonempty] = np.bitwise_or.reduceat(np.where(is_y | is_z, bits, zero), starts)
    return xs, zs, cs
