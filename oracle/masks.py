"""TEST INFRASTRUCTURE ONLY — bit-mask CPU oracle (vectorised numpy).

Same mathematics as ``oracle/dense.py`` but O(2^n) per operation, so parity
tests can reach n ~ 24 in seconds.  It is validated against the dense oracle
in ``tests/test_oracle.py`` before anything is compared with it.

Conventions (reference lines cited in oracle/dense.py):
reference qubit q  <->  basis-index bit (n-1-q).  A Pauli string is packed as
two index-space masks (x, z): I=(0,0) X=(1,0) Z=(0,1) Y=(1,1), so that
P = i^{|x&z|} X^x Z^z and

    (P psi)_i = i^{ny} (-1)^{popcount((i^x) & z)} psi_{i^x},   ny = popcount(x&z).
"""
import numpy as np


def pack_pauli(nbqbits, op, qbits):
    x = 0
    z = 0
    for ch, q in zip(op, qbits):
        bit = 1 << (nbqbits - 1 - q)
        if ch == "X":
            x |= bit
        elif ch == "Y":
            x |= bit
            z |= bit
        elif ch == "Z":
            z |= bit
        elif ch != "I":
            raise ValueError(ch)
    return x, z


def _parity(v):
    v = v.copy()
    for s in (32, 16, 8, 4, 2, 1):
        v ^= v >> np.uint64(s)
    return (v & np.uint64(1)).astype(np.int64)


def _popcount_int(v):
    return bin(int(v)).count("1")


def pauli_apply(psi, x, z, index_offset=0):
    """Return P psi.  ``index_offset`` adds high (global) index bits for shard-local use
    (x must then be purely local)."""
    n = psi.shape[0]
    idx = np.arange(n, dtype=np.uint64) + np.uint64(index_offset)
    j = np.arange(n, dtype=np.int64) ^ int(x & (n - 1))
    sign = 1.0 - 2.0 * _parity((idx ^ np.uint64(x)) & np.uint64(z))
    phase = (1j) ** (_popcount_int(x & z) % 4)
    return phase * sign * psi[j]


def rotate(psi, x, z, phi):
    """exp(-i phi P) psi = cos(phi) psi - i sin(phi) P psi."""
    return np.cos(phi) * psi - 1j * np.sin(phi) * pauli_apply(psi, x, z)


def expectation(psi, xs, zs, coeffs, constant=0.0):
    """Re sum_t c_t <psi|P_t|psi> + constant."""
    total = 0.0 + 0.0j
    for x, z, c in zip(xs, zs, coeffs):
        total += complex(c) * np.vdot(psi, pauli_apply(psi, int(x), int(z)))
    return float(total.real) + float(np.real(constant))


def apply_pauli_sum(psi, xs, zs, coeffs):
    """sigma = sum_t c_t P_t psi (complex coefficients allowed)."""
    out = np.zeros_like(psi)
    for x, z, c in zip(xs, zs, coeffs):
        out += complex(c) * pauli_apply(psi, int(x), int(z))
    return out


def bilinear(phi, psi, x, z):
    """<phi|P|psi>."""
    return np.vdot(phi, pauli_apply(psi, x, z))


# ---------------------------------------------------------------- gate level
def gate_1q(psi, nbqbits, qubit, m):
    """Apply 2x2 matrix m on reference qubit ``qubit``."""
    bit = nbqbits - 1 - qubit
    v = psi.reshape(1 << (nbqbits - 1 - bit), 2, 1 << bit)
    out = np.empty_like(v)
    out[:, 0, :] = m[0, 0] * v[:, 0, :] + m[0, 1] * v[:, 1, :]
    out[:, 1, :] = m[1, 0] * v[:, 0, :] + m[1, 1] * v[:, 1, :]
    return out.reshape(-1)


def gate_cnot(psi, nbqbits, control, target):
    cb = 1 << (nbqbits - 1 - control)
    tb = 1 << (nbqbits - 1 - target)
    idx = np.arange(psi.shape[0], dtype=np.int64)
    src = np.where(idx & cb, idx ^ tb, idx)
    return psi[src]
