"""TEST INFRASTRUCTURE ONLY — the Clifford frame of a literal gate list, on the host, in tableau form.

The reference executes its QUCCSD templates gate by gate (ref:openvqe/common_files/circuit.py:13-106: X, H, CNOT, RZ(+-pi/2),
RY(+-pi/2) and the parametrised RY(+-theta), RY(-2 theta); conventions RY(a) = exp(-i a Y / 2), RZ(a) = exp(-i a Z / 2),
``apply(CNOT, control, target)`` — SURVEY.md section 8 a4).  Every gate of such a list is either a Clifford gate or a Pauli rotation,
so the circuit g_N ... g_2 g_1 equals

        C_N . R'_K ... R'_2 R'_1 ,        R'_k = exp(-i phi_k  C^+ P_k C)   with C the product of the Clifford gates BEFORE rotation k,

i.e. a sequence of Pauli rotations about conjugated strings followed by the net Clifford operator C_N (which every template of
circuit.py closes to a phase: `closed`).  This module restates that algebra independently of the product's frame compiler
(openvqe_amd/csrc/sv_frame_host.hpp): Python integers as bit masks, the map P -> C^+ P C kept as the images of X_q and Z_q and
updated per Clifford gate from the two or four rows the gate touches.  tests/test_oracle.py pins it against gate-by-gate dense
simulation; tests/test_gpu_fullsize.py feeds ITS rotation list of the un-thinned N2 QUCCSD circuit (62 852 gates -> 13 300
rotations) to the C oracle, so that the full-size parity of configs[3] no longer borrows the product's compiled sequence.

Pauli operators are triples (x, z, k) = i^k X^x Z^z with index-bit masks (qubit q <-> bit n - 1 - q); the
Hermitian string the backend and the C oracle mean by (x, z) is i^{|x & z|} X^x Z^z.
"""
import math

import numpy as np


def _mul(a, b):
    """(i^ka X^xa Z^za)(i^kb X^xb Z^zb): moving Z^za past X^xb costs (-1)^{|za & xb|}"""
    return a[0] ^ b[0], a[1] ^ b[1], (a[2] + b[2] + 2 * bin(a[1] & b[0]).count("1")) & 3


def _anticommute(a, b):
    return (bin(a[0] & b[1]).count("1") + bin(a[1] & b[0]).count("1")) & 1


class Frame:
    """the map P -> C^+ P C of the Clifford gates seen so far: rows[q] = image of X_q, rows[n + q] = image of Z_q"""

    def __init__(self, nbqbits):
        self.n = int(nbqbits)
        self.rows = [(self._bit(q), 0, 0) for q in range(self.n)] + [(0, self._bit(q), 0) for q in range(self.n)]

    def _bit(self, q):
        return 1 << (self.n - 1 - int(q))

    def image(self, pauli):
        """C^+ P C for P = (x, z, k) given on the circuit's qubits"""
        x, z, k = pauli
        out = (0, 0, k & 3)
        for q in range(self.n):          # X_q^{x_q} Z_q^{z_q} qubit by qubit, the order of the raw form
            b = self._bit(q)
            if x & b:
                out = _mul(out, self.rows[q])
            if z & b:
                out = _mul(out, self.rows[self.n + q])
        return out

    def _set(self, new):
        for idx, val in new.items():
            self.rows[idx] = val

    def append(self, name, qubits, quarter_turns=0):
        """C <- g C for a Clifford gate g: the rows of g's qubits become image(g^+ G g), computed from the OLD rows"""
        n = self.n
        q = int(qubits[0])
        b = self._bit(q)
        X, Z = (b, 0, 0), (0, b, 0)
        if name == "H":                  # H X H = Z, H Z H = X
            self._set({q: self.image(Z), n + q: self.image(X)})
        elif name == "X":                # X Z X = -Z
            self._set({n + q: self.image((0, b, 2))})
        elif name == "CNOT":             # X_c -> X_c X_t, Z_t -> Z_c Z_t (CNOT is its own inverse)
            t = int(qubits[1])
            bt = self._bit(t)
            self._set({q: self.image((b | bt, 0, 0)), n + t: self.image((0, b | bt, 0))})
        elif name in ("RX", "RY", "RZ"):
            # g = exp(-i m pi/4 Q): g^+ P g = P for [P, Q] = 0, else P (cos a - i sin a Q) with a = m pi/2:
            # m = 1: -i P Q,  m = 2: -P,  m = 3: +i P Q
            m = quarter_turns & 3
            if m == 0:
                return
            Q = {"RX": (b, 0, 0), "RY": (b, b, 1), "RZ": (0, b, 0)}[name]
            new = {}
            for idx, G in ((q, X), (n + q, Z)):
                if not _anticommute(G, Q):
                    continue
                if m == 2:
                    P = (G[0], G[1], (G[2] + 2) & 3)
                else:
                    P = _mul(G, Q)
                    P = (P[0], P[1], (P[2] + (3 if m == 1 else 1)) & 3)
                new[idx] = self.image(P)
            self._set(new)
        else:
            raise ValueError(f"not a Clifford gate: {name}")

    def is_identity(self):
        n = self.n
        return all(self.rows[q] == (self._bit(q), 0, 0) and self.rows[n + q] == (0, self._bit(q), 0) for q in range(n))


def _quarter_turns(angle):
    """m with angle == m pi/2 bitwise (as the templates write their basis changes), else None"""
    m = angle / (0.5 * math.pi)
    return int(round(m)) if m == round(m) else None


def rotation_sequence(nbqbits, gates):
    """gates: [(name, qubits, angle_scale, angle_const, param_index or -1)] (the input of Statevector.set_gate_program) ->
    (xs, zs, coeffs, phi0s, pidx, closed): rotation r is exp(-i (coeffs[r] theta[pidx[r]] + phi0s[r]) P_r) with the Hermitian
    string P_r = i^{|x & z|} X^x Z^z (signs folded into coeffs / phi0s), in circuit order; ``closed``: the net Clifford operator
    behind the rotations is the identity up to a phase"""
    fr = Frame(nbqbits)
    xs, zs, cs, p0, pi = [], [], [], [], []
    for name, qubits, scale, const, p in gates:
        if name in ("X", "H", "CNOT"):
            fr.append(name, qubits)
            continue
        if name not in ("RX", "RY", "RZ"):
            raise ValueError(name)
        m = _quarter_turns(const) if p < 0 else None
        if p < 0 and m is not None:
            fr.append(name, qubits, m)
            continue
        b = fr._bit(qubits[0])
        Q = {"RX": (b, 0, 0), "RY": (b, b, 1), "RZ": (0, b, 0)}[name]
        x, z, k = fr.image(Q)
        rel = (k - bin(x & z).count("1")) & 3          # i^rel relative to the Hermitian string of (x, z): +1 or -1
        if rel & 1:
            raise AssertionError("conjugated Pauli is not Hermitian")
        sign = -1.0 if rel else 1.0
        xs.append(x)
        zs.append(z)
        cs.append(sign * 0.5 * (scale if p >= 0 else 0.0))     # RY(a) = exp(-i (a / 2) Y)
        p0.append(sign * 0.5 * const)
        pi.append(int(p))
    return (np.array(xs, np.uint64), np.array(zs, np.uint64), np.array(cs, np.float64), np.array(p0, np.float64),
            np.array(pi, np.int32), fr.is_identity())
