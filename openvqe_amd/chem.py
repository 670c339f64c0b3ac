"""Minimal ab-initio front-end for s-type Gaussian basis sets (hydrogen / helium: STO-3G, 6-31G):
one- and two-electron integrals, restricted Hartree-Fock, MO-basis integrals.

Purpose: the reference obtains molecular integrals from PySCF through myQLM
(``perform_pyscf_computation``, ref:openvqe/common_files/molecule_factory.py:306-322); neither exists on
the GPU box, so the molecule configs whose basis functions are all s-type — H2/6-31G (what
``main_ucc.py`` really runs), H2/STO-3G, H4/H6/STO-3G, HeH+ — are rebuilt here from the published basis-set
tables and the closed-form s-type Gaussian integrals (Boys function F0).  That is enough to replay the
reference's stored known answers K1, K3-K6 end to end.
"""
from __future__ import annotations

import math

import numpy as np

from . import fermion

BOHR = 0.52917721092  # Angstrom per Bohr (the value PySCF uses)

# exponents / contraction coefficients (normalised primitives), EMSL basis-set exchange tables
_STO3G_1S = [0.15432897, 0.53532814, 0.44463454]
_STO3G_2S = [-0.09996723, 0.39951283, 0.70011547]
_STO3G_2P = [0.15591627, 0.60768372, 0.39195739]
# shells with p functions: (l, exponents, coefficients); handled by the general McMurchie-Davidson code (gto.py)
BASIS_SP = {
    "sto-3g": {
        "H": [(0, [3.42525091, 0.62391373, 0.16885540], _STO3G_1S)],
        "Li": [(0, [16.1195750, 2.9362007, 0.7946505], _STO3G_1S),
               (0, [0.6362897, 0.1478601, 0.0480887], _STO3G_2S), (1, [0.6362897, 0.1478601, 0.0480887], _STO3G_2P)],
        "O": [(0, [130.7093200, 23.8088610, 6.4436083], _STO3G_1S),
              (0, [5.0331513, 1.1695961, 0.3803890], _STO3G_2S), (1, [5.0331513, 1.1695961, 0.3803890], _STO3G_2P)],
    },
}
BASIS = {
    "sto-3g": {
        "H": [([3.42525091, 0.62391373, 0.16885540], [0.15432897, 0.53532814, 0.44463454])],
        "He": [([6.36242139, 1.15892300, 0.31364979], [0.15432897, 0.53532814, 0.44463454])],
    },
    "6-31g": {
        "H": [([18.7311370, 2.8253937, 0.6401217], [0.03349460, 0.23472695, 0.81375733]),
              ([0.1612778], [1.0])],
        "He": [([38.4216340, 5.7780300, 1.2417740], [0.0237660, 0.1546790, 0.4696300]),
               ([0.2979640], [1.0])],
    },
}
CHARGE = {"H": 1, "He": 2, "Li": 3, "O": 8}


def _f0(t):
    t = np.asarray(t, dtype=float)
    small = t < 1e-12
    ts = np.where(small, 1.0, t)
    val = 0.5 * np.sqrt(np.pi / ts) * np.vectorize(math.erf)(np.sqrt(ts))
    return np.where(small, 1.0 - t / 3.0, val)


class Molecule:
    """geometry: [(symbol, (x, y, z) in Angstrom), ...]"""

    def __init__(self, geometry, basis="sto-3g", charge=0):
        self.atoms = [(sym, np.array(xyz, dtype=float) / BOHR) for sym, xyz in geometry]
        self.basis_name = basis.lower()
        self.charge = charge
        self.n_elec = sum(CHARGE[s] for s, _ in self.atoms) - charge
        self.general = any(sym not in BASIS.get(self.basis_name, {}) for sym, _ in self.atoms)
        if self.general:
            from . import gto
            self.functions = []
            for sym, pos in self.atoms:
                for l, exps, coefs in BASIS_SP[self.basis_name][sym]:
                    for lmn in ([(0, 0, 0)] if l == 0 else [(1, 0, 0), (0, 1, 0), (0, 0, 1)]):
                        self.functions.append(gto.BasisFunction(pos, lmn, exps, coefs))
            self.nao = len(self.functions)
            return
        # contracted s functions: (centre, exponents, coefficients incl. primitive normalisation)
        self.shells = []
        for sym, pos in self.atoms:
            for exps, coefs in BASIS[self.basis_name][sym]:
                a = np.array(exps)
                c = np.array(coefs) * (2.0 * a / np.pi) ** 0.75
                self.shells.append((pos, a, c))
        self.nao = len(self.shells)

    def nuclear_repulsion(self):
        e = 0.0
        for i, (si, ri) in enumerate(self.atoms):
            for sj, rj in self.atoms[i + 1:]:
                e += CHARGE[si] * CHARGE[sj] / np.linalg.norm(ri - rj)
        return e

    # -- integrals over contracted s functions ------------------------------------------------
    def one_electron(self):
        n = self.nao
        S = np.zeros((n, n))
        T = np.zeros((n, n))
        V = np.zeros((n, n))
        for i, (A, a, ca) in enumerate(self.shells):
            for j, (B, b, cb) in enumerate(self.shells):
                ab2 = float(np.dot(A - B, A - B))
                p = a[:, None] + b[None, :]
                mu = a[:, None] * b[None, :] / p
                pref = ca[:, None] * cb[None, :]
                s = (np.pi / p) ** 1.5 * np.exp(-mu * ab2)
                S[i, j] = np.sum(pref * s)
                T[i, j] = np.sum(pref * mu * (3.0 - 2.0 * mu * ab2) * s)
                P = (a[:, None, None] * A + b[None, :, None] * B) / p[:, :, None]
                for sym, C in self.atoms:
                    pc2 = np.sum((P - C) ** 2, axis=2)
                    V[i, j] += np.sum(pref * (-CHARGE[sym]) * (2.0 * np.pi / p) * np.exp(-mu * ab2) * _f0(p * pc2))
        return S, T, V

    def two_electron(self):
        """(ij|kl) in chemists' notation"""
        n = self.nao
        eri = np.zeros((n, n, n, n))
        pairs = {}
        for i, (A, a, ca) in enumerate(self.shells):
            for j, (B, b, cb) in enumerate(self.shells):
                p = (a[:, None] + b[None, :]).ravel()
                mu = (a[:, None] * b[None, :]).ravel() / p
                K = (ca[:, None] * cb[None, :]).ravel() * np.exp(-mu * float(np.dot(A - B, A - B)))
                P = ((a[:, None, None] * A + b[None, :, None] * B).reshape(-1, 3)) / p[:, None]
                pairs[(i, j)] = (p, K, P)
        for i in range(n):
            for j in range(i + 1):
                p, Kp, P = pairs[(i, j)]
                for k in range(n):
                    for l in range(k + 1):
                        if (i * (i + 1) // 2 + j) < (k * (k + 1) // 2 + l):
                            continue
                        q, Kq, Q = pairs[(k, l)]
                        pq2 = np.sum((P[:, None, :] - Q[None, :, :]) ** 2, axis=2)
                        pp, qq = p[:, None], q[None, :]
                        val = np.sum(Kp[:, None] * Kq[None, :] * 2.0 * np.pi ** 2.5 / (pp * qq * np.sqrt(pp + qq))
                                     * _f0(pp * qq / (pp + qq) * pq2))
                        for (w, x, y, z) in ((i, j, k, l), (j, i, k, l), (i, j, l, k), (j, i, l, k),
                                             (k, l, i, j), (l, k, i, j), (k, l, j, i), (l, k, j, i)):
                            eri[w, x, y, z] = val
        return eri

    # -- restricted Hartree-Fock ------------------------------------------------------------------
    def rhf(self, tol=1e-12, max_iter=200):
        if self.general:
            from . import gto
            S, T, V, eri = gto.integrals(self.functions, [(CHARGE[s], r) for s, r in self.atoms])
        else:
            S, T, V = self.one_electron()
            eri = self.two_electron()
        hcore = T + V
        nocc = self.n_elec // 2
        sval, svec = np.linalg.eigh(S)
        X = svec @ np.diag(sval ** -0.5) @ svec.T
        D = np.zeros_like(S)
        e_old = 0.0
        fock_hist, err_hist = [], []
        for it in range(max_iter):
            J = np.einsum("pqrs,rs->pq", eri, D)
            Kx = np.einsum("prqs,rs->pq", eri, D)
            F = hcore + 2.0 * J - Kx
            err = X.T @ (F @ D @ S - S @ D @ F) @ X
            fock_hist.append(F)
            err_hist.append(err)
            if len(fock_hist) > 8:
                fock_hist.pop(0)
                err_hist.pop(0)
            if len(fock_hist) > 1:  # DIIS
                m = len(fock_hist)
                Bm = -np.ones((m + 1, m + 1))
                Bm[m, m] = 0.0
                for a in range(m):
                    for b in range(m):
                        Bm[a, b] = np.sum(err_hist[a] * err_hist[b])
                rhs = np.zeros(m + 1)
                rhs[m] = -1.0
                try:
                    coef = np.linalg.solve(Bm, rhs)[:m]
                    F = sum(c * f for c, f in zip(coef, fock_hist))
                except np.linalg.LinAlgError:
                    pass
            eps, Cp = np.linalg.eigh(X.T @ F @ X)
            C = X @ Cp
            D = C[:, :nocc] @ C[:, :nocc].T
            J = np.einsum("pqrs,rs->pq", eri, D)
            Kx = np.einsum("prqs,rs->pq", eri, D)
            e = np.sum(D * (2.0 * hcore + 2.0 * J - Kx))
            if abs(e - e_old) < tol and np.abs(err).max() < 1e-9:
                break
            e_old = e
        self.mo_coeff, self.mo_energy = C, eps
        self.e_hf = e + self.nuclear_repulsion()
        self.h_mo = C.T @ hcore @ C
        self.eri_mo = np.einsum("pqrs,pi,qj,rk,sl->ijkl", eri, C, C, C, C, optimize=True)
        return self.e_hf

    # -- qubit objects ------------------------------------------------------------------------------
    def jw_hamiltonian(self):
        """JW molecular Hamiltonian in the HF molecular-orbital basis (spin orbitals interleaved),
        constant = nuclear repulsion — the object ``generate_hamiltonian`` hands to the hot path
        (ref:openvqe/common_files/molecule_factory.py:336-350)."""
        if not hasattr(self, "h_mo"):
            self.rhf()
        hpq, hpqrs = fermion.spin_orbital_integrals(self.h_mo, self.eri_mo)
        return fermion.jw_molecular_hamiltonian(hpq, hpqrs, self.nuclear_repulsion())

    def hf_init(self):
        return fermion.hf_integer(2 * self.nao, self.n_elec)


def molecule(symbol):
    """geometries / bases of the reference's table (ref:openvqe/common_files/molecule_factory.py:45-120,
    ref:openvqe/common_files/get_energy_WSSVQE.py:46-51) for the all-s-type cases"""
    s = symbol.upper()
    if s == "H2":
        return Molecule([("H", (0, 0, 0)), ("H", (0, 0, 0.75))], "6-31g")
    if s == "H2-STO3G-WSSVQE":
        return Molecule([("H", (0, 0, 0)), ("H", (0, 0, 0.98))], "sto-3g")
    if s == "H4":
        r = 0.85
        return Molecule([("H", (0, 0, k * r)) for k in range(4)], "sto-3g")
    if s == "H6":
        r = 1.0
        return Molecule([("H", (0, 0, k * r)) for k in range(6)], "sto-3g")
    if s == "LIH":
        return Molecule([("Li", (0, 0, 0)), ("H", (0, 0, 1.45))], "sto-3g")
    if s == "H2O":
        r, theta = 1.0285, 0.538 * np.pi
        return Molecule([("O", (0, 0, 0)), ("H", (0, 0, r)),
                         ("H", (0, r * np.sin(np.pi - theta), r * np.cos(np.pi - theta)))], "sto-3g")
    if s == "HEH+":
        return Molecule([("He", (0, 0, 0)), ("H", (0, 0, 1.0))], "6-31g", charge=1)
    raise KeyError(symbol)
