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
        "N": [(0, [99.1061690, 18.0523120, 4.8856602], _STO3G_1S),
              (0, [3.7804559, 0.8784966, 0.2857144], _STO3G_2S), (1, [3.7804559, 0.8784966, 0.2857144], _STO3G_2P)],
    },
    # Dunning's correlation-consistent polarised valence double zeta, (9s4p1d) -> [3s2p1d], spherical d functions
    "cc-pvdz": {
        "N": [(0, [9046.0, 1357.0, 309.3, 87.73, 28.56, 10.21, 3.838, 0.7466],
               [0.000700, 0.005389, 0.027406, 0.103207, 0.278723, 0.448540, 0.278238, 0.015440]),
              (0, [9046.0, 1357.0, 309.3, 87.73, 28.56, 10.21, 3.838, 0.7466],
               [-0.000153, -0.001208, -0.005992, -0.024544, -0.067459, -0.158078, -0.121831, 0.549003]),
              (0, [0.2248], [1.0]),
              (1, [13.55, 2.917, 0.7973], [0.039919, 0.217169, 0.510319]),
              (1, [0.2185], [1.0]),
              (2, [0.817], [1.0])],
        "H": [(0, [13.01, 1.962, 0.4446], [0.019685, 0.137977, 0.478148]), (0, [0.122], [1.0]), (1, [0.727], [1.0])],
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
CHARGE = {"H": 1, "He": 2, "Li": 3, "N": 7, "O": 8}


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
            cart = {0: [(0, 0, 0)], 1: [(1, 0, 0), (0, 1, 0), (0, 0, 1)],
                    2: [(2, 0, 0), (0, 2, 0), (0, 0, 2), (1, 1, 0), (1, 0, 1), (0, 1, 1)]}
            self.functions = []
            blocks = []          # Cartesian -> spherical transformation, block by block (identity for s and p)
            for sym, pos in self.atoms:
                for l, exps, coefs in BASIS_SP[self.basis_name][sym]:
                    for lmn in cart[l]:
                        self.functions.append(gto.BasisFunction(pos, lmn, exps, coefs))
                    blocks.append(gto.spherical_d_transform(cart[2]) if l == 2 else np.eye(len(cart[l])))
            nc, ns = sum(b.shape[0] for b in blocks), sum(b.shape[1] for b in blocks)
            self.cart2sph = np.zeros((nc, ns))
            r = c = 0
            for b in blocks:
                self.cart2sph[r:r + b.shape[0], c:c + b.shape[1]] = b
                r += b.shape[0]
                c += b.shape[1]
            self.has_d = nc != ns
            self.nao = ns
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
    def ao_integrals(self):
        """S, T, V, (ij|kl) over the molecule's atomic orbitals (s-only closed forms; McMurchie-Davidson for p and d shells —
        compiled when libovqe_gto.so is built, which d shells require; d functions as the five spherical components)"""
        if not self.general:
            S, T, V = self.one_electron()
            return S, T, V, self.two_electron()
        from . import gto
        charges = [(CHARGE[s], r) for s, r in self.atoms]
        if gto._clib() is not None:
            S, T, V, eri = gto.integrals_compiled(self.functions, charges)
        elif self.has_d:
            raise RuntimeError("d shells need the compiled integral code: run __graft_entry__.build()")
        else:
            S, T, V, eri = gto.integrals(self.functions, charges)
        if self.has_d:
            U = self.cart2sph
            S, T, V = U.T @ S @ U, U.T @ T @ U, U.T @ V @ U
            eri = np.einsum("pqrs,pi,qj,rk,sl->ijkl", eri, U, U, U, U, optimize=True)
        return S, T, V, eri

    def _atomic_guess_density(self):
        """diagonal AO density (in units of electron PAIRS, like C_occ C_occ^T) from the atoms' ground configurations:
        s electrons fill the atom's s shells in listing order, p electrons its first p shell evenly; zero (= the
        core-Hamiltonian guess) for the s-only molecules, which have no competing SCF solutions"""
        D = np.zeros((self.nao, self.nao))
        if not self.general:
            return D
        config = {"H": (1, 0), "He": (2, 0), "Li": (3, 0), "N": (4, 3), "O": (4, 4)}    # (s electrons, p electrons)
        k = 0
        for sym, _ in self.atoms:
            ns, np_ = config[sym]
            first_p = True
            for l, _, _ in BASIS_SP[self.basis_name][sym]:
                width = {0: 1, 1: 3, 2: 5}[l]
                if l == 0:
                    occ = min(2, ns)
                    ns -= occ
                    D[k, k] = occ / 2.0
                elif l == 1 and first_p:
                    for c in range(3):
                        D[k + c, k + c] = np_ / 6.0
                    first_p = False
                k += width
        return D

    def rhf(self, tol=1e-12, max_iter=200, grad_tol=1e-9):
        """restricted Hartree-Fock with DIIS.  Converged when |dE| < tol and the orbital gradient (max element of the
        orthonormalised commutator FDS - SDF) < grad_tol.  PySCF's defaults — what the reference's myQLM front-end runs
        with — are conv_tol = 1e-9 on the energy and sqrt(conv_tol) = 3.2e-5 on the gradient norm: ``rhf(tol=1e-9,
        grad_tol=3.2e-5)`` stops there and leaves orbitals that are rotated by ~1e-7 .. 1e-6 against the converged ones
        (tests/test_scf_threshold.py: that rotation IS the 1e-8-level offset between the stored notebook numbers and a
        tightly converged replay)."""
        S, T, V, eri = self.ao_integrals()
        hcore = T + V
        nocc = self.n_elec // 2
        sval, svec = np.linalg.eigh(S)
        X = svec @ np.diag(sval ** -0.5) @ svec.T
        # start from a superposition of atomic ground-state occupations (the role of PySCF's default 'minao' guess): the bare
        # core-Hamiltonian guess locks N2 onto an excited closed-shell SCF solution 0.73 Ha above the ground state
        D = self._atomic_guess_density()
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
            if abs(e - e_old) < tol and np.abs(err).max() < grad_tol:
                break
            e_old = e
        # orbital phase gauge of PySCF's SCF eigensolver (pyscf.scf.hf.eig): the AO coefficient of largest magnitude of
        # every MO is positive (first such coefficient on ties, which symmetric molecules have).  Energies of UCC-type
        # circuits at FIXED parameters depend on this gauge (theta_k -> -theta_k under a sign flip of an orbital that
        # excitation k touches an odd number of times), so replaying the reference's stored evaluations needs it.
        for k in range(C.shape[1]):
            col = np.abs(C[:, k])
            lead = int(np.argmax(col >= col.max() * (1.0 - 1e-9)))
            if C[lead, k] < 0:
                C[:, k] = -C[:, k]
        self.mo_coeff, self.mo_energy = C, eps
        self.e_hf = e + self.nuclear_repulsion()
        self.h_mo = C.T @ hcore @ C
        self.eri_mo = np.einsum("pqrs,pi,qj,rk,sl->ijkl", eri, C, C, C, C, optimize=True)
        return self.e_hf

    def rotate_orbitals(self, kappa):
        """C <- C exp(K), K the antisymmetric matrix with K[p, q] = kappa[(p, q)] = -K[q, p]; MO integrals are rebuilt,
        orbital energies kept.  Models a not-fully-converged SCF (orbitals rotated against the stationary ones)."""
        import scipy.linalg
        K = np.zeros((self.nao, self.nao))
        for (p, q), v in kappa.items():
            K[p, q] += v
            K[q, p] -= v
        self._set_orbitals(self.mo_coeff @ scipy.linalg.expm(K))

    def _set_orbitals(self, C):
        _, T, V, eri = self.ao_integrals()
        self.mo_coeff = C
        self.h_mo = C.T @ (T + V) @ C
        self.eri_mo = np.einsum("pqrs,pi,qj,rk,sl->ijkl", eri, C, C, C, C, optimize=True)

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

    # -- correlated quantities the reference prints / selects with --------------------------------------------------
    def mp2_energy(self):
        """closed-shell MP2 total energy (``info['MP2']``, ref:openvqe/common_files/molecule_factory.py:318-322)"""
        if not hasattr(self, "h_mo"):
            self.rhf()
        o, e, g = self.n_elec // 2, self.mo_energy, self.eri_mo
        ov = g[:o, o:, :o, o:]                                   # (ia|jb)
        den = e[:o, None, None, None] - e[None, o:, None, None] + e[None, None, :o, None] - e[None, None, None, o:]
        return self.e_hf + float(np.sum(ov * (2.0 * ov - ov.transpose(0, 3, 2, 1)) / den))

    def ci_ground_state(self, max_excitation=None):
        """lowest eigenpair of the JW Hamiltonian in the N-electron determinant space, optionally restricted to
        determinants at most ``max_excitation``-fold excited from the HF determinant (2 = CISD; None = FCI).
        -> (energy, {basis index: amplitude}).  Determinant space built from the 2^n x 2^n sparse matrix: registers up to
        ~16 qubits, which is where the reference's own PySCF FCI / CISD step is used."""
        import scipy.sparse.linalg
        ham = self.jw_hamiltonian()
        n = ham.nbqbits
        if n > 18:
            raise ValueError("determinant-space CI restated for registers up to 18 qubits")
        hf = self.hf_init()
        idx = np.array([i for i in range(1 << n) if bin(i).count("1") == self.n_elec and
                        (max_excitation is None or bin(i & ~hf).count("1") <= max_excitation)], dtype=np.int64)
        sub = ham.get_matrix(sparse=True)[idx][:, idx].real
        if len(idx) <= 600:
            w, v = np.linalg.eigh(sub.toarray())
            vec = v[:, 0]
        else:
            w, v = scipy.sparse.linalg.eigsh(sub, k=1, which="SA", tol=1e-12)
            vec = v[:, 0]
        return float(w[0]), dict(zip(idx.tolist(), vec.tolist()))

    def natural_occupations(self, max_excitation=2):
        """(noons descending, natural orbitals as columns in the MO basis) of the spin-summed one-particle density of the
        CISD wave function — the ``rdm1`` of myQLM's ``perform_pyscf_computation`` (H4/STO-3G: the reference prints
        Noons = [1.98158247, 1.94333400, 0.06054808, 0.01453545], ref:notebooks/demo_quccsd.ipynb; CISD reproduces
        them to 3e-7, FCI differs at 1e-3), diagonalised as ref:openvqe/common_files/molecule_factory.py:367-373 does
        (eigh, both reversed)."""
        _, psi = self.ci_ground_state(max_excitation)
        n = 2 * self.nao
        rdm = np.zeros((self.nao, self.nao))
        for det, amp in psi.items():
            occ = [q for q in range(n) if (det >> (n - 1 - q)) & 1]
            for q in occ:                                  # annihilate q, create p of the same spin
                rest = det & ~(1 << (n - 1 - q))
                for p in range(q % 2, n, 2):
                    if (rest >> (n - 1 - p)) & 1:
                        continue
                    new = rest | (1 << (n - 1 - p))
                    if new not in psi:
                        continue
                    lo, hi = min(p, q), max(p, q)
                    between = sum((rest >> (n - 1 - r)) & 1 for r in range(lo + 1, hi))
                    rdm[p // 2, q // 2] += (-1) ** between * psi[new] * amp
        w, v = np.linalg.eigh(rdm)
        return w[::-1].copy(), v[:, ::-1].copy()

    def problem(self, active=False):
        """The objects ``MoleculeFactory.generate_hamiltonian(symbol, active)`` returns
        (ref:openvqe/common_files/molecule_factory.py:306-434) as one ``Problem``: full space in the HF orbitals, or the
        NOON-selected active space in the natural orbitals."""
        if not hasattr(self, "h_mo"):
            self.rhf()
        noons, natorb = self.natural_occupations()
        if not active:
            return Problem(self.h_mo, self.eri_mo, self.nuclear_repulsion(), self.n_elec, list(noons), list(self.mo_energy))
        h = natorb.T @ self.h_mo @ natorb
        g = np.einsum("pqrs,pi,qj,rk,sl->ijkl", self.eri_mo, natorb, natorb, natorb, natorb, optimize=True)
        eps1 = 2.0 - noons[0]                                    # molecule_factory.py:378-383
        eps2 = 0.01 if len(noons) < 3 else noons[3]
        frozen, act = select_active_orbitals(noons, self.n_elec, eps1, eps2)
        const = self.nuclear_repulsion()
        for i in frozen:
            const += 2.0 * h[i, i]
            for j in frozen:
                const += 2.0 * g[i, i, j, j] - g[i, j, j, i]
        h_act = h[np.ix_(act, act)].copy()
        for i in frozen:
            h_act += 2.0 * g[np.ix_(act, act, [i], [i])][:, :, 0, 0] - g[np.ix_(act, [i], [i], act)][:, 0, 0, :]
        g_act = g[np.ix_(act, act, act, act)]
        return Problem(h_act, g_act, const, self.n_elec - 2 * len(frozen), [noons[i] for i in act],
                       [self.mo_energy[i] for i in act], thresholds=(eps1, eps2), frozen=frozen, active=act)


def cas_problem(mol, n_frozen, n_active):
    """user-chosen complete active space in the HF orbitals: the lowest ``n_frozen`` orbitals stay doubly occupied (folded
    into the constant and the one-body part), the next ``n_active`` orbitals are kept — e.g. N2/cc-pVDZ (10 electrons, 12
    orbitals) = 24 qubits, the size of BASELINE.json configs[3].  (The reference selects active spaces by NOON thresholds,
    ``Molecule.problem(active=True)``; its CISD step is out of reach at 56 qubits, hence the explicit choice here.)"""
    if not hasattr(mol, "h_mo"):
        mol.rhf()
    frozen = list(range(n_frozen))
    act = list(range(n_frozen, n_frozen + n_active))
    h, g = mol.h_mo, mol.eri_mo
    const = mol.nuclear_repulsion()
    for i in frozen:
        const += 2.0 * h[i, i]
        for j in frozen:
            const += 2.0 * g[i, i, j, j] - g[i, j, j, i]
    h_act = h[np.ix_(act, act)].copy()
    for i in frozen:
        h_act += 2.0 * g[np.ix_(act, act, [i], [i])][:, :, 0, 0] - g[np.ix_(act, [i], [i], act)][:, 0, 0, :]
    n_act_el = mol.n_elec - 2 * n_frozen
    occ = [2.0 if k < n_act_el // 2 else 0.0 for k in range(n_active)]
    return Problem(h_act, g[np.ix_(act, act, act, act)], const, n_act_el, occ, [mol.mo_energy[i] for i in act],
                   frozen=frozen, active=act)


def select_active_orbitals(noons, n_elec, threshold_1, threshold_2):
    """NOON-based selection of myQLM's ``get_active_space_hamiltonian`` (called at
    ref:openvqe/common_files/molecule_factory.py:384-392; published rule): active A = {i : e2 <= n_i < 2 - e1} +
    {i : n_i >= 2 - e1 and 2(i+1) >= N_e}; frozen doubly occupied O = {i : n_i >= 2 - e1 and 2(i+1) < N_e}; the rest
    is discarded.  Pins: H4/STO-3G -> 6 qubits (pool sizes 8 / 18 / 69, ref:tests/test_main_*_active_space.py:15,
    ref:tests/test_main_fermionic_adapt.py:15), H2/6-31G -> all 8 qubits (ref:notebooks/demo_fermionic_adapt.ipynb)."""
    frozen, active = [], []
    for i, ni in enumerate(noons):
        if ni >= 2.0 - threshold_1:
            (active if 2 * (i + 1) >= n_elec else frozen).append(i)
        elif ni >= threshold_2:
            active.append(i)
    return frozen, active


class Problem:
    """spatial-orbital integrals + electron count of one (full or active-space) electronic-structure problem and the
    hot-path inputs derived from them"""

    def __init__(self, h, eri, constant, n_elec, noons, orbital_energies, thresholds=None, frozen=(), active=None):
        self.h, self.eri, self.constant, self.n_elec = np.asarray(h), np.asarray(eri), float(constant), int(n_elec)
        self.noons_full = [v for n in noons for v in (n, n)]
        self.orb_energies_full = [v for e in orbital_energies for v in (e, e)]
        self.thresholds, self.frozen = thresholds, list(frozen)
        self.active = list(range(len(noons))) if active is None else list(active)
        self.nbqbits = 2 * self.h.shape[0]
        self.hpq, self.hpqrs = fermion.spin_orbital_integrals(self.h, self.eri)

    def jw_hamiltonian(self):
        return fermion.jw_molecular_hamiltonian(self.hpq, self.hpqrs, self.constant)

    def spin_hamiltonian(self, transform="JW"):
        """"JW" | "Bravyi-Kitaev" | "parity_basis" (ref:openvqe/common_files/molecule_factory.py:349-356)"""
        return fermion.jw_molecular_hamiltonian(self.hpq, self.hpqrs, self.constant, transform=transform)

    def hf_init(self, transform="JW"):
        """HF determinant as a basis index of the encoded register (``recode_integer(hf_init, get_*_code(n))``,
        ref:…molecule_factory.py:478-486)"""
        return fermion.recode_occupation(fermion.hf_integer(self.nbqbits, self.n_elec), self.nbqbits, transform)

    def uccsd(self, transform="JW"):
        """(pool_size, cluster_ops, cluster_ops_sp, theta_MP2, hf_init) of ``MoleculeFactory.calculate_uccsd`` /
        ``uccsd`` (ref:openvqe/common_files/molecule_factory.py:436-470, ref:…generator_excitations.py:40-80)"""
        from .fermionic import spin_operator
        ops, theta, hf = fermion.cluster_ops_and_mp2_guess(self.n_elec, self.orb_energies_full, self.hpqrs)
        return len(ops), ops, [spin_operator(o, transform) for o in ops], theta, hf


    def fci_energy(self, tol=1e-10, device=0):
        """full-CI energy of this (active-space) problem in the Hartree-Fock determinant's (N_alpha, N_beta) sector — the
        ``info['FCI']`` the reference's drivers take from PySCF (ref:openvqe/common_files/molecule_factory.py:120-125) — for
        active spaces beyond the determinant-space CI of ``Molecule.ci_ground_state``: Lanczos on the device, on the
        materialised Hamiltonian of the determinant's sector (``ovqe_sector_ground_state`` on the closure of |hf> under the
        Hamiltonian; 12 qubits and more — when that is declined, on the sector tables of the problem's UCCSD program)"""
        from ._lib import BackendError
        from .backend import Statevector
        ham = self.jw_hamiltonian()
        _, _, generators, _, hf = self.uccsd()
        with Statevector(ham.nbqbits, device=device) as sv:
            sv.set_hamiltonian(ham)
            try:
                sv.init_basis(hf)
                return sv.sector_ground_state(tol=tol)[0]
            except BackendError:
                sv.set_ucc_program(generators, hf)
                return sv.sector_ground_state(tol=tol)[0]


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
    if s == "N2":                      # ref:openvqe/common_files/molecule_factory.py:245-250
        return Molecule([("N", (0, 0, 0.5488)), ("N", (0, 0, -0.5488))], "sto-3g")
    if s == "N2-CCPVDZ":               # the basis BASELINE.json configs[3] names (not in the reference's table)
        return Molecule([("N", (0, 0, 0.5488)), ("N", (0, 0, -0.5488))], "cc-pvdz")
    if s == "HEH+":
        return Molecule([("He", (0, 0, 0)), ("H", (0, 0, 1.0))], "6-31g", charge=1)
    raise KeyError(symbol)
