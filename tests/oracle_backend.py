"""TEST INFRASTRUCTURE ONLY — a CPU stand-in with the method surface of openvqe_amd.backend.Statevector,
built on the bit-mask oracle.  CPU tests monkeypatch it in to exercise the HOST logic (qat stand-ins,
L1 mirrors, ADAPT loops, distributed planner) where no GPU exists; it is never importable from the product."""
import numpy as np

from openvqe_amd.backend import GATE_OPCODES, compile_ucc_program
from openvqe_amd.operators import pack_string, pack_terms
from oracle import dense, masks


class OracleStatevector:
    def __init__(self, n_qubits, device=0, n_global=0, shard_index=0):
        assert n_global == 0
        self.nbqbits = self.n_local = int(n_qubits)
        self.n_global = 0
        self.psi = np.zeros(1 << self.nbqbits, complex)
        self.psi[0] = 1
        self._ham = None
        self._prog = None
        self._K = 0

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def set_option(self, *a):
        pass

    # state
    def init_basis(self, index):
        self.psi[:] = 0
        self.psi[int(index)] = 1

    def set_state(self, psi):
        self.psi = np.array(psi, complex)

    def get_state(self):
        return self.psi.copy()

    def norm2(self):
        return float(np.vdot(self.psi, self.psi).real)

    # unit ops
    def apply_pauli_rotation(self, x, z, phi):
        self.psi = masks.rotate(self.psi, int(x), int(z), float(phi))

    def apply_pauli_rotations(self, xs, zs, phis):
        for x, z, p in zip(xs, zs, phis):
            self.apply_pauli_rotation(x, z, p)

    def rotate(self, op, qbits, phi):
        self.apply_pauli_rotation(*pack_string(self.nbqbits, op, qbits), phi)

    def apply_gate(self, name, qubits, angle=0.0):
        n = self.nbqbits
        if name == "CNOT":
            self.psi = masks.gate_cnot(self.psi, n, qubits[0], qubits[1])
        else:
            self.psi = masks.gate_1q(self.psi, n, qubits[0], dense.gate_matrix(name, angle))

    def expectation(self, hamiltonian):
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        return masks.expectation(self.psi, xs, zs, cs.real, complex(hamiltonian.constant_coeff or 0).real)

    # compiled
    def set_hamiltonian(self, hamiltonian):
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        assert np.abs(cs.imag).max(initial=0) < 1e-12
        self._ham = (xs, zs, cs.real, complex(hamiltonian.constant_coeff or 0).real)

    def energy_gradient(self, theta):
        """checker of ovqe_energy_gradient: central differences of the oracle energy"""
        theta = np.asarray(theta, float)
        g = np.zeros(len(theta))
        for k in range(len(theta)):
            tp, tm = theta.copy(), theta.copy()
            tp[k] += 1e-5; tm[k] -= 1e-5
            g[k] = (self.energy(tp) - self.energy(tm)) / 2e-5
        return self.energy(theta), g

    def ground_state(self, tol=1e-10, max_iter=3000, seed=0):
        """checker of ovqe_ground_state: matrix-free scipy Lanczos (ARPACK) on the bit-mask oracle's H|v>"""
        import scipy.sparse.linalg as sla
        xs, zs, cs, const = self._ham
        op = sla.LinearOperator((self.psi.size,) * 2, dtype=complex,
                                matvec=lambda v: masks.apply_pauli_sum(np.asarray(v, complex).ravel(), xs, zs, cs))
        vals, vecs = sla.eigsh(op, k=1, which="SA", tol=tol)
        self.psi = vecs[:, 0] / np.linalg.norm(vecs[:, 0])
        res = np.linalg.norm(masks.apply_pauli_sum(self.psi, xs, zs, cs) - vals[0] * self.psi)
        return float(vals[0] + const), float(res), 0

    def set_rotation_program(self, xs, zs, coeffs, pidx, n_params, hf_init, phi0=None):
        phi0 = np.zeros(len(xs)) if phi0 is None else np.asarray(phi0)
        self._prog = ("rot", np.asarray(xs), np.asarray(zs), np.asarray(coeffs), phi0, np.asarray(pidx), int(hf_init))
        self._K = int(n_params)

    def set_ucc_program(self, generators, hf_init, n_params=None):
        xs, zs, cs, ps, K = compile_ucc_program(self.nbqbits, generators, n_params)
        self.set_rotation_program(xs, zs, cs, ps, K, hf_init)
        return K

    def set_gate_program(self, gates, n_params, hf_init):
        for g in gates:
            assert g[0] in GATE_OPCODES
        self._prog = ("gates", list(gates), int(hf_init))
        self._K = int(n_params)

    def prepare_state(self, theta):
        theta = np.asarray(theta, float).reshape(-1)
        if self._prog[0] == "rot":
            _, xs, zs, cs, p0, pi, hf = self._prog
            self.init_basis(hf)
            for x, z, c, c0, p in zip(xs, zs, cs, p0, pi):
                self.apply_pauli_rotation(x, z, c0 + (c * theta[p] if p >= 0 else 0.0))
        else:
            _, gates, hf = self._prog
            self.init_basis(hf)
            for name, qubits, sc, co, p in gates:
                self.apply_gate(name, qubits, co + (sc * theta[p] if p >= 0 else 0.0))

    #: True: Pauli-rotation programs are evaluated by the C restatement of the same oracle (oracle/c/ovqe_oracle.c, fused mask sweeps —
    #: what tests/test_oracle.py pins against this class's numpy form) instead of the numpy loops: ~100 x faster at 14 qubits, for the
    #: flows that evaluate hundreds of energies (tests/test_gpu_flows.py, H2O ADAPT)
    use_c_oracle = False

    def energy(self, theta):
        if self.use_c_oracle and self._prog[0] == "rot" and not np.any(self._prog[4]):
            from oracle import cref
            _, xs, zs, cs, _, pi, hf = self._prog
            hx, hz, hc, const = self._ham
            e, psi = cref.ucc_energy(self.nbqbits, hf, xs, zs, cs, np.asarray(pi, np.int32), np.asarray(theta, float).reshape(-1),
                                     hx, hz, hc, const, 0)
            self.psi = psi
            return float(e)
        self.prepare_state(theta)
        xs, zs, cs, const = self._ham
        return masks.expectation(self.psi, xs, zs, cs, const)

    def energy_batch(self, thetas):
        return np.array([self.energy(t) for t in np.asarray(thetas, float)])

    # ADAPT
    def pool_gradients(self, pool_ops, mode):
        xs, zs, cs, const = self._ham
        sig = masks.apply_pauli_sum(self.psi, xs, zs, cs) + const * self.psi
        out = []
        for op in pool_ops:
            px, pz, pc = pack_terms(self.nbqbits, op.terms)
            val = np.vdot(sig, masks.apply_pauli_sum(self.psi, px, pz, pc))
            out.append(2 * val.real if mode == 0 else 2 * abs(val))
        return np.array(out)

    def apply_exp_pauli_sum(self, operator, theta, prefactor=1.0):
        mat = dense.operator_matrix(operator, sparse=True, with_constant=False) * prefactor
        self.psi = dense.exact_exp_state(self.psi, [mat], [theta])
