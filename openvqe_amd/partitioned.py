"""The ``Statevector`` face (what the L1 mirrors and the ``qat`` stand-ins call: ``backend.Statevector``) on the index-bit-partitioned
register of ``distributed.ShardedStatevector`` — so that ``EnergyUCC.ucc_action`` / ``get_energies``, ``fermionic_adapt_vqe`` and
``qubit_adapt_vqe`` (ref:openvqe/ucc_family/get_energy_ucc.py:8-50, ref:openvqe/adapt/fermionic_adapt_vqe.py:77-122,371-593,
ref:openvqe/adapt/qubit_adapt_vqe.py:310-605) run unchanged on a register that no single device holds.  One process per GPU, every
rank calls the same methods with the same arguments and gets the same numbers back (the collectives are inside).

What maps to what:
  ``set_hamiltonian`` + ``set_ucc_program`` / ``set_rotation_program``  ->  ``compile_program`` (exchange plan + cross-shard <H> plan, once)
  ``energy(theta)``                                                      ->  ``program_energy``
  ``init_basis`` / ``apply_exp_pauli_sum``  (the ADAPT screen state)     ->  rotations when the operator's strings commute (every JW single /
                                                                             double excitation, every pool string), else a Taylor series of sigma = A psi
  ``pool_gradients``                                                     ->  ``ShardedStatevector.pool_gradients``
Not offered on the partitioned register (each raises with a plain message): literal gate programs (the QUCCSD templates stop at 24
qubits in every config), the adjoint gradient, device Lanczos.
"""
from __future__ import annotations

import numpy as np

from .backend import GRAD_FERMIONIC, _real_coeff, compile_ucc_program
from .operators import pack_terms

#: tests: a callable (n_local, n_global, rank) -> shard engine that replaces the HIP engine (CPU runs of the N > 1 logic)
ENGINE_FACTORY = None

#: ``get_state`` gathers the whole register on every rank: refused above this size
GATHER_MAX_QUBITS = 28


class PartitionedStatevector:
    def __init__(self, n_qubits, device=None):
        from .distributed import ShardedStatevector
        self.nbqbits = int(n_qubits)
        self.sharded = ShardedStatevector(self.nbqbits, engine_factory=ENGINE_FACTORY, device=device)
        self.n_local = self.sharded.n_local
        self._ham = None
        self._ham_version = 0
        self._rot = None
        self._prog = None
        self._K = 0
        self._pool_cache = None

    # -- lifecycle ------------------------------------------------------------------------------------------------------
    def close(self):
        self._drop_program()
        self.sharded = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name, value):     # tuning knobs of the one-GPU handle: nothing to tune here
        pass

    def _drop_program(self):
        if self._prog is not None and self._prog.get("ham") is not None and self.sharded is not None:
            self.sharded.free_plan(self._prog["ham"])
        self._prog = None

    # -- compiled evaluation --------------------------------------------------------------------------------------------
    def set_hamiltonian(self, hamiltonian):
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        coeff = np.array([_real_coeff(c, "observable") for c in cs], np.float64)
        const = _real_coeff(getattr(hamiltonian, "constant_coeff", 0.0) or 0.0, "observable constant")
        self._ham = (xs, zs, coeff, const)
        self._ham_version += 1
        self._drop_program()

    def set_rotation_program(self, xs, zs, coeffs, pidx, n_params, hf_init, phi0=None):
        self._rot = (np.ascontiguousarray(xs, np.uint64), np.ascontiguousarray(zs, np.uint64), np.ascontiguousarray(coeffs, np.float64),
                     np.ascontiguousarray(pidx, np.int64), None if phi0 is None else np.ascontiguousarray(phi0, np.float64), int(hf_init))
        self._K = int(n_params)
        self._drop_program()

    def set_ucc_program(self, generators, hf_init, n_params=None):
        xs, zs, cs, ps, K = compile_ucc_program(self.nbqbits, generators, n_params)
        self.set_rotation_program(xs, zs, cs, ps, K, hf_init)
        return K

    def set_gate_program(self, gates, n_params, hf_init):
        raise NotImplementedError("literal gate programs are not offered on the partitioned register (the QUCCSD templates of "
                                  "ref:openvqe/common_files/circuit.py are used up to 24 qubits: one device)")

    def _program(self):
        if self._rot is None:
            raise RuntimeError("no program set")
        if self._prog is None:
            xs, zs, cs, ps, p0, hf = self._rot
            self._prog = self.sharded.compile_program(xs, zs, cs, ps, hf, hamiltonian=self._ham, rot_phi0=p0)
        return self._prog

    def _theta(self, theta):
        theta = np.asarray(theta, np.float64).reshape(-1)
        if theta.shape[0] < self._K:
            raise ValueError(f"expected {self._K} parameters")
        return theta[: self._K]

    def energy(self, theta):
        if self._ham is None:
            raise RuntimeError("no Hamiltonian set")
        return self.sharded.program_energy(self._program(), self._theta(theta))

    def energy_batch(self, thetas):
        thetas = np.ascontiguousarray(thetas, np.float64)
        if thetas.ndim != 2 or thetas.shape[1] != self._K:
            raise ValueError(f"expected a (B, {self._K}) array")
        return np.array([self.energy(t) for t in thetas])

    def energy_gradient(self, theta):
        raise NotImplementedError("the adjoint gradient is a one-device path; on the partitioned register use the default "
                                  "(forward-difference) Jacobian of the optimiser")

    def prepare_state(self, theta):
        self.sharded.run_program(self._program(), self._theta(theta))

    def get_state(self):
        if self.nbqbits > GATHER_MAX_QUBITS:
            raise MemoryError(f"get_state would gather 2^{self.nbqbits} amplitudes on every rank")
        return np.asarray(self.sharded.gather_state())

    def get_support(self, capacity=None):
        return None          # (the non-zero list of a one-device handle: callers fall back to get_state)

    def program_info(self):
        prog = self._program()
        return {"rotations": int(len(prog["coeff"])), "exchanges": int(prog["swaps"]), "real_stream": int(bool(prog["real"])),
                "partitioned_over": self.sharded.world}

    # -- the state of the ADAPT screens ------------------------------------------------------------------------------------------
    def init_basis(self, index):
        self.sharded.perm = list(range(self.nbqbits))
        self.sharded.init_basis(int(index))

    def norm2(self):
        return self.sharded.norm2()

    def apply_pauli_rotations(self, xs, zs, phis):
        self.sharded.apply_pauli_rotations(xs, zs, phis)

    def apply_exp_pauli_sum(self, operator, theta, prefactor=1.0):
        """psi <- exp(theta * prefactor * operator) psi, exact (ref:openvqe/adapt/fermionic_adapt_vqe.py:35-38 expm_multiply,
        ref:openvqe/adapt/qubit_adapt_vqe.py:45-52 expm)"""
        xs, zs, cs = pack_terms(self.nbqbits, operator.terms)
        cs = cs * prefactor
        if getattr(operator, "constant_coeff", 0.0):
            raise ValueError("apply_exp_pauli_sum: operator with a constant term")
        xs = [int(v) for v in xs]
        zs = [int(v) for v in zs]
        imaginary = bool(np.all(np.abs(cs.real) <= 1e-14 * np.maximum(1.0, np.abs(cs.imag))))
        commute = all((bin(xs[a] & zs[b]).count("1") + bin(zs[a] & xs[b]).count("1")) % 2 == 0
                      for a in range(len(xs)) for b in range(a))
        if imaginary and commute:
            # exp(theta sum_j i a_j P_j) = prod_j exp(-i (-theta a_j) P_j) for commuting strings: local sweeps and half-shard exchanges
            self.sharded.apply_pauli_rotations(xs, zs, [-float(theta) * float(c.imag) for c in cs])
            return
        self._taylor(xs, zs, cs, float(theta))

    def _taylor(self, xs, zs, cs, theta, tol=1e-30):
        """sum_k (theta A)^k / k! psi with sigma = A psi on the partitioned register; the series stops when a term's squared norm
        falls below ``tol`` of the state's (|theta A| is a few tenths for the pool operators of an ADAPT run: about twenty terms)"""
        import torch
        import torch.distributed as dist
        sh = self.sharded
        sh._complex_storage()
        state = sh.engine.tensor
        acc = state.clone()
        real_before = sh.real
        sh.real = False                       # (partner reads of the intermediate vectors travel complex)
        for k in range(1, 200):
            sigma = sh.apply_hamiltonian(xs, zs, cs, 0.0)
            sigma.mul_(theta / k)
            state.copy_(sigma)
            acc.add_(sigma)
            n2 = torch.stack([(sigma.abs() ** 2).sum().real, (acc.abs() ** 2).sum().real])
            if sh._dist:
                dist.all_reduce(n2, group=sh.group)
            if float(n2[0]) <= tol * float(n2[1]):
                break
        else:
            raise RuntimeError("apply_exp_pauli_sum: Taylor series did not converge")
        state.copy_(acc)
        # the exponential of a REAL antisymmetric matrix keeps a real state real: strings with an odd number of Y, imaginary coefficients
        sh.real = real_before and all(bin(x & z).count("1") & 1 for x, z in zip(xs, zs)) and \
            bool(np.all(np.abs(np.asarray(cs).real) <= 1e-14 * np.maximum(1.0, np.abs(np.asarray(cs).imag))))

    def expectation(self, hamiltonian):
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        coeff = np.array([_real_coeff(c, "observable") for c in cs], np.float64)
        const = _real_coeff(getattr(hamiltonian, "constant_coeff", 0.0) or 0.0, "observable constant")
        return self.sharded.expectation(xs, zs, coeff, const)

    def pool_gradients(self, pool_ops, mode):
        """gradient screen over ``pool_ops`` on the resident state with the stored Hamiltonian (same values on every rank)"""
        if self._ham is None:
            raise RuntimeError("no Hamiltonian set")
        cache = self._pool_cache
        if cache is None or cache[0] is not pool_ops or cache[1] != len(pool_ops):
            packed = []
            for op in pool_ops:
                px, pz, pc = pack_terms(self.nbqbits, op.terms)
                packed.append(([int(v) for v in px], [int(v) for v in pz], [complex(c) for c in pc]))
            cache = self._pool_cache = (pool_ops, len(pool_ops), packed)
        return np.asarray(self.sharded.pool_gradients(self._ham, cache[2], "fermionic" if int(mode) == GRAD_FERMIONIC else "qubit"))

    # -- one-device features ----------------------------------------------------------------------------------------------------
    def ground_state(self, *args, **kwargs):
        raise NotImplementedError("device Lanczos is a one-device path (ovqe_ground_state); the partitioned register offers energies, "
                                  "states and gradient screens")

    sector_ground_state = ground_state

    def last_screen_support(self):
        return -1

    def last_screen_sector(self):
        return 0


def make_backend(nbqbits, device=None, statevector_cls=None):
    """the statevector object for a register of ``nbqbits`` qubits in THIS process: partitioned across the ranks of the process group
    when ``replicas.partitioned`` says so, else the one-device handle on this rank's GPU (``statevector_cls``: the caller module's
    ``Statevector`` name — what the CPU tests replace by their oracle-backed stand-in)"""
    from . import replicas
    if replicas.partitioned(nbqbits):
        return PartitionedStatevector(nbqbits, device=replicas.device() if device is None else device)
    if statevector_cls is None:
        from .backend import Statevector as statevector_cls
    return statevector_cls(nbqbits, device=replicas.device() if device is None else device)
