"""QUCCSD (gate-level "efficient" excitation circuits) on the MI355X backend — mirror of
ref:openvqe/ucc_family/get_energy_qucc.py (same class/method names, argument order, printed lines,
result schemas).  The literal RY/RZ/H/CNOT list of circuit.py is traced ONCE with symbolic angles
and executed by libovqe_sv's gate program for every optimiser step."""
import numpy as np
import scipy.optimize
from numpy import binary_repr

from ..common_files import bfgs
from ..common_files.circuit import count, efficient_fermionic_ansatz
from ..evaluator import GateEvaluator
from ..qat_compat import AffineParam, Program, X, lower_circuit


def _hf_x_gates(prog, register, hf_init_sp, nbqbits):
    """X on qubit j when character j of binary_repr(hf) is '1' — no zero padding, exactly like
    get_energy_qucc.py:40-45 (an HF integer whose qubit 0 is empty raises IndexError there too)."""
    bits = [int(c) for c in binary_repr(hf_init_sp)]
    for j in range(nbqbits):
        if bits[j] == 1:
            prog.apply(X, register[j])


def _excitation_indices(cluster_ops):
    return [op.terms[0].qbits for op in cluster_ops]


class EnergyUCC:
    #: opt-in: exact Jacobian by the adjoint method (ovqe_energy_gradient) instead of scipy's forward differences
    adjoint_gradient = False

    def __init__(self):
        self._cache = None

    def _circuit(self, hamiltonian_sp, hf_init_sp, cluster_ops, theta):
        prog = Program()
        q = prog.qalloc(hamiltonian_sp.nbqbits)
        _hf_x_gates(prog, q, hf_init_sp, hamiltonian_sp.nbqbits)
        efficient_fermionic_ansatz(q, prog, _excitation_indices(cluster_ops), theta)
        return prog.to_circ()

    def _evaluator(self, hamiltonian_sp, cluster_ops, hf_init_sp):
        c = self._cache
        if c is None or c[0] is not hamiltonian_sp or c[1] is not cluster_ops or c[2] != hf_init_sp:
            K = len(cluster_ops)
            traced = self._circuit(hamiltonian_sp, hf_init_sp, cluster_ops, [AffineParam(k) for k in range(K)])
            hf, kind, gates = lower_circuit(traced)
            if kind != "gates":
                gates = []
            self._cache = (hamiltonian_sp, cluster_ops, hf_init_sp, GateEvaluator(hamiltonian_sp, gates, K, hf))
        return self._cache[3]

    def action_quccsd(self, theta_0, hamiltonian_sp, cluster_ops, hf_init_sp, energies=[]):
        """Energy of the QUCCSD circuit at theta_0 (get_energy_qucc.py:11-56); appended to ``energies``."""
        ev = self._evaluator(hamiltonian_sp, cluster_ops, hf_init_sp)
        value = ev.energy(np.asarray(theta_0, dtype=float)[: ev.n_params])
        energies.append(value)
        return value

    def prepare_hf_state(self, hf_init_sp, cluster_ops_sp):
        prog = Program()
        nbqbits = cluster_ops_sp[0].nbqbits
        qb = prog.qalloc(nbqbits)
        _hf_x_gates(prog, qb, hf_init_sp, nbqbits)
        return prog.to_circ()

    def prepare_state_ansatz(self, hamiltonian_sp, hf_init_sp, cluster_ops, theta):
        return self._circuit(hamiltonian_sp, hf_init_sp, cluster_ops, theta)

    def get_energies(self, hamiltonian_sp, cluster_ops, hf_init_sp, theta_current1, theta_current2, FCI):
        """BFGS (tol 1e-5) from the MP2 guess and from the constant guess (get_energy_qucc.py:136-244)."""
        iterations = {
            "minimum_energy_result1_guess": [],
            "minimum_energy_result2_guess": [],
            "theta_optimized_result1": [],
            "theta_optimized_result2": [],
        }
        result = {}
        tolerance = 10 ** (-5)
        method = "BFGS"
        print("tolerance= ", tolerance)
        print("method= ", method)
        energies1, energies2 = [], []
        runs = []
        jac = None
        fun_of = lambda sink: (lambda theta: self.action_quccsd(theta, hamiltonian_sp, cluster_ops, hf_init_sp, sink))  # noqa: E731
        if self.adjoint_gradient:
            # one device pass gives E and all of dE/dtheta: the optimiser asks for both at every trial point of its line search,
            # so the energy call keeps the gradient for the Jacobian call at the same theta (and appends to `energies` as ever)
            ev = self._evaluator(hamiltonian_sp, cluster_ops, hf_init_sp)
            last = {}

            def both(theta):
                key = np.asarray(theta, dtype=float).tobytes()
                if last.get("key") != key:
                    e, g = ev.energy_gradient(np.asarray(theta, dtype=float))
                    full = np.zeros(len(theta))
                    full[: ev.n_params] = g
                    last.update(key=key, e=float(e), g=full)
                return last

            def fun_of(sink):   # noqa: F811
                def fun(theta):
                    e = both(theta)["e"]
                    sink.append(e)
                    return e
                return fun

            def jac(theta):
                return both(theta)["g"].copy()
        for x0, sink in ((theta_current1, energies1), (theta_current2, energies2)):
            # (bfgs.minimize IS scipy.optimize.minimize below 256 parameters or without a Jacobian; above, the same BFGS with its
            # inverse-Hessian update in rank-two form)
            runs.append(bfgs.minimize(fun_of(sink), x0=x0, jac=jac, method=method, tol=tolerance,
                                      options={"maxiter": 50000, "disp": True}))
        opt_result1, opt_result2 = runs
        theta_optimized_result1 = [opt_result1.x[si] for si in range(len(theta_current1))]
        theta_optimized_result2 = [opt_result2.x[si] for si in range(len(theta_current2))]
        gates1 = self.prepare_state_ansatz(hamiltonian_sp, hf_init_sp, cluster_ops, theta_optimized_result1).ops
        gates2 = self.prepare_state_ansatz(hamiltonian_sp, hf_init_sp, cluster_ops, theta_optimized_result2).ops
        iterations["minimum_energy_result1_guess"].append(opt_result1.fun)
        iterations["minimum_energy_result2_guess"].append(opt_result2.fun)
        iterations["theta_optimized_result1"].append(theta_optimized_result1)
        iterations["theta_optimized_result2"].append(theta_optimized_result2)
        result["CNOT1"] = count("CNOT", gates1)
        result["CNOT2"] = count("CNOT", gates2)
        result["len_op1"] = len(theta_optimized_result1)
        result["len_op2"] = len(theta_optimized_result2)
        result["energies_1"] = energies1
        result["energies_2"] = energies2
        result["energies1_substracted_from_FCI"] = abs(opt_result1.fun - FCI)
        result["energies2_substracted_from_FCI"] = abs(opt_result2.fun - FCI)
        return iterations, result
