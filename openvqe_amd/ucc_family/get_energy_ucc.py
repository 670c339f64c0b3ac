"""UCC-family VQE on the MI355X backend — mirror of ref:openvqe/ucc_family/get_energy_ucc.py
(class name, method names, argument order, printed lines and the returned dict schemas are the
reference's; the simulation runs in libovqe_sv instead of myQLM)."""
import numpy as np
import scipy.optimize

from ..common_files import bfgs

from ..common_files.circuit import count
from ..evaluator import UCCEvaluator
from ..qat_compat import Program, build_ucc_ansatz


class EnergyUCC:
    #: opt-in: supply scipy with a forward-difference Jacobian computed by ONE batched device call
    #: (same step as scipy's jac=None path, so the iterates coincide)
    batched_gradient = False
    #: opt-in: exact Jacobian by the adjoint method (ovqe_energy_gradient): about three circuit executions for all K
    #: derivatives; iterates differ from the reference's forward-difference path at the 1e-8 level
    adjoint_gradient = False

    def __init__(self):
        self._cache = {}
        self._last = None

    def _evaluator(self, hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params):
        last = self._last   # (the optimiser's loop asks for the same evaluator tens of thousands of times)
        if last is not None and last[0] is hamiltonian_sp and last[1] is cluster_ops_sp and last[2] == hf_init_sp and last[3] == n_params:
            return last[4]
        ev = self._evaluator_lookup(hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params)
        self._last = (hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params, ev)
        return ev

    def _evaluator_lookup(self, hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params):
        key = (id(hamiltonian_sp), id(cluster_ops_sp), int(hf_init_sp), int(n_params))
        ev = self._cache.get(key)
        if ev is None or ev.generators_ref is not cluster_ops_sp or ev.hamiltonian is not hamiltonian_sp:
            ev = UCCEvaluator(hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params)
            ev.generators_ref = cluster_ops_sp
            self._cache = {key: ev}
        return ev

    def ucc_action(self, theta_current, hamiltonian_sp, cluster_ops_sp, hf_init_sp, energies=[]):
        """Energy of prod_k exp(-i theta_k G_k)|HF> (get_energy_ucc.py:8-50).  ``zip`` semantics: only the
        first min(len(ops), len(theta)) operators are applied.  Appends the value to ``energies``."""
        n_params = min(len(cluster_ops_sp), len(theta_current))
        ev = self._evaluator(hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params)
        value = ev.energy(theta_current)   # (Statevector.energy takes the first n_params entries)
        energies.append(value)
        return value

    def prepare_state_ansatz(self, hamiltonian_sp, cluster_ops_sp, hf_init_sp, parameters):
        """Circuit object of the optimised ansatz, for gate counting (get_energy_ucc.py:52-90)."""
        prog = Program()
        reg = prog.qalloc(hamiltonian_sp.nbqbits)
        for n_term, (term, theta_term) in enumerate(zip(cluster_ops_sp, parameters)):
            init = hf_init_sp if n_term == 0 else 0
            prog.apply(build_ucc_ansatz([term], init, n_steps=1)([theta_term]), reg)
        return prog.to_circ()

    def _minimize(self, hamiltonian_sp, ops, hf_init_sp, x0, energies, method, tolerance):
        fun = lambda theta: self.ucc_action(theta, hamiltonian_sp, ops, hf_init_sp, energies)  # noqa: E731
        jac = None
        if self.adjoint_gradient:
            n_params = min(len(ops), len(x0))
            ev = self._evaluator(hamiltonian_sp, ops, hf_init_sp, n_params)

            def jac(theta):
                g = np.zeros(len(theta))
                g[:n_params] = ev.energy_gradient(np.asarray(theta, dtype=float))[1]
                return g
        elif self.batched_gradient:
            n_params = min(len(ops), len(x0))
            ev = self._evaluator(hamiltonian_sp, ops, hf_init_sp, n_params)
            eps = np.sqrt(np.finfo(float).eps)

            def jac(theta):
                theta = np.asarray(theta, dtype=float)
                pts = np.tile(theta, (len(theta) + 1, 1))
                pts[1:] += eps * np.eye(len(theta))
                vals = ev.energy_batch(pts)
                return (vals[1:] - vals[0]) / eps
        # (bfgs.minimize IS scipy.optimize.minimize below 256 parameters or without a Jacobian; above, the same BFGS with its
        # inverse-Hessian update in rank-two form: O(n^2) instead of scipy's two n x n matrix products per iteration)
        return bfgs.minimize(fun, x0=x0, jac=jac, method=method, tol=tolerance, options={"maxiter": 50000, "disp": True})

    def get_energies(self, hamiltonian_sp, cluster_ops_sp, pool_generator, hf_init_sp, theta_current1,
                     theta_current2, fci):
        """Two BFGS minimisations (fermionic cluster operators, then the qubit-pool operators) and the
        gate/accuracy summary (get_energy_ucc.py:92-206)."""
        iterations = {
            "minimum_energy_result1_guess": [],
            "minimum_energy_result2_guess": [],
            "theta_optimized_result1": [],
            "theta_optimized_result2": [],
        }
        result = {}
        tolerance = 10 ** (-4)
        method = "BFGS"
        print("tolerance= ", tolerance)
        print("method= ", method)
        energies_1, energies_2 = [], []
        opt_result1 = self._minimize(hamiltonian_sp, cluster_ops_sp, hf_init_sp, theta_current1, energies_1, method,
                                     tolerance)
        opt_result2 = self._minimize(hamiltonian_sp, pool_generator, hf_init_sp, theta_current2, energies_2, method,
                                     tolerance)
        theta_optimized_result1 = [opt_result1.x[si] for si in range(len(theta_current1))]
        theta_optimized_result2 = [opt_result2.x[si] for si in range(len(theta_current2))]
        # NB both circuits are rebuilt from cluster_ops_sp, as in the reference (lines 184-189): CNOT2 == CNOT1
        gates1 = self.prepare_state_ansatz(hamiltonian_sp, cluster_ops_sp, hf_init_sp, theta_optimized_result1).ops
        gates2 = self.prepare_state_ansatz(hamiltonian_sp, cluster_ops_sp, hf_init_sp, theta_optimized_result2).ops
        iterations["minimum_energy_result1_guess"].append(opt_result1.fun)
        iterations["minimum_energy_result2_guess"].append(opt_result2.fun)
        iterations["theta_optimized_result1"].append(theta_optimized_result1)
        iterations["theta_optimized_result2"].append(theta_optimized_result2)
        result["CNOT1"] = count("CNOT", gates1)
        result["CNOT2"] = count("CNOT", gates2)
        result["len_op1"] = len(theta_optimized_result1)
        result["len_op2"] = len(theta_optimized_result2)
        result["energies1_substracted_from_FCI"] = abs(opt_result1.fun - fci)
        result["energies2_substracted_from_FCI"] = abs(opt_result2.fun - fci)
        result["energies_1"] = energies_1
        result["energies_2"] = energies_2
        return iterations, result
