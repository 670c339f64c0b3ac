"""Qubit ADAPT-VQE on the MI355X backend — mirror of ref:openvqe/adapt/qubit_adapt_vqe.py (names,
argument order, printed lines, convergence logic, result schemas).  The per-iteration Kronecker
rebuild of every pool matrix and the scipy products of the reference (qubit_adapt_vqe.py:81-150,462-471)
are replaced by one device gradient screen g_i = 2 |<psi|H P_i|psi>|; the screen state is
prod_k exp(-i theta_k P_k)|HF> with exact exponentials (qubit_adapt_vqe.py:20-55), the energies come
from the compiled Pauli-rotation program.
"""
import numpy as np
import scipy.optimize
from numpy import binary_repr

from ..backend import GRAD_QUBIT, Statevector
from ..common_files.circuit import count
from ..common_files.sorted_gradient import abs_sort_desc, corresponding_index, index_without_0, value_without_0
from ..evaluator import UCCEvaluator
from ..qat_compat import Program, X, build_ucc_ansatz, get_default_qpu

_screens = {}
_evaluators = {}


def _screen_backend(nbqbits):
    if nbqbits not in _screens:
        _screens[nbqbits] = Statevector(nbqbits)
    return _screens[nbqbits]


def prepare_adapt_state(hf_init_sp, ansatz, coefficients, nbqbits=None):
    """device state prod_k exp(-i c_k O_k)|HF>, O_k a pool operator (single Pauli string in every
    pool of qubit_pool.py) — returns the backend holding it"""
    nbqbits = nbqbits or ansatz[0].nbqbits
    sv = _screen_backend(nbqbits)
    sv.init_basis(hf_init_sp)
    for theta, op in zip(coefficients, ansatz):
        sv.apply_exp_pauli_sum(op, theta, prefactor=-1j)
    return sv


def calculate_gradients(pool_mix, hamiltonian_sp, screen):
    """2 |<psi|H P_i|psi>| for the whole pool in one device call (qubit_adapt_vqe.py:126-150)"""
    if getattr(screen, "_ham_token", None) is not hamiltonian_sp:
        screen.set_hamiltonian(hamiltonian_sp)
        screen._ham_token = hamiltonian_sp
    return [float(g) for g in screen.pool_gradients(pool_mix, GRAD_QUBIT)]


def prepare_state_ansatz(cluster_ops_sp, hf_init_sp, parameters):
    prog = Program()
    reg = prog.qalloc(cluster_ops_sp[0].nbqbits)
    for n_term, (term, theta_term) in enumerate(zip(cluster_ops_sp, parameters)):
        init = hf_init_sp if n_term == 0 else 0
        prog.apply(build_ucc_ansatz([term], init, n_steps=1)([theta_term]), reg)
    return prog.to_circ()


def compute_commutator_i(commutator, curr_state):
    return get_default_qpu().submit(curr_state.to_job(job_type="OBS", observable=commutator)).value


def prepare_hf_state(hf_init_sp, cluster_ops_sp):
    prog = Program()
    nbqbits = cluster_ops_sp[0].nbqbits
    bits = [int(c) for c in binary_repr(hf_init_sp)]
    qb = prog.qalloc(nbqbits)
    for j in range(nbqbits):
        if bits[j] == 1:
            prog.apply(X, qb[j])
    return prog.to_circ()


def hf_energy(hf_state, hamiltonian_sp):
    return get_default_qpu().submit(hf_state.to_job(job_type="OBS", observable=hamiltonian_sp)).value


def ucc_action(hamiltonian_sp, cluster_ops_sp, hf_init_sp, theta_current):
    """E(theta) of the Trotterised ansatz (qubit_adapt_vqe.py:271-307)."""
    n_params = min(len(cluster_ops_sp), len(theta_current))
    key = (id(hamiltonian_sp), id(cluster_ops_sp), int(hf_init_sp), n_params)
    ev = _evaluators.get(key)
    if ev is None or ev.hamiltonian is not hamiltonian_sp or ev.generators_ref is not cluster_ops_sp \
            or ev.generators[:n_params] != list(cluster_ops_sp[:n_params]):
        ev = UCCEvaluator(hamiltonian_sp, cluster_ops_sp, hf_init_sp, n_params)
        ev.generators_ref = cluster_ops_sp
        _evaluators.clear()
        _evaluators[key] = ev
    return ev.energy(np.asarray(theta_current, dtype=float)[:n_params])


def qubit_adapt_vqe(hamiltonian_sp, hamiltonian_sp_sparse, reference_ket, nqubits, pool_mix, hf_init_sp, fci,
                    n_max_grads=2, adapt_conver="norm", adapt_thresh=1e-08, adapt_maxiter=45, tolerance_sim=1e-07,
                    method_sim="BFGS"):
    """The qubit-ADAPT loop of qubit_adapt_vqe.py:310-605; returns
    (iterations_sim, iterations_ana, result_sim, result_ana) with the 'ana' dicts left empty like the reference."""
    iterations_sim = {"energies": [], "energies_substracted_from_fci": [], "norms": [], "Max_gradient": [],
                      "CNOTs": [], "Hadamard": [], "RY": [], "RX": []}
    result_sim = {}
    iterations_ana = {"energies": [], "energies_substracted_from_fci": [], "norms": [], "Max_gradient": []}
    result_ana = {}
    parameters_sim = []
    parameters_ana = []
    ansatz_ops = []
    curr_state = prepare_hf_state(hf_init_sp, pool_mix)
    ref_energy = hf_energy(curr_state, hamiltonian_sp)
    screen = prepare_adapt_state(hf_init_sp, ansatz_ops, parameters_ana, nqubits)
    if getattr(screen, "_ham_token", None) is not hamiltonian_sp:
        screen.set_hamiltonian(hamiltonian_sp)
        screen._ham_token = hamiltonian_sp
    ref_energy_ana = screen.expectation(hamiltonian_sp)
    print("reference_energy from the simulator:", ref_energy)
    print("reference_energy from the analytical calculations:", ref_energy_ana)
    print(" --------------------------------------------------------------------------")
    print("                                                          ")
    print("                      Start Qubit ADAPT-VQE algorithm:")
    print("                                                          ")
    print(" --------------------------------------------------------------------------")
    print("                                                          ")
    Y = int(n_max_grads)
    print(" ------------------------------------------------------")
    print("        The number of maximum gradients inserted in each iteration:", Y)
    print(" ------------------------------------------------------")
    op_indices = []
    prev_norm = 0.0
    opt_result_sim = None
    for n_iter in range(adapt_maxiter):
        print("\n")
        print(" --------------------------------------------------------------------------")
        print("                         Qubit ADAPT-VQE iteration: ", n_iter)
        print(" --------------------------------------------------------------------------")
        next_deriv = 0
        curr_norm = 0
        print("\n")
        print(" ------------------------------------------------------")
        print("        Start the analytical gradient calculation:")
        print(" ------------------------------------------------------")
        list_grad = calculate_gradients(pool_mix, hamiltonian_sp, screen)
        for gi in list_grad:
            curr_norm += gi * gi
            if abs(gi) > abs(next_deriv):
                next_deriv = gi
        values = value_without_0(list_grad)
        indices = index_without_0(list_grad)
        sorted_values = abs_sort_desc(value_without_0(list_grad))
        print("sorted_mylist_value of gradient_without_0", sorted_values)
        sorted_index = corresponding_index(values, indices, sorted_values)
        curr_norm = np.sqrt(curr_norm)
        max_of_gi = next_deriv
        print(" Norm of <[H,A]> = %12.8f" % curr_norm)
        print(" Max  of <[H,A]> = %12.8f" % max_of_gi)
        converged = False
        if adapt_conver == "norm":
            if curr_norm < adapt_thresh:
                converged = True
        else:
            print(" FAIL: Convergence criterion not defined")
            raise SystemExit()
        if converged or (abs(curr_norm - prev_norm) < 10 ** (-7)):
            print(" Ansatz Growth Converged!")
            result_sim["optimizer"] = method_sim
            result_sim["final_norm"] = curr_norm
            result_sim["indices"] = op_indices
            result_sim["len_operators"] = len(op_indices)
            result_sim["parameters"] = parameters_sim
            result_sim["final_energy"] = opt_result_sim.fun
            print(" -----------Final ansatz----------- ")
            print(" %4s %12s %18s" % ("#", "Coeff", "Term"))
            for si in range(len(ansatz_ops)):
                print(" %4i %12.8f" % (si, parameters_sim[si]))
            break
        chosen_batch = sorted_values
        gamma1 = []
        sorted_index1 = []
        curr_norm1 = 0
        for z in chosen_batch:
            # the square root is taken INSIDE the accumulation loop in the reference (lines 530-532)
            curr_norm1 += z * z
            curr_norm1 = np.sqrt(curr_norm1)
        for i in range(Y):
            gamma1.append(chosen_batch[i] / curr_norm1)
            sorted_index1.append(sorted_index[i])
        for m in range(len(gamma1)):
            parameters_sim.append(gamma1[m])
            parameters_ana.append(gamma1[m])
            ansatz_ops.append(pool_mix[sorted_index1[m]])
            op_indices.append(sorted_index1[m])
        print("initial parameters", parameters_sim)
        print("op_indices of iteration_%d" % n_iter, op_indices)
        opt_result_sim = scipy.optimize.minimize(
            lambda theta: ucc_action(hamiltonian_sp, ansatz_ops, hf_init_sp, theta),
            x0=parameters_sim, method=method_sim, tol=tolerance_sim, options={"maxiter": 100000, "disp": False})
        xlist_sim = opt_result_sim.x
        print(" ----------- ansatz from the simulator----------- ")
        print(" %s\t %s\t\t %s" % ("#", "Coeff", "Term"))
        parameters_sim = []
        for si in range(len(ansatz_ops)):
            print(" %i\t %f\t %s" % (si, xlist_sim[si], op_indices[si]))
            parameters_sim.append(xlist_sim[si])
        print(" Energy reached from the simulator: %20.20f" % opt_result_sim.fun)
        curr_state = prepare_state_ansatz(ansatz_ops, hf_init_sp, parameters_sim)
        screen = prepare_adapt_state(hf_init_sp, ansatz_ops, parameters_sim, nqubits)
        prev_norm = curr_norm
        gates = curr_state.ops
        iterations_sim["energies"].append(opt_result_sim.fun)
        iterations_sim["energies_substracted_from_fci"].append(abs(opt_result_sim.fun - fci))
        iterations_sim["norms"].append(curr_norm)
        iterations_sim["Max_gradient"].append(sorted_values[0])
        iterations_sim["CNOTs"].append(count("CNOT", gates))
        iterations_sim["Hadamard"].append(count("H", gates))
        iterations_sim["RY"].append(count("RY", gates))
        iterations_sim["RX"].append(count("RX", gates))
    return iterations_sim, iterations_ana, result_sim, result_ana
