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
from ..common_files.host_threads import on_one_blas_thread
from .. import replicas
from ..evaluator import UCCEvaluator
from ..partitioned import make_backend
from .driver import AdaptEngine, Flavour, iterated_root_norm, rank_gradients
from ..qat_compat import Program, X, build_ucc_ansatz, get_default_qpu

_screens = {}
_evaluators = {}


def _screen_backend(nbqbits):
    if nbqbits not in _screens:
        _screens[nbqbits] = make_backend(nbqbits, None, Statevector)    # this rank's GPU, or the partitioned register (partitioned.make_backend)
    return _screens[nbqbits]


def _pool_gradients(screen, pool, mode):
    """the device screen; with several GPUs and the register on each of them, the pool's operators are shared between the ranks"""
    if replicas.active(screen.nbqbits):
        return replicas.pool_gradients(screen, pool, mode)
    return screen.pool_gradients(pool, mode)


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
    return [float(g) for g in _pool_gradients(screen, pool_mix, GRAD_QUBIT)]


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


_FLAVOUR = Flavour(
    title="Qubit ADAPT-VQE iteration: ",
    stall=1e-7,
    trace_keys={"energy": "energies", "error": "energies_substracted_from_fci", "norm": "norms", "leader": "Max_gradient"},
    unknown_criterion=" FAIL: Convergence criterion not defined",
    optimiser_display=False,
)
_RULE = " " + "-" * 74
_SHORT_RULE = " " + "-" * 54


def _boxed(text, rule=_RULE, pad=True):
    print(rule)
    if pad:
        print(" " * 58)
    print(text)
    if pad:
        print(" " * 58)
    print(rule)


@on_one_blas_thread
def qubit_adapt_vqe(hamiltonian_sp, hamiltonian_sp_sparse, reference_ket, nqubits, pool_mix, hf_init_sp, fci,
                    n_max_grads=2, adapt_conver="norm", adapt_thresh=1e-08, adapt_maxiter=45, tolerance_sim=1e-07,
                    method_sim="BFGS"):
    """Qubit ADAPT-VQE with the reference's signature and result schemas (qubit_adapt_vqe.py:310-605); returns
    (iterations_sim, iterations_ana, result_sim, result_ana), the 'ana' pair left empty like the reference.  The loop is
    ``adapt.driver.AdaptEngine``: device screen 2 |<psi|H P_i|psi>| over the pool of Pauli strings, ranking, stop test,
    ``n_max_grads`` new strings whose parameters start at (gradient / iterated-root norm), re-optimisation, record."""
    trace = {key: [] for key in (*_FLAVOUR.trace_keys.values(), *_FLAVOUR.gate_keys)}
    result_sim, result_ana = {}, {}
    iterations_ana = {key: [] for key in _FLAVOUR.trace_keys.values()}
    hf_circuit = prepare_hf_state(hf_init_sp, pool_mix)
    ref_energy = hf_energy(hf_circuit, hamiltonian_sp)
    state = {"screen": prepare_adapt_state(hf_init_sp, [], [], nqubits)}
    if getattr(state["screen"], "_ham_token", None) is not hamiltonian_sp:
        state["screen"].set_hamiltonian(hamiltonian_sp)
        state["screen"]._ham_token = hamiltonian_sp
    print("reference_energy from the simulator:", ref_energy)
    print("reference_energy from the analytical calculations:", state["screen"].expectation(hamiltonian_sp))
    _boxed(" " * 22 + "Start Qubit ADAPT-VQE algorithm:")
    print(" " * 58)
    how_many = int(n_max_grads)
    _boxed("        The number of maximum gradients inserted in each iteration: %d" % how_many, _SHORT_RULE, pad=False)

    def rebuild(selected, theta):
        state["screen"] = prepare_adapt_state(hf_init_sp, engine.generators, theta, nqubits)
        return prepare_state_ansatz(engine.generators, hf_init_sp, theta)

    def start_values(ranked, ranked_index, count_new):
        scale = iterated_root_norm(ranked)
        return [ranked[k] / scale for k in range(count_new)]

    engine = AdaptEngine(
        flavour=_FLAVOUR, pool=pool_mix,
        screen_gradients=lambda: calculate_gradients(pool_mix, hamiltonian_sp, state["screen"]),
        energy=lambda gens, t: ucc_action(hamiltonian_sp, gens, hf_init_sp, t),
        make_generator=lambda idx: pool_mix[idx], new_parameters=start_values, rebuild=rebuild)
    engine.circuit = hf_circuit
    for n_iter in range(adapt_maxiter):
        print("\n")
        print(_RULE)
        print(" " * 25 + _FLAVOUR.title, n_iter)
        print(_RULE)
        print("\n")
        _boxed("        Start the analytical gradient calculation:", _SHORT_RULE, pad=False)
        grads, norm, leader, _ = engine.screen()
        ranked, ranked_index = rank_gradients(grads)
        print("sorted_mylist_value of gradient_without_0", ranked)
        print(" Norm of <[H,A]> = %12.8f" % norm)
        print(" Max  of <[H,A]> = %12.8f" % leader)
        if engine.should_stop(norm, adapt_conver, adapt_thresh):
            print(" Ansatz Growth Converged!")
            result_sim.update(optimizer=method_sim, final_norm=norm, indices=engine.selected,
                              len_operators=len(engine.selected), parameters=engine.theta,
                              final_energy=engine.require_fit().fun)
            print(" -----------Final ansatz----------- ")
            print(" %4s %12s %18s" % ("#", "Coeff", "Term"))
            for k, t in enumerate(engine.theta):
                print(" %4i %12.8f" % (k, t))
            break
        engine.grow(ranked, ranked_index, how_many)
        print("initial parameters", engine.theta)
        print("op_indices of iteration_%d" % n_iter, engine.selected)
        fit = engine.optimise(method_sim, tolerance_sim)
        print(" ----------- ansatz from the simulator----------- ")
        print(" %s\t %s\t\t %s" % ("#", "Coeff", "Term"))
        for k, (t, idx) in enumerate(zip(engine.theta, engine.selected)):
            print(" %i\t %f\t %s" % (k, t, idx))
        print(" Energy reached from the simulator: %20.20f" % fit.fun)
        engine.previous_norm = norm
        engine.record(trace, {"energy": fit.fun, "error": abs(fit.fun - fci), "norm": norm, "leader": ranked[0]})
    return trace, iterations_ana, result_sim, result_ana
