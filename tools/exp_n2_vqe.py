"""UCCSD-VQE on N2 / cc-pVDZ (10 electrons, 12 orbitals) = 24 qubits, 1715 parameters (BASELINE configs[3]): scipy BFGS / L-BFGS
with the exact gradient of ovqe_energy_gradient (sector tables from the second call on)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import minimize
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
mol = chem.molecule("N2-CCPVDZ"); e_rhf = mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
size, _, spin_ops, theta_mp2, hf = prob.uccsd()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
method = args[0] if len(args) > 0 else "L-BFGS-B"
maxit = int(args[1]) if len(args) > 1 else 60
quccsd = "--quccsd" in sys.argv   # the reference's QUCCSD gate list (ref:openvqe/common_files/circuit.py) instead of UCCSD
print(f"E_RHF={e_rhf:.10f} E_MP2(full space)={mol.mp2_energy():.10f} parameters={size}", flush=True)
with Statevector(24) as sv:
    for a in sys.argv[1:]:
        if a.startswith("--opt="):
            k, v = a[6:].split("="); sv.set_option(k, int(v))
    sv.set_hamiltonian(ham)
    if quccsd:
        from openvqe_amd.common_files.circuit import quccsd_gate_list
        cluster_ops = prob.uccsd()[1]
        gates, K, _ = quccsd_gate_list(12, 5, 1, excitations=[op.terms[0].qbits for op in cluster_ops])
        sv.set_gate_program(gates, K, hf)
    else:
        sv.set_ucc_program(spin_ops, hf)
    calls = []
    def fun(th):
        t = time.perf_counter(); e, g = sv.energy_gradient(th); calls.append(time.perf_counter() - t)
        return e, g
    t0 = time.perf_counter()
    res = minimize(fun, np.array(theta_mp2), jac=True, method=method, options={"maxiter": maxit, "gtol": 1e-6, **({"ftol": 1e-14} if method == "L-BFGS-B" else {})})
    wall = time.perf_counter() - t0
    print(f"{method}: E={res.fun:.10f} iterations={res.nit} gradient calls={len(calls)} |g|inf={np.abs(res.jac).max():.2e} wall={wall:.2f}s "
          f"(first two calls {calls[0]:.2f}+{calls[1]:.2f}s, then {1e3*np.median(calls[2:]):.1f} ms each)  E(theta_MP2)={fun(np.array(theta_mp2))[0]:.10f}", flush=True)
    print(sv.program_info())
    if not quccsd:
        t0 = time.perf_counter(); e_fci, r, its = sv.sector_ground_state(tol=1e-10); t = time.perf_counter() - t0
        print(f"FCI of the (5 alpha, 5 beta) sector (Lanczos on the sector tables): E={e_fci:.10f} residual={r:.1e} iterations={its} in {t:.2f}s; "
              f"E_UCCSD - E_FCI = {res.fun - e_fci:.3e}")
