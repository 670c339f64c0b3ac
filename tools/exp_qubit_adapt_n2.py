"""qubit ADAPT-VQE mirror (ref:openvqe/adapt/qubit_adapt_vqe.py) on N2 / cc-pVDZ (10e,12o) = 24 qubits with the 'full_without_Z' pool
of the UCCSD cluster operators: a few macro-iterations, wall time per phase"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.adapt import qubit_adapt_vqe as qav
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
size, cluster_ops, spin_ops, theta_mp2, hf = prob.uccsd()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pool_size, pool = pools.generate_pool_from_cluster("full_without_Z", spin_ops, 24)
print(f"pool {pool_size} strings, built in {time.perf_counter()-t:.1f}s", flush=True)
marks = []
def _timed(name):
    orig = getattr(qav, name)
    def wrapper(*a, **k):
        t = time.perf_counter(); r = orig(*a, **k); marks.append((name, time.perf_counter() - t)); return r
    setattr(qav, name, wrapper)
names = ("calculate_gradients", "ucc_action", "prepare_adapt_state", "prepare_state_ansatz", "hf_energy")
for name in names:
    _timed(name)
buf = io.StringIO()
t0 = time.perf_counter()
try:
    with contextlib.redirect_stdout(buf):
        trace, _, result, _ = qav.qubit_adapt_vqe(ham, None, None, 24, pool, hf, -109.0765315037, n_max_grads=1, adapt_maxiter=iters,
                                                  method_sim="BFGS")
except Exception:
    print(buf.getvalue()[-1500:]); raise
print(f"wall={time.perf_counter()-t0:.1f}s")
for name in names:
    d = [x for k, x in marks if k == name]
    print(f"  {name}: {len(d)} calls, {sum(d):.2f}s" + (f" (median {np.median(d)*1e3:.2f} ms)" if d else ""))
print("energies", [float(e) for e in trace["energies"]] if "energies" in trace else {k: v for k, v in trace.items() if v})
