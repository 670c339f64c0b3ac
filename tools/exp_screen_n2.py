"""fermionic ADAPT screen on N2 / cc-pVDZ (10e,12o), 24 qubits, state of five spin-adapted generators: wall time per
ovqe_pool_gradients call (665-operator singlet pool), for profiling (rocprofv3 --kernel-trace --stats -- python3 tools/exp_screen_n2.py)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.backend import GRAD_FERMIONIC, Statevector
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, singlets = pools.singlet_sd(10, 12)
nops = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sv = Statevector(24)
sv.set_hamiltonian(ham)
sv.init_basis(hf)
rng = np.random.default_rng(0)
for k in rng.choice(len(singlets), nops, replace=False):
    sv.apply_exp_pauli_sum(singlets[k], 0.2)
for rep in range(6):
    t = time.perf_counter(); g = sv.pool_gradients(singlets, GRAD_FERMIONIC); dt = time.perf_counter() - t
    print(f"screen {dt*1e3:.2f} ms, support {sv.last_screen_support()}, max |g| {np.abs(g).max():.6f}", flush=True)
