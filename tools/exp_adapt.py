"""H2O/STO-3G ADAPT gradient screen (1246-operator pool) and exact-exponential step: timings"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.backend import Statevector, GRAD_FERMIONIC
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian(); hf = mol.hf_init()
size, pool = pools.spin_complement_gsd(mol.n_elec, mol.nao)
print("pool", size, "terms", sum(len(p.terms) for p in pool))
with Statevector(14) as sv:
    sv.set_hamiltonian(ham); sv.init_basis(hf)
    for k in (5, 40, 300):
        if pool[k].terms: sv.apply_exp_pauli_sum(pool[k], 0.05)
    for rep in range(3):
        t = time.perf_counter(); g = sv.pool_gradients(pool, GRAD_FERMIONIC); dt = time.perf_counter() - t
        print(f"pool_gradients: {dt*1e3:.2f} ms  (norm {np.linalg.norm(g):.6f})")
    t = time.perf_counter()
    for k in range(20): sv.apply_exp_pauli_sum(pool[5], 0.01)
    print(f"apply_exp_pauli_sum: {(time.perf_counter()-t)/20*1e3:.2f} ms each")
