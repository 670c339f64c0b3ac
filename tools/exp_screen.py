"""ADAPT gradient screen (sigma = H psi, then g_k for every pool operator) at 2*m qubits (timing helper)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion, pools
from openvqe_amd.backend import Statevector, GRAD_FERMIONIC
m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 3)
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
t = time.time(); size, pool = pools.spin_complement_gsd(2 * o, m); tp = time.time() - t
n = 2 * m
print(f"n={n} pool {size} ({sum(len(p.terms) for p in pool)} strings, built in {tp:.1f} s); H terms {len(ham.terms)}", flush=True)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
with Statevector(n) as sv:
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf); sv.prepare_state(theta)
    for rep in range(3):
        t = time.time(); g = sv.pool_gradients(pool, GRAD_FERMIONIC); dt = time.time() - t
        print(f"pool_gradients: {dt*1e3:.2f} ms  (norm {np.linalg.norm(g):.6f})", flush=True)
