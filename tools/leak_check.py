"""create / use / destroy handles in a loop and watch the device memory in use (diagnostics)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector, GRAD_FERMIONIC
ham, gens, hf = fermion.synthetic_molecule(8, 3, seed=1)
pool = fermion.uccsd_pool_antihermitian(8, 3)[:50]
theta = np.zeros(len(gens)) + 0.05
def used(): f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
base = None
for it in range(30):
    with Statevector(16) as sv:
        sv.set_option("force_path", 2)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        sv.energy(theta); sv.prepare_state(theta); sv.pool_gradients(pool, GRAD_FERMIONIC); sv.energy_gradient(theta)
        sv.ground_state(tol=1e-6, max_iter=20); sv.expectation(ham)
        sv.set_option("real_stream", 0); sv.energy(theta)
    if it == 4: base = used()
    if it % 5 == 4: print(f"iteration {it}: {used():.1f} MiB in use", flush=True)
print("growth since iteration 4:", round(used() - base, 2), "MiB")
