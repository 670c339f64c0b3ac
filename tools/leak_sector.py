"""create / use / destroy handles whose sector tables are built, invalidated (new Hamiltonian, new program) and rebuilt, and
watch the free device memory (diagnostics: the figure must not drift)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
ham, gens, hf = fermion.synthetic_molecule(10, 4, seed=3)
ham2, _, _ = fermion.synthetic_molecule(10, 4, seed=4)
th = np.random.default_rng(0).uniform(-0.2, 0.2, len(gens))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
base = free()
for rep in range(6):
    with Statevector(20) as sv:
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        for _ in range(3): sv.energy(th)
        sv.energy_gradient(th); sv.sector_ground_state()
        sv.set_hamiltonian(ham2)                      # tables invalidated and rebuilt
        for _ in range(3): sv.energy(th)
        sv.set_ucc_program(gens[::2], hf)
        for _ in range(3): sv.energy(th[::2])
        info = sv.program_info()
    print(rep, "free MiB after destroy: %.0f (start %.0f) support %d" % (free(), base, info["sector_support"]), flush=True)
