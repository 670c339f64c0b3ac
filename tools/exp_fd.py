"""finite-difference-gradient-sized batches (B = K+1) and a few other small batch sizes on the fused kernels"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
for name in ("LIH", "H2O"):
    mol = chem.molecule(name); mol.rhf(); ham = mol.jw_hamiltonian()
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2); K = len(gens)
    with Statevector(ham.nbqbits) as sv:
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, mol.hf_init())
        for B in (16, 64, K + 1, 512, 1024, 2048):
            th = np.random.default_rng(B).uniform(-0.1, 0.1, (B, K))
            sv.energy_batch(th)
            t = time.perf_counter()
            for _ in range(20): sv.energy_batch(th)
            dt = (time.perf_counter() - t) / 20
            print(f"{name} B={B:5d}: {dt*1e3:.3f} ms  {B/dt/1e6:.2f} M evals/s", flush=True)
