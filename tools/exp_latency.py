"""single-evaluation latency of ovqe_energy (what scipy's optimisers see) for H2 / LiH / H2O UCCSD"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
for name in ("H2-STO3G-WSSVQE", "LIH", "H2O"):
    mol = chem.molecule(name); mol.rhf(); ham = mol.jw_hamiltonian()
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2); K = len(gens)
    th = np.random.default_rng(K).uniform(-0.1, 0.1, (200, K))
    with Statevector(ham.nbqbits) as sv:
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, mol.hf_init())
        e_b = sv.energy_batch(th)
        for k in range(20): sv.energy(th[k])
        t = time.perf_counter()
        e_s = [sv.energy(th[k]) for k in range(200)]
        dt = (time.perf_counter() - t) / 200
        print(f"{name:16s} n={ham.nbqbits:2d}: {dt*1e6:7.1f} us per ovqe_energy call ({1/dt:,.0f} evaluations/s sequential); "
              f"max|single - batch| = {np.abs(np.array(e_s) - e_b).max():.1e}", flush=True)
