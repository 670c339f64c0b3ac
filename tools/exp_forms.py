"""which sector-path kernel forms the automatic selection takes, per geometry (Statevector.sector_forms): molecule-shaped UCCSD at
16-26 qubits, energy, gradient and a batch; QUCCSD gate list at 18 qubits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
for m, o in ((9, 3), (9, 4), (10, 3), (10, 5), (11, 5), (12, 5), (12, 6), (13, 6)):
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=m)
    rng = np.random.default_rng(m)
    th = rng.uniform(-0.1, 0.1, len(gens))
    with Statevector(2 * m) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        t0 = time.perf_counter()
        for _ in range(3):
            sv.energy(th)
        f_e = sorted(sv.sector_forms())
        sv.energy_gradient(th)
        f_g = sorted(sv.sector_forms() - set(f_e))
        sv.energy_batch(np.tile(th, (6, 1)))
        f_b = sorted(sv.sector_forms() - set(f_e) - set(f_g))
        info = sv.program_info()
        print(f"{2*m} qubits ({m},{o}): K={len(gens)} support={info['sector_support']} sweeps={info['sector_sweeps']} pairs={info['sector_pairs']}"
              f" energy {f_e} gradient +{f_g} batch +{f_b}  ({time.perf_counter()-t0:.1f} s)", flush=True)
