"""time to the first energy from the tables, with and without the per-wave streams (sector_sweep 3 / 2 at build time): wall time of
the table build's phases (sector_debug 4, stderr).  usage: OVQE_LIB=testing exp_stream_setup.py [m o]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 5)
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
for rep in range(2):
    for form in (2, 3):
        with Statevector(2 * m) as sv:
            sv.set_option("sector_min_qubits", 8)
            sv.set_option("sector_sweep", form)
            sv.set_option("sector_debug", 4 if rep else 0)
            sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); sv.energy(theta); ts.append(1e3 * (time.perf_counter() - t0))
            print(f"form {form}: calls {[round(t, 2) for t in ts]} ms", flush=True)
