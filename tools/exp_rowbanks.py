"""materialised <H> of the sector path with and without the bank-aware order of the rows' elements (option sector_row_banks):
k_sector_expect microseconds (HIP events), evaluation wall time, energies.  usage: exp_rowbanks.py [m o]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, len(gens))
ref = None
for rb in (0, 1):
    with Statevector(2 * m) as sv:
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_row_banks", rb)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        sv.energy(theta)
        t0 = time.perf_counter(); sv.energy(theta); tb = time.perf_counter() - t0
        sv.set_option("sector_profile", 1)
        us, wall = [], []
        for _ in range(6):
            t0 = time.perf_counter(); e = sv.energy(theta); wall.append(1e3 * (time.perf_counter() - t0))
            us.append(sv.program_info()["sector_expect_us"])
        eg, g = sv.energy_gradient(theta)
        t0 = time.perf_counter(); sv.energy_gradient(theta); tg = 1e3 * (time.perf_counter() - t0)
        hb = sv.program_info()["sector_h_stream_bytes"]
        if ref is None: ref = (e, g)
        print(f"row_banks={rb}: <H> {min(us[1:])} us = {hb / min(us[1:]) / 1e6:.2f} TB/s, evaluation {min(wall[1:]):.3f} ms, gradient {tg:.2f} ms, "
              f"tables built in {tb:.2f} s, dE {e - ref[0]:.1e}, max |dg| {np.abs(g - ref[1]).max():.1e}", flush=True)
