"""batched evaluations on the sector tables (ovqe_energy_batch) against one evaluation at a time: molecule-shaped UCCSD at 2 m
qubits.  usage: exp_sector_batch.py [m o] [--B=64]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
Bs = [int(a[4:]) for a in sys.argv if a.startswith("--B=")] or [2, 8, 64]
opts = [a[6:].split("=") for a in sys.argv if a.startswith("--opt=")]
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
rng = np.random.default_rng(1)
with Statevector(2 * m) as sv:
    sv.set_option("sector_min_qubits", 8)
    for k, v in opts: sv.set_option(k, int(v))
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    th = rng.uniform(-0.1, 0.1, (max(Bs), len(gens)))
    sv.energy(th[0]); sv.energy(th[0])
    t0 = time.perf_counter(); serial = np.array([sv.energy(t) for t in th[:8]]); t_serial = (time.perf_counter() - t0) / 8
    print(f"serial: {1e3 * t_serial:.3f} ms per evaluation", sv.program_info()["sector_support"], flush=True)
    for B in Bs:
        sv.energy_batch(th[:B])
        t0 = time.perf_counter(); e = sv.energy_batch(th[:B]); dt = time.perf_counter() - t0
        print(f"B={B}: {1e3 * dt / B:.3f} ms per evaluation = {t_serial / (dt / B):.2f} x the serial rate, "
              f"max |dE| vs serial {np.abs(e[:8] - serial[:min(B, 8)]).max() if B >= 8 else np.abs(e - serial[:B]).max():.1e}", flush=True)
