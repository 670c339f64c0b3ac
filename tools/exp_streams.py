"""circuit sweeps of the sector path: third form (per-wave streams, barriers at run boundaries) against the second form on the SAME
tables: per-launch microseconds from HIP events (option sector_profile), energies compared bit for bit.
usage: OVQE_LIB=testing exp_streams.py [m o]..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
args = [int(a) for a in sys.argv[1:] if "=" not in a]
cases = list(zip(args[0::2], args[1::2])) or [(12, 5)]
for m, o in cases:
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
    n = 2 * m
    theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
    with Statevector(n) as sv:
        sv.set_option("sector_min_qubits", 8)
        for k, v in opts.items():
            sv.set_option(k, int(v))
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        t0 = time.perf_counter()
        sv.energy(theta); sv.energy(theta)
        setup = 1e3 * (time.perf_counter() - t0)
        info = sv.program_info()
        S = max(info["sector_sweeps"], 1)
        print(f"m={m} o={o}: {len(gens)} generators, setup {setup:.0f} ms,", {k: v for k, v in info.items() if k.startswith("sector")}, flush=True)
        sv.set_option("sector_profile", 1)
        es = {}
        for sweep in (3, 2):
            sv.set_option("sector_sweep", sweep)
            for dbg in (0, 1):
                sv.set_option("sector_sweep_dbg", dbg)
                us, wall = [], []
                for _ in range(8):
                    t0 = time.perf_counter(); e = sv.energy(theta); wall.append(1e3 * (time.perf_counter() - t0))
                    us.append(sv.program_info()["sector_circuit_us"])
                if dbg == 0:
                    es[sweep] = e
                print(f"  sweep form {sweep} {'without ops' if dbg else 'full'}: circuit {min(us[1:])} us = {min(us[1:]) / S:.2f} us/sweep, evaluation wall {min(wall[1:]):.3f} ms, E {e!r}", flush=True)
            sv.set_option("sector_sweep_dbg", 0)
        print("  energies equal bit for bit:", es[3] == es[2], es[3] - es[2])
        sv.set_option("sector_profile", 0)
        gs = {}
        for sweep in (3, 2):
            sv.set_option("sector_sweep", sweep)
            e, g = sv.energy_gradient(theta)
            gs[sweep] = g
            t0 = time.perf_counter()
            for _ in range(5):
                e, g = sv.energy_gradient(theta)
            print(f"  form {sweep}: gradient {1e3 * (time.perf_counter() - t0) / 5:.3f} ms, |g| {np.linalg.norm(g):.12f}")
        print("  gradients: max |g3 - g2| =", np.abs(gs[3] - gs[2]).max())
