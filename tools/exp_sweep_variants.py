"""circuit sweeps of the sector path: first form (staged pair words, gather on read) against the second form (scatter on write,
64-bit pair words in registers) over tile bits / workgroup sizes, with and without the ops: per-launch microseconds from HIP
events (option sector_profile).  usage: exp_sweep_variants.py [m o] [--bits=14 --bits=16 ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
bits = [int(a[7:]) for a in sys.argv if a.startswith("--bits=")] or [0]
chunks = [int(a[8:]) for a in sys.argv if a.startswith("--chunk=")] or [4096]
bits = [(b, c) for b in bits for c in chunks]
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
n = 2 * m
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, len(gens))
for b, chunk in bits:
    variants = [(1, 1024)] + [(2, nt) for nt in (1024, 512, 256) if 1 <= chunk // nt <= 8 and not (nt == 256 and chunk == 4096)]
    with Statevector(n) as sv:
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_bits", b)
        sv.set_option("sector_chunk", chunk)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        sv.set_option("sector", 0)
        e_dense = sv.energy(theta)
        sv.set_option("sector", 1)
        sv.energy(theta); sv.energy(theta)
        info = sv.program_info()
        print("bits", b, {k: v for k, v in info.items() if k.startswith("sector")}, flush=True)
        sv.set_option("sector_profile", 1)
        for sweep, nt in variants:
            sv.set_option("sector_sweep", sweep); sv.set_option("sector_threads", nt)
            row = {}
            for dbg in (0, 1):
                sv.set_option("sector_debug", dbg)
                us, wall = [], []
                for _ in range(6):
                    t0 = time.perf_counter(); e = sv.energy(theta); wall.append(1e3 * (time.perf_counter() - t0))
                    us.append(sv.program_info()["sector_circuit_us"])
                row[dbg] = (min(us[1:]), min(wall[1:]), e)
            sv.set_option("sector_debug", 0)
            S = max(info["sector_sweeps"], 1)
            print(f"  bits={b} chunk={chunk} sweep={sweep} nt={nt}: circuit {row[0][0]} us = {row[0][0] / S:.2f} us/sweep (without ops {row[1][0] / S:.2f}), "
                  f"evaluation wall {row[0][1]:.3f} ms, dE vs dense {row[0][2] - e_dense:.2e}", flush=True)
