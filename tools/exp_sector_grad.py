"""exact gradient on the sector tables against the dense-state adjoint pass: 2*m qubits, molecule-shaped UCCSD"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
opts = [a[6:].split("=") for a in sys.argv if a.startswith("--opt=")]
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
n = 2 * m
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, len(gens))
res = {}
for sector in ((1,) if "--sector-only" in sys.argv else (1, 0)):
    with Statevector(n) as sv:
        sv.set_option("sector", sector)
        sv.set_option("sector_min_qubits", 8)
        for k, v in opts: sv.set_option(k, int(v))
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        ts = []
        for rep in range(4 if sector else 2):
            t = time.perf_counter(); e, g = sv.energy_gradient(theta); ts.append(1e3 * (time.perf_counter() - t))
        t = time.perf_counter(); e2 = sv.energy(theta); te = 1e3 * (time.perf_counter() - t)
        res[sector] = (e, g)
        print(f"sector={sector} gradient ms={['%.2f' % t for t in ts]} energy ms={te:.2f} K={len(gens)} E={e:.12f} |g|={np.linalg.norm(g):.6f}", flush=True)
if 0 in res: print("max |dg|", np.abs(res[0][1] - res[1][1]).max(), "dE", abs(res[0][0] - res[1][0]))
