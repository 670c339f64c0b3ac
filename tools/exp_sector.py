"""sector path against the dense streaming path: molecule-shaped UCCSD at 2*m qubits (energies, timings, table sizes)"""
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
thetas = [rng.uniform(-0.1, 0.1, len(gens)) for _ in range(3)]
res = {}
for sector in ((1,) if "--sector-only" in sys.argv else (0, 1)):
    with Statevector(n) as sv:
        sv.set_option("sector", sector)
        sv.set_option("sector_min_qubits", 8)
        for k, v in opts: sv.set_option(k, int(v))
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        es, ts = [], []
        for rep in range(6):
            t = time.perf_counter(); e = sv.energy(thetas[rep % 3]); ts.append(1e3 * (time.perf_counter() - t)); es.append(e)
        info = sv.program_info()
        res[sector] = es
        print(f"sector={sector} ms={['%.2f' % t for t in ts]}", {k: v for k, v in info.items() if k.startswith('sector') or k in ('sweeps', 'real_stream')}, flush=True)
if 0 in res:
    d = max(abs(a - b) for a, b in zip(res[0], res[1]))
    print("E", res[1][:3], "max |dE| sector vs dense", d, flush=True)
