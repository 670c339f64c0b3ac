"""latency outliers of single evaluations at 24 qubits: an ADAPT-sized program (the first generators of the N2 singles-and-doubles
list) evaluated many times with small random angles; prints the evaluations that took more than 5 ms and their positions.
arguments: [generators=30] [evaluations=20000] [name=value options]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
pos = [a for a in sys.argv[1:] if "=" not in a]
ngen = int(pos[0]) if pos else 30
nev = int(pos[1]) if len(pos) > 1 else 20000
m = chem.molecule("N2-CCPVDZ"); m.rhf(); P = chem.cas_problem(m, 2, 12)
ham = P.jw_hamiltonian()
size, ops, spin_ops, th, hf = P.uccsd()
rng = np.random.default_rng(5)
with Statevector(24) as sv:
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("="); sv.set_option(k, int(v))
    sv.set_hamiltonian(ham)
    pick = list(range(len(spin_ops) - ngen, len(spin_ops)))   # doubles
    sv.set_ucc_program([spin_ops[i] for i in pick], hf)
    ts = np.empty(nev)
    for i in range(nev):
        x = 0.05 * rng.standard_normal(ngen)
        t = time.perf_counter(); sv.energy(x); ts[i] = time.perf_counter() - t
    slow = np.nonzero(ts > 5e-3)[0]
    print(f"{nev} evaluations: median {1e3 * np.median(ts):.3f} ms, total {ts.sum():.2f} s; above 5 ms: {len(slow)} totalling {ts[slow].sum():.2f} s")
    print("  positions / ms:", [(int(i), round(1e3 * float(ts[i]), 1)) for i in slow[:40]])
    info = sv.program_info()
    print("  support", info.get("sector_support"))
