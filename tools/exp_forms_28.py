"""the automatic selection at 28 qubits (14 orbitals, 7 + 7 electrons: support 3432^2 = 11.8 M): sector-path energy and gradient against the
dense-state kernels of the same handle, and the forms that served (pair streams beyond 16 M words per sweep: the first sweep form)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
m, o = 14, 7
t0 = time.perf_counter()
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=m)
print(f"{2*m} qubits: {len(gens)} generators, {len(ham.terms)} Hamiltonian terms ({time.perf_counter()-t0:.1f} s)", flush=True)
rng = np.random.default_rng(m)
th = rng.uniform(-0.1, 0.1, len(gens))
res = {}
for sector in (1, 0):
    with Statevector(2 * m) as sv:
        sv.set_option("sector", sector)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        t0 = time.perf_counter()
        es = []
        for _ in range(3):
            t1 = time.perf_counter(); es.append(sv.energy(th)); print(f"  sector={sector} energy {es[-1]:.12f} in {time.perf_counter()-t1:.2f} s", flush=True)
        f_e = sorted(sv.sector_forms())
        g = None
        if sector:
            t1 = time.perf_counter(); eg, g = sv.energy_gradient(th); print(f"  gradient in {time.perf_counter()-t1:.2f} s, |g|max {np.abs(g).max():.3e}, E {eg:.12f}", flush=True)
        info = sv.program_info()
        res[sector] = (es, f_e, sorted(sv.sector_forms()), info)
        print(f"sector={sector}: support={info['sector_support']} sweeps={info['sector_sweeps']} pairs={info['sector_pairs']} h_elements={info['sector_h_elements']} "
              f"bytes={info['sector_bytes']} forms(energy)={f_e} forms(all)={sorted(sv.sector_forms())}", flush=True)
l1 = float(np.abs(ham.packed()[2]).sum())
print("max |E_sector - E_dense| / |H|_1 =", max(abs(a - b) for a, b in zip(res[1][0], res[0][0])) / l1)
