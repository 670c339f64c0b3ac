"""the reference's QUCCSD gate list on N2 / cc-pVDZ (10e,12o), sector path on the regular (spin-parity) support: a few evaluations
for the profilers (tools/profile_any.sh / profile_pmc_any.sh); options name=value on the command line"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import quccsd_gate_list
m = chem.molecule("N2-CCPVDZ"); m.rhf(); P = chem.cas_problem(m, 2, 12)
ham = P.jw_hamiltonian()
size, ops, _, th, hf = P.uccsd()
gates, K, _ = quccsd_gate_list(12, 5, 1, excitations=[op.terms[0].qbits for op in ops])
th = np.array(th)
reps = 6
grad = 0
with Statevector(24) as sv:
    sv.set_option("sector_profile", 1)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k == "reps":
            reps = int(v)
        elif k == "grad":
            grad = int(v)
        else:
            sv.set_option(k, int(v))
    sv.set_hamiltonian(ham); sv.set_gate_program(gates, K, hf)
    ts, es = [], []
    for rep in range(reps):
        t = time.perf_counter(); es.append(sv.energy(th)); ts.append(1e3 * (time.perf_counter() - t))
    info = sv.program_info()
    print(f"ms={['%.2f' % t for t in ts]} E={es[-1]:.12f} dE={max(es)-min(es):.2e}", {k: v for k, v in info.items() if k.startswith("sector")}, flush=True)
    if grad:
        tg = []
        for rep in range(grad):
            t = time.perf_counter(); e, g = sv.energy_gradient(th); tg.append(1e3 * (time.perf_counter() - t))
        print(f"gradient ms={['%.2f' % t for t in tg]} E={e:.12f} |g|={np.linalg.norm(g):.9f} g[:3]={g[:3]}", flush=True)
