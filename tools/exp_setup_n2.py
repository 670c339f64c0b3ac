"""time-to-first-energy at 24 qubits (N2 / cc-pVDZ (10e,12o)): set_hamiltonian + set_program + first evaluations, UCCSD and the
reference's QUCCSD gate list; options name=value; sector_debug=4 prints the phases of the table build on stderr"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import quccsd_gate_list
t0 = time.perf_counter()
m = chem.molecule("N2-CCPVDZ"); m.rhf(); P = chem.cas_problem(m, 2, 12)
ham = P.jw_hamiltonian()
size, ops, spin_ops, th, hf = P.uccsd()
print(f"front-end {time.perf_counter() - t0:.2f} s", flush=True)
th = np.array(th)
which = [a for a in sys.argv[1:] if "=" not in a] or ["uccsd", "quccsd"]
for kind in which:
    with Statevector(24) as sv:
        for a in sys.argv[1:]:
            if "=" in a:
                k, v = a.split("="); sv.set_option(k, int(v))
        t = time.perf_counter(); sv.set_hamiltonian(ham); t_h = time.perf_counter() - t
        t = time.perf_counter()
        if kind == "uccsd":
            sv.set_ucc_program(spin_ops, hf)
        else:
            gates, K, _ = quccsd_gate_list(12, 5, 1, excitations=[op.terms[0].qbits for op in ops])
            t = time.perf_counter()
            sv.set_gate_program(gates, K, hf)
        t_p = time.perf_counter() - t
        ts = []
        for rep in range(4):
            t = time.perf_counter(); e = sv.energy(th); ts.append(1e3 * (time.perf_counter() - t))
        print(f"{kind}: set_hamiltonian {1e3 * t_h:.1f} ms, set_program {1e3 * t_p:.1f} ms, evaluations {['%.2f' % x for x in ts]} ms, "
              f"setup (H + program + evaluations up to the first one from the tables) {1e3 * (t_h + t_p) + ts[0] + (ts[1] if ts[1] > 3 * ts[3] else 0):.1f} ms, E = {e:.10f}", flush=True)
