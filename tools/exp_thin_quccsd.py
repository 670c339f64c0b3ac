import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import quccsd_gate_list
m = chem.molecule("N2-CCPVDZ"); m.rhf(); P = chem.cas_problem(m, 2, 12)
ham = P.jw_hamiltonian()
size, ops, _, th, hf = P.uccsd()
gates, K, _ = quccsd_gate_list(12, 5, 5, excitations=[op.terms[0].qbits for op in ops])
th = np.array(th[::5]) + 0.01
with Statevector(24) as sv:
    sv.set_option("sector_debug", 2)
    sv.set_hamiltonian(ham); sv.set_gate_program(gates, K, hf)
    for r in range(3): print(sv.energy(th))
    print({k: v for k, v in sv.program_info().items() if k.startswith("sector")})
