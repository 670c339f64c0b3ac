"""which QUCCSD templates of N2/cc-pVDZ (10e,12o) in the reference's operator order do not close their Clifford frame"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import quccsd_gate_list
m = chem.molecule("N2-CCPVDZ"); m.rhf(); P = chem.cas_problem(m, 2, 12)
size, ops, _, th, hf = P.uccsd()
exci = [op.terms[0].qbits for op in ops]
bad = []
with Statevector(24) as sv:
    for k, e in enumerate(exci):
        gates, K, _ = quccsd_gate_list(12, 5, 1, excitations=[e])
        sv.set_gate_program(gates, K, hf)
        if sv.program_info()["literal_gates"]:
            bad.append((k, e))
print(len(bad), bad[:20])
