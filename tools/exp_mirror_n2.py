"""the reference's QUCCSD entry point on configs[3]: openvqe_amd.ucc_family.get_energy_qucc.EnergyUCC.get_energies (mirror of
ref:openvqe/ucc_family/get_energy_qucc.py:136-244: BFGS, tol 1e-5, from the MP2 guess and from the constant guess) on
N2 / cc-pVDZ (10e,12o) = 24 qubits, 1715 cluster operators, with the opt-in exact Jacobian"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
mol = chem.molecule("N2-CCPVDZ"); e_rhf = mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
size, cluster_ops, _, theta_mp2, hf = prob.uccsd()
EnergyUCC.adjoint_gradient = True
eng = EnergyUCC()
t0 = time.perf_counter()
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    iterations, result = eng.get_energies(ham, cluster_ops, hf, list(theta_mp2), [0.01] * size, -109.0745445341)
wall = time.perf_counter() - t0
print(buf.getvalue()[-600:])
print(f"E_RHF={e_rhf:.10f} E1={iterations['minimum_energy_result1_guess'][0]:.10f} E2={iterations['minimum_energy_result2_guess'][0]:.10f} "
      f"CNOT={result['CNOT1']} evaluations={len(result['energies_1'])}+{len(result['energies_2'])} wall={wall:.1f}s")
