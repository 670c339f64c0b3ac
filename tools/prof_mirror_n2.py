"""cProfile of the QUCCSD get_energies mirror on N2 (tools/exp_mirror_n2.py's run): where the wall time of the entry point goes"""
import os, sys, time, io, contextlib, cProfile, pstats
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
pr = cProfile.Profile()
buf = io.StringIO()
t0 = time.perf_counter()
pr.enable()
with contextlib.redirect_stdout(buf):
    iterations, result = eng.get_energies(ham, cluster_ops, hf, list(theta_mp2), [0.01] * size, -109.0745445341)
pr.disable()
print("wall", time.perf_counter() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
