"""H2O QUCCSD gate list in Clifford-frame form: dense LDS kernel vs support-compacted kernel"""
import sys, os, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import bench
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian()
gates, K, hf = bench._quccsd_gates(mol.nao, mol.n_elec // 2)
th = np.random.default_rng(5).uniform(-0.1, 0.1, (8192, K))
with Statevector(14) as sv:
    sv.set_hamiltonian(ham); sv.set_gate_program(gates, K, hf)
    for fp in (0, 1, 3):
        sv.set_option("force_path", fp)
        e = sv.energy_batch(th)
        t = time.perf_counter(); e = sv.energy_batch(th); dt = time.perf_counter() - t
        print("force_path", fp, f"{len(th)/dt:,.0f} evals/s", e[0], sv.program_info()["support"], flush=True)
