"""the H2O `get_energies` mirror three times over (scipy's own Jacobian / batched forward differences / exact Jacobian): wall time per
run, to see the spread between repetitions"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
mol = chem.molecule("H2O"); mol.rhf()
prob = mol.problem(active=False)
ham = prob.jw_hamiltonian()
size, _, spin_ops, theta_mp2, hf = prob.uccsd()
e_fci = mol.ci_ground_state()[0]
for rep in range(3):
    for label, flags in (("default", {}), ("batched", {"batched_gradient": True}), ("adjoint", {"adjoint_gradient": True})):
        u = EnergyUCC()
        for k, v in flags.items(): setattr(u, k, v)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            it, res = u.get_energies(ham, spin_ops, spin_ops, hf, list(theta_mp2), [0.0] * size, e_fci)
        print(rep, label, round(time.perf_counter() - t0, 3), len(res["energies_1"]) + len(res["energies_2"]))
