"""pair-table builder, first against second form (sector_pairs_form 1 / 2): same tables — energies, gradients and pair counts of a
20-qubit synthetic molecule must agree exactly"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
ham, gens, hf = fermion.synthetic_molecule(10, 4, seed=3)
th = np.random.default_rng(0).uniform(-0.2, 0.2, len(gens))
out = {}
for f in (1, 2):
    with Statevector(20) as sv:
        sv.set_option("sector_pairs_form", f)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
        e = [sv.energy(th) for _ in range(3)]
        eg, g = sv.energy_gradient(th)
        info = sv.program_info()
        out[f] = (e[-1], eg, g, info["sector_pairs"], info["sector_support"])
print("dE", out[1][0] - out[2][0], "dE(gradient call)", out[1][1] - out[2][1], "max |dg|", np.abs(out[1][2] - out[2][2]).max(),
      "pairs", out[1][3], out[2][3], "support", out[1][4])
