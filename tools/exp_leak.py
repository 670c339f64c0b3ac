import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.backend import GRAD_FERMIONIC, Statevector
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, singlets = pools.singlet_sd(10, 12)
nops = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sv = Statevector(24)
sv.set_hamiltonian(ham)
sv.init_basis(hf)
rng = np.random.default_rng(0)
for k in rng.choice(len(singlets), nops, replace=False):
    sv.apply_exp_pauli_sum(singlets[k], 0.2)
psi = sv.get_state()
nz = np.nonzero(psi)[0]
a = np.abs(psi[nz])
print("nonzero", len(nz), "above 1e-14", int((a > 1e-14).sum()), "above 1e-10", int((a > 1e-10).sum()), "min nonzero", a.min())
pc = np.array([bin(int(i)).count("1") for i in nz[:200000]])
print("popcounts of non-zero indices:", dict(zip(*np.unique(pc, return_counts=True))))
even = np.array([bin(int(i) & 0xAAAAAA).count("1") for i in nz[:200000]])
print("alpha counts:", dict(zip(*np.unique(even, return_counts=True))))
small = nz[a < 1e-14][:5]
print("examples of tiny amplitudes:", [(int(i), bin(int(i)).count('1'), psi[i]) for i in small])
