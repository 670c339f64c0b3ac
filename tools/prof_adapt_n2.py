"""cProfile of the 30-iteration fermionic-ADAPT mirror on N2 / cc-pVDZ (10e,12o) (the run of tools/exp_adapt_n2.py): where the host time goes"""
import os, sys, io, contextlib, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openvqe_amd import chem, pools
from openvqe_amd.adapt import fermionic_adapt_vqe as fav
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, pool = pools.singlet_sd(10, 12)
fav.SECTOR_GROUND_SPACE = True
fav._FLAVOUR.optimiser_display = False
pr = cProfile.Profile()
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable()
    fav.fermionic_adapt_vqe(None, None, None, ham, pool, hf, 1, -109.0745445341, "COBYLA", 1e-6, "norm", 1e-3, 30)
    pr.disable()
print("wall", time.perf_counter() - t0)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
