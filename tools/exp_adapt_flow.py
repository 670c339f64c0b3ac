"""configs[2]: fermionic ADAPT-VQE on H2O/STO-3G (14 qubits, spin_complement_gsd pool of 1246 operators) through the
L1 mirror on the GPU: wall time per macro-iteration.  `python tools/exp_adapt_flow.py [iterations]`"""
import sys, os, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.adapt.fermionic_adapt_vqe import fermionic_adapt_vqe

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
t = time.time()
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian(); hf = mol.hf_init()
size, pool = pools.spin_complement_gsd(mol.n_elec, mol.nao)
print(f"front-end {time.time()-t:.1f} s; pool {size}; E_HF {mol.e_hf:.8f}", flush=True)
t = time.time()
buf = io.StringIO()
import cProfile, pstats
prof = cProfile.Profile() if "--profile" in sys.argv else None
if prof: prof.enable()
with contextlib.redirect_stdout(buf):
    iterations, result = fermionic_adapt_vqe(None, None, None, ham, pool, hf, n_max_grads=1, fci=0.0,
                                             optimizer="COBYLA", tolerance=1e-6, type_conver="norm",
                                             threshold_needed=1e-2, max_external_iterations=iters)
dt = time.time() - t
if prof:
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
print(f"{len(iterations['energies'])} ADAPT iterations in {dt:.2f} s  ({dt/len(iterations['energies']):.2f} s each)")
print("indices", result.get("indices"))
print("energies", [round(e, 8) for e in iterations["energies"]])
print("norms", [round(e, 6) for e in iterations["norms"]])
