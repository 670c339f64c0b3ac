"""evaluation cost of an ADAPT-sized ansatz at 24 qubits (N2 / cc-pVDZ (10e,12o)): K spin-adapted singlet generators (Trotterised like
ucc_action), energies through the library's automatic dispatch, program_info with the sector profile.  python tools/exp_adapt_eval_n2.py [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.backend import Statevector
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, singlets = pools.singlet_sd(10, 12)
rng = np.random.default_rng(3)
order = rng.permutation(len(singlets))
for K in [int(a) for a in sys.argv[1:]] or [8, 16, 28]:
    gens = [1j * singlets[k] for k in order[:K]]
    with Statevector(24) as sv:
        sv.set_option("sector_profile", 1)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        theta = rng.uniform(-0.2, 0.2, K)
        ts = []
        for rep in range(8):
            t = time.perf_counter(); e = sv.energy(theta + 0.01 * rep); ts.append(1e3 * (time.perf_counter() - t))
        info = sv.program_info()
        e_sector, g_sector = sv.energy_gradient(theta)
        sv.set_option("sector", 0)
        e_dense = sv.energy(theta)
        print(f"   E sector path {e_sector:.12f}  dense kernels {e_dense:.12f}  diff {abs(e_sector - e_dense):.2e}; max |g| {np.abs(g_sector).max():.4f}")
        print(f"K={K}: ms {[round(t, 2) for t in ts]}", {k: info[k] for k in ("rotations", "support", "sector_support", "sector_sweeps", "sector_pairs", "sector_h_sweeps", "sector_h_elements", "sector_bytes", "sector_circuit_us", "sector_expect_us") if k in info}, flush=True)
