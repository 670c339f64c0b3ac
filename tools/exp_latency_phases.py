"""phases of the latency kernel (k_sparse_vqe_wg) for one H2O evaluation: needs the testing build (OVQE_LIB=testing), option sparse_dbg=9"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2); K = len(gens)
th = np.random.default_rng(K).uniform(-0.1, 0.1, (200, K))
with Statevector(ham.nbqbits) as sv:
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, mol.hf_init())
    for k in range(20): sv.energy(th[k])
    t = time.perf_counter()
    for k in range(200): sv.energy(th[k])
    print(f"{(time.perf_counter() - t) / 200 * 1e6:.1f} us per call", flush=True)
    sv.set_option("sparse_dbg", 9)
    for k in range(5): sv.energy(th[k])
    sv.set_option("sparse_dbg", 0)
    print(sv.program_info())
