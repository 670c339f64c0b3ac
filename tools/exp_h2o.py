"""H2O/STO-3G UCCSD on the fused kernels: thread counts / options sweep (round-1 tuning)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
mol = chem.molecule(sys.argv[1] if len(sys.argv) > 1 else "H2O"); mol.rhf(); ham = mol.jw_hamiltonian(); hf = mol.hf_init()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
n = ham.nbqbits
rng = np.random.default_rng(0)
with Statevector(n) as sv:
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    ref = None
    for path, label in ((1, "dense LDS kernel"), (3, "support-compacted kernel")):
        sv.set_option("force_path", path)
        for B in (1, 141, 4096, 65536):
            for spw in ((0,) if path == 1 or B < 4096 else (1, 2, 4)):
                sv.set_option("sparse_spw", spw)
                th = rng.uniform(-.1, .1, (B, len(gens)))
                e = sv.energy_batch(th)
                ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(3))
                print(f"{label:26s} B={B:6d} spw={spw} {ms:9.3f} ms -> {B/ms*1e3:12.0f} evals/s")
    th = rng.uniform(-.2, .2, (64, len(gens)))
    sv.set_option("force_path", 1); a = sv.energy_batch(th)
    sv.set_option("force_path", 3); b = sv.energy_batch(th)
    print("max |dense - sparse| =", np.abs(a - b).max())
