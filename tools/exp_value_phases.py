"""the headline kernel (k_sparse_vqe_rows<2>, H2O/STO-3G UCCSD, 65 536 evaluations per launch) without one of its phases
("sparse_dbg": 1 no sincos, 2 no circuit rows, 3 no Hamiltonian entries): where its time goes.  python tools/exp_value_phases.py [name=value]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian(); hf = mol.hf_init()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
rng = np.random.default_rng(0)
B = 65536
th = rng.uniform(-.1, .1, (B, len(gens)))
with Statevector(ham.nbqbits) as sv:
    for a in sys.argv[1:]:
        k, v = a.split("="); sv.set_option(k, int(v))
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    for dbg, label in ((0, "whole kernel"), (1, "no sincos"), (2, "no circuit rows"), (3, "no Hamiltonian entries")):
        sv.set_option("sparse_dbg", dbg)
        sv.energy_batch(th)
        ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(5))
        print(f"{label:36s} {ms:7.3f} ms per {B} evaluations -> {B / ms * 1e3 / 1e6:6.1f} M evaluations/s")
    info = sv.program_info()
    print({k: info[k] for k in info if k.startswith("sparse") or k in ("rotations",)})
