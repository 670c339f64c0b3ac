import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian
mol = chem.molecule("H2O"); mol.rhf(); ham = mol.jw_hamiltonian(); hf = mol.hf_init()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
B = 4096
th = np.random.default_rng(0).uniform(-.1, .1, (B, len(gens)))
def xw(t): return sum(1 for c in t.op if c in "XY")
def run(label, H, G):
    with Statevector(14) as sv:
        sv.set_hamiltonian(H); sv.set_ucc_program(G, hf)
        sv.energy_batch(th[:, :len(G)])
        ms = min((sv.energy_batch(th[:, :len(G)]), sv.last_batch_ms())[1] for _ in range(3))
        print(f"{label:34s} {ms:8.3f} ms / {B} = {B/ms*1e3:10.0f} evals/s  ({ms/16*1e3:6.1f} us per eval per CU)")
diag1 = Hamiltonian(14, [t for t in ham.terms if xw(t) == 0][:1], 0.0, do_clean_up=False)
run("full", ham, gens)
run("rotations (1 diag term)", diag1, gens)
run("floor (1 gen, 1 diag term)", diag1, gens[:1])
run("expectation all (1 gen)", ham, gens[:1])
run("  diag group", Hamiltonian(14, [t for t in ham.terms if xw(t) == 0], 0.0, do_clean_up=False), gens[:1])
run("  weight-2 groups", Hamiltonian(14, [t for t in ham.terms if xw(t) == 2], 0.0, do_clean_up=False), gens[:1])
run("  weight-4 groups", Hamiltonian(14, [t for t in ham.terms if xw(t) == 4], 0.0, do_clean_up=False), gens[:1])
