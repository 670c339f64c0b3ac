"""scratch experiment: where does the 14-qubit fused kernel spend its time?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian

ham, gens, hf = fermion.synthetic_molecule(7, 5, 1086)
K = len(gens)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
th = np.random.default_rng(0).uniform(-.1, .1, (B, K))
def run(label, H, G, reps=3):
    with Statevector(14) as sv:
        sv.set_hamiltonian(H); sv.set_ucc_program(G, hf)
        sv.energy_batch(th[:, :len(G)])
        ms = []
        for _ in range(reps):
            sv.energy_batch(th[:, :len(G)]); ms.append(sv.last_batch_ms())
        print(f"{label:40s} {min(ms):9.3f} ms / {B} evals = {B/min(ms)*1e3:10.0f} evals/s")
run("full (1000 rot, 3381 terms)", ham, gens)
run("rotations only (10 H terms)", Hamiltonian(14, ham.terms[:10], 0.0, do_clean_up=False), gens)
run("expectation only (1 generator)", ham, gens[:1])
run("diag group only", Hamiltonian(14, [t for t in ham.terms if set(t.op) <= {"Z"}], 0.0, do_clean_up=False), gens[:1])
def xw(t):
    return sum(1 for c in t.op if c in "XY")
run("weight-2 groups only", Hamiltonian(14, [t for t in ham.terms if xw(t) == 2], 0.0, do_clean_up=False), gens[:1])
run("weight-4 groups only", Hamiltonian(14, [t for t in ham.terms if xw(t) == 4], 0.0, do_clean_up=False), gens[:1])
w4 = [t for t in ham.terms if xw(t) == 4]
run("weight-4, first 400 terms", Hamiltonian(14, w4[:400], 0.0, do_clean_up=False), gens[:1])
run("floor: 1 generator, 10 H terms", Hamiltonian(14, ham.terms[:10], 0.0, do_clean_up=False), gens[:1])
run("20 singles only, 10 H terms", Hamiltonian(14, ham.terms[:10], 0.0, do_clean_up=False), gens[:20])
run("first 60 doubles only, 10 H terms", Hamiltonian(14, ham.terms[:10], 0.0, do_clean_up=False), gens[20:80])
