import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian, Term
ham, gens, hf = fermion.synthetic_molecule(7, 5, 1086)
rng = np.random.default_rng(0)
diag = [t for t in ham.terms if set(t.op) <= {"Z"}]
w4 = [t for t in ham.terms if sum(c in "XY" for c in t.op) == 4]
cases = [("0 gens, 1 diag term", [diag[0]], []), ("0 gens, all diag", diag, []), ("0 gens, 1 w4 term", [w4[0]], []),
         ("1 gen, 1 diag term", [diag[0]], gens[20:21]), ("1 single gen, 1 diag term", [diag[0]], gens[:1]),
         ("8 gens, 1 diag term", [diag[0]], gens[20:28])]
for n in (14, 12):
    for label, terms, G in cases:
        if n == 12:
            continue
        with Statevector(n) as sv:
            sv.set_hamiltonian(Hamiltonian(n, terms, 0.0, do_clean_up=False)); sv.set_ucc_program(G, hf)
            for B in (256,):
                th = rng.uniform(-.1, .1, (B, max(1, len(G))))[:, :len(G)]
                sv.energy_batch(th)
                ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(5))
                print(f"n={n} {label:28s} B={B:5d} {ms*1e3:9.1f} us")
