import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian
ham, gens, hf = fermion.synthetic_molecule(7, 5, 1086)
K = len(gens)
rng = np.random.default_rng(0)
for label, H, G in (("floor", Hamiltonian(14, ham.terms[:10], 0.0, do_clean_up=False), gens[:1]), ("full", ham, gens)):
    with Statevector(14) as sv:
        sv.set_hamiltonian(H); sv.set_ucc_program(G, hf)
        for B in (1, 64, 256, 1024, 4096, 16384):
            th = rng.uniform(-.1, .1, (B, len(G)))
            sv.energy_batch(th)
            ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(3))
            print(f"{label:6s} B={B:6d} {ms:9.3f} ms  {B/ms*1e3:12.0f} evals/s   per-eval-per-CU {ms/ max(1,B/256)*1e3:8.1f} us")
