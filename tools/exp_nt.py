import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian
ham, gens, hf = fermion.synthetic_molecule(7, 5, 1086)
rng = np.random.default_rng(0)
H1 = Hamiltonian(14, [t for t in ham.terms if set(t.op) <= {"Z"}][:1], 0.0, do_clean_up=False)
th = rng.uniform(-.1, .1, (2048, len(gens)))
for label, H in (("rot only", H1), ("full", ham)):
    with Statevector(14) as sv:
        sv.set_hamiltonian(H); sv.set_ucc_program(gens, hf)
        ref = None
        for nt in (1024, 512, 256):
            sv.set_option("small_threads", nt)
            e = sv.energy_batch(th)
            ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(3))
            if ref is None: ref = e
            print(f"{label:9s} NT={nt:5d} {ms:8.3f} ms / 2048 -> {2048/ms*1e3:10.0f} evals/s  maxdiff {np.abs(e-ref).max():.2e}")
