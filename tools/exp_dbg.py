import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian
ham, gens, hf = fermion.synthetic_molecule(7, 5, 1086)
rng = np.random.default_rng(0)
H1 = Hamiltonian(14, [t for t in ham.terms if set(t.op) <= {"Z"}][:1], 0.0, do_clean_up=False)
with Statevector(14) as sv:
    sv.set_hamiltonian(H1); sv.set_ucc_program(gens, hf)
    th = rng.uniform(-.1, .1, (256, len(gens)))
    for dbg in (0, 1):
        sv.set_option("dbg", dbg)
        sv.energy_batch(th)
        ms = min((sv.energy_batch(th), sv.last_batch_ms())[1] for _ in range(5))
        print(f"dbg={dbg} all gens, 1 diag term: {ms*1e3:8.1f} us per eval")
