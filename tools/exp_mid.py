import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 5)
t = time.time(); ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24); n = 2 * m
R = sum(len(g.terms) for g in gens); G = len(set(ham.packed()[0].tolist()))
print(f"n={n} build {time.time()-t:.1f}s terms={len(ham.terms)} groups={G} gens={len(gens)} rots={R}", flush=True)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
with Statevector(n) as sv:
    t = time.time(); sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf); print(f"upload {time.time()-t:.2f}s", flush=True)
    for rep in range(2):
        t = time.time(); sv.prepare_state(theta); t1 = time.time() - t
        t = time.time(); e = sv.expectation(ham); t2 = time.time() - t
        t = time.time(); e2 = sv.energy(theta); t3 = time.time() - t
        print(f"prepare {t1*1e3:.1f} ms ({32*2**n*len(gens)/t1/1e9:.0f} GB/s per fused sweep; {32*2**n*R/t1/1e9:.0f} GB/s per rotation)  "
              f"expect {t2*1e3:.1f} ms ({16*2**n*G/t2/1e9:.0f} GB/s algorithmic)  energy {t3*1e3:.1f} ms  E={e:.10f} {e2:.10f}", flush=True)
