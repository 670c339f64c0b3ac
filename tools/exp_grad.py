"""exact adjoint gradient vs forward differences at 2*m qubits (timing helper): `python tools/exp_grad.py 10 5 [every]`"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 5)
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
gens = gens[::every]
n, K = 2 * m, len(gens)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, K)
with Statevector(n) as sv:
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    sv.energy(theta)
    t = time.time(); e0 = sv.energy(theta); te = time.time() - t
    sv.energy_gradient(theta)
    t = time.time(); e, g = sv.energy_gradient(theta); tg = time.time() - t
    eps = 1e-6
    ks = list(range(0, K, max(1, K // 8)))
    fd = []
    for k in ks:
        tp = theta.copy(); tp[k] += eps; tm = theta.copy(); tm[k] -= eps
        fd.append((sv.energy(tp) - sv.energy(tm)) / (2 * eps))
    print(f"n={n} K={K} rotations={sum(len(x.terms) for x in gens)}: energy {te*1e3:.2f} ms; adjoint gradient (all {K}) {tg*1e3:.1f} ms "
          f"= {tg/te:.1f} evaluations; forward differences would take {(K+1)*te*1e3:.0f} ms ({(K+1)*te/tg:.0f}x); "
          f"|E-E0|={abs(e-e0):.1e}; max|g-fd| on {len(ks)} samples = {np.abs(g[ks]-np.array(fd)).max():.1e}", flush=True)
