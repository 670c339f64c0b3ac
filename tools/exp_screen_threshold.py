"""ADAPT gradient screen (ovqe_pool_gradients) over the support list against the whole register, as a function of how much of
the register the state fills: molecule-shaped problems at 2 m qubits, ADAPT-like states of a few exact exponentials.
usage: exp_screen_threshold.py m o"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import GRAD_FERMIONIC, Statevector
m, o = int(sys.argv[1]), int(sys.argv[2])
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=7)
pool = fermion.uccsd_pool_antihermitian(m, o)
rng = np.random.default_rng(3)
picks = rng.permutation(len(pool))[:30]
with Statevector(2 * m) as sv:
    sv.set_hamiltonian(ham)
    ref = None
    for den in (16, 4, 2, 0):
        sv.set_option("screen_sparse", den)
        ts = []
        for rep in range(3):
            sv.init_basis(hf)
            for k in picks: sv.apply_exp_pauli_sum(pool[k], 0.1 + 0.01 * (k % 7))
            sv.norm2()
            t0 = time.perf_counter(); g = sv.pool_gradients(pool, GRAD_FERMIONIC); ts.append(1e3 * (time.perf_counter() - t0))
        if ref is None: ref = g
        print(f"{2*m} qubits, pool {len(pool)}, screen_sparse={den}: screen {min(ts):.2f} ms, amplitudes walked {sv.last_screen_support()} "
              f"of {1 << (2*m)}, max |dg| {np.abs(np.array(g) - np.array(ref)).max():.1e}", flush=True)
