"""tiled <H> and H psi on a 2*m-qubit random state for the first `terms` strings of a molecule-shaped Hamiltonian: time per call for
tile-cover settings given as --opt=name=value (e.g. ham_tile_low).  python tools/exp_hexp_big.py 15 2000 --opt=ham_tile_low=4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, nterms = int(args[0]), int(args[1])
opts = [a[6:].split("=") for a in sys.argv if a.startswith("--opt=")]
ham, _, _ = fermion.synthetic_molecule(m, 5, seed=24)
ham = Hamiltonian(2 * m, ham.terms[:nterms], do_clean_up=False)
with Statevector(2 * m) as sv:
    for k, v in opts: sv.set_option(k, int(v))
    sv.randomize(7, 1.0)
    sv.set_hamiltonian(ham)
    ts = []
    for rep in range(3):
        t = time.perf_counter(); e = sv.expectation(ham); ts.append(time.perf_counter() - t)
    print(f"{2*m} qubits, {nterms} strings, {opts}: <H> = {e:.10f}, ms per call {[round(1e3*t,1) for t in ts]}", flush=True)
    t = time.perf_counter(); eg, r, it = sv.ground_state(tol=1e-3, max_iter=6); dt = time.perf_counter() - t
    print(f"   6 Lanczos steps (two passes): {dt:.2f}s -> {dt/12*1e3:.1f} ms per H psi (+ dots)", flush=True)
