"""QUCCSD gate list (reference templates on all UCCSD excitations of an (m spatial, o occupied) register): literal
execution vs Clifford-frame form.  `python tools/exp_frame.py 7 5` = H2O/STO-3G-sized (14 qubits)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import efficient_fermionic_ansatz
from openvqe_amd.qat_compat import AffineParam, Program, lower_circuit

m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7, 5)
nexc = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
n = 2 * m
singles, doubles = fermion.uccsd_excitations(m, o)
exci = [[i, a] for i, a in singles] + [[i, j, a, b] for i, j, a, b in doubles]
exci = exci[::max(1, len(exci) // nexc)]
K = len(exci)
prog = Program(); reg = prog.qalloc(n)
efficient_fermionic_ansatz(reg, prog, exci, [AffineParam(k) for k in range(K)])
_, kind, gates = lower_circuit(prog.to_circ())
hf = fermion.hf_integer(n, 2 * o)
ham, _, _ = fermion.synthetic_molecule(m, o, seed=24) if n <= 16 else (None, None, None)
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, K)
thetas = rng.uniform(-0.1, 0.1, (B, K))
print(f"n={n} excitations={K} literal gates={len(gates)}", flush=True)
with Statevector(n) as sv:
    if ham is not None: sv.set_hamiltonian(ham)
    ref = None
    for mode in (0, 1):
        sv.set_option("clifford_frame", mode)
        t = time.time(); sv.set_gate_program(gates, K, hf); tc = time.time() - t
        info = sv.program_info()
        sv.prepare_state(theta)
        t = time.time(); sv.prepare_state(theta); tp = time.time() - t
        line = f"clifford_frame={mode}: compile {tc*1e3:.0f} ms  {info}  prepare_state {tp*1e3:.2f} ms"
        if ham is not None:
            e = sv.energy(theta)
            t = time.time(); e = sv.energy(theta); te = time.time() - t
            nb = B if mode else max(8, B // 64)
            sv.energy_batch(thetas[:nb])
            t = time.time(); eb = sv.energy_batch(thetas[:nb]); tb = time.time() - t
            line += f"  energy {te*1e3:.3f} ms  batch {nb}: {nb/tb:,.0f} evals/s  E={e:.12f} {sv.program_info()['support']}"
            if ref is None: ref = eb
            else: line += f"  max|dE vs literal|={np.abs(eb[:len(ref)]-ref).max():.1e}"
        print(line, flush=True)
