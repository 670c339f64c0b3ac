"""the reference's QUCCSD templates on all UCCSD excitations of a synthetic (m spatial orbitals, o occupied) molecule — 2 m qubits: the
sector path on the regular spin-parity support (bit-arithmetic sweeps forwards and backwards) against the dense-state kernels of the
same handle; `python tools/exp_quccsd_synth.py 13 3` = 26 qubits"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import quccsd_gate_list
m, o = int(sys.argv[1]), int(sys.argv[2])
n = 2 * m
ham, _, hf = fermion.synthetic_molecule(m, o, seed=77)
gates, K, hf2 = quccsd_gate_list(m, o, 1)
assert hf2 == hf
rng = np.random.default_rng(5)
th = rng.uniform(-0.2, 0.2, K)
print(f"{n} qubits, {K} parameters, {len(gates)} gates, {len(ham.terms)} Hamiltonian terms", flush=True)
with Statevector(n) as sv:
    for a in sys.argv[3:]:
        k, v = a.split("="); sv.set_option(k, int(v))
    sv.set_hamiltonian(ham); sv.set_gate_program(gates, K, hf)
    ts = []
    for r in range(5):
        t = time.perf_counter(); e = sv.energy(th); ts.append(1e3 * (time.perf_counter() - t))
    info = sv.program_info()
    tg = []
    for r in range(3):
        t = time.perf_counter(); eg, g = sv.energy_gradient(th); tg.append(1e3 * (time.perf_counter() - t))
    print(f"sector path: evaluations {['%.2f' % x for x in ts]} ms, gradients {['%.1f' % x for x in tg]} ms, E = {e:.12f}",
          {k: v for k, v in info.items() if k.startswith("sector") and v}, flush=True)
    sv.set_option("sector", 0)
    t = time.perf_counter(); ed = sv.energy(th); td = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); ed = sv.energy(th); td = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter(); egd, gd = sv.energy_gradient(th); tgd = 1e3 * (time.perf_counter() - t)
    l1 = float(np.abs(ham.packed()[2]).sum())
    print(f"dense states: evaluation {td:.1f} ms, gradient {tgd:.0f} ms; |dE| = {abs(e - ed):.2e}, |dE(grad call)| = {abs(eg - egd):.2e}, "
          f"max |dg| = {np.abs(g - gd).max():.2e}  (|H|_1 = {l1:.1f})")
