"""per-call times of the QUCCSD mirror's evaluator on N2 (what get_energies' optimiser calls)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
size, cluster_ops, _, theta_mp2, hf = prob.uccsd()
eng = EnergyUCC()
ev = eng._evaluator(ham, cluster_ops, hf)
th = np.array(theta_mp2)
for r in range(6):
    t = time.perf_counter(); e, g = ev.energy_gradient(th + 1e-3 * r); print(f"gradient call {r}: {1e3 * (time.perf_counter() - t):.1f} ms  E={e:.10f}")
for r in range(3):
    t = time.perf_counter(); e = ev.energy(th + 1e-3 * r); print(f"energy call {r}: {1e3 * (time.perf_counter() - t):.1f} ms  E={e:.10f}")
print({k: v for k, v in ev.sv.program_info().items() if k.startswith("sector") or k in ("rotations", "literal_gates", "ops")})
# idle gaps between calls (what an optimiser's host work leaves): does the device clock down?
for gap_ms in (0, 2, 5, 10, 20, 50):
    ts = []
    for r in range(6):
        time.sleep(gap_ms * 1e-3)
        t = time.perf_counter(); ev.energy_gradient(th + 1e-3 * r); ts.append(1e3 * (time.perf_counter() - t))
    print(f"idle {gap_ms} ms between calls: gradient calls {['%.1f' % x for x in ts]} ms")
# the same gaps filled with host BLAS work (numpy matrix-vector products of the BFGS size)
H = np.eye(1715); v = np.ones(1715)
for reps in (1, 10, 40):
    ts = []
    for r in range(6):
        t0 = time.perf_counter()
        for _ in range(reps): w = H @ v; H += 1e-9 * np.outer(w, v)
        gap = 1e3 * (time.perf_counter() - t0)
        t = time.perf_counter(); ev.energy_gradient(th + 1e-3 * r); ts.append(1e3 * (time.perf_counter() - t))
    print(f"{gap:.1f} ms of numpy between calls: gradient calls {['%.1f' % x for x in ts]} ms")
