"""CHECKER (imports oracle/: lives under tests/, not a test pytest collects).  The energy bench.py keeps on record for configs[4]'s strong leg (bench.SHARDED_KNOWN) from the C ORACLE, on the host: the synthetic
state of 2^n amplitudes (openvqe_amd/synth.py, generated in chunks), bench.sharded_workload's rotations one fused mask sweep each
(oracle/c orc_pauli_rotation) and <H> x-group by x-group (orc_expectation_grouped).  n = 31: a 32-GiB host state, about ten minutes on 16
cores.  usage: python tests/oracle_sharded_energy.py [n] [rotations] [terms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from openvqe_amd import synth
from oracle import cref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L = cref.lib()
t0 = time.perf_counter()
psi = np.empty(1 << n, np.complex128)
step = 1 << min(n, 24)
norm2 = 0.0
for lo in range(0, 1 << n, step):
    chunk = synth.amplitudes(bench.SHARDED_SEED, np.arange(lo, lo + step, dtype=np.uint64))
    psi[lo:lo + step] = chunk
    norm2 += float(np.vdot(chunk, chunk).real)
for lo in range(0, 1 << n, step):
    psi[lo:lo + step] *= 1.0 / np.sqrt(norm2)
print(f"state of 2^{n} amplitudes on the host: {time.perf_counter() - t0:.1f} s, |psi|^2 before scaling {norm2:.6e}", flush=True)
xs, zs, phis, hx, hz, hc = bench.sharded_workload(n, R, T)
t0 = time.perf_counter()
for x, z, p in zip(xs, zs, phis):
    L.orc_pauli_rotation(psi, n, int(x), int(z), float(p))
print(f"{R} rotations: {time.perf_counter() - t0:.1f} s ({cref.usable_cpus()} threads)", flush=True)
sx, sz, sc = cref.sort_by_x(np.array(hx, np.uint64), np.array(hz, np.uint64), np.array(hc, np.float64))
t0 = time.perf_counter()
e = L.orc_expectation_grouped(psi, n, len(sx), sx, sz, sc)
print(f"<H> over {len(set(sx.tolist()))} x-groups: {time.perf_counter() - t0:.1f} s", flush=True)
l1 = float(np.abs(sc).sum())
known = bench.SHARDED_KNOWN.get((n, R, T))
print(f"oracle energy {e!r}  |H|_1 {l1:.3f}" + (f"  on record {known!r}  difference {abs(e - known):.3e} = {abs(e - known) / l1:.2e} |H|_1" if known is not None else ""))
