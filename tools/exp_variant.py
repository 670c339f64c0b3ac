"""30-qubit single-string sweeps: launch geometry variants of the pair sweep (rot_variant option)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import pack_string
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
variants = [int(v) for v in sys.argv[2:]] or [0, 16, 19, 21]
with Statevector(n) as sv:
    sv.randomize(1)
    tot = {v: [] for v in variants}
    for name, (op, qs) in bench.m1_strings(n):
        x, z = pack_string(n, op, qs)
        if x == 0: continue
        row = []
        for v in variants:
            sv.set_option("rot_variant", v)
            ms = sv.time_pauli_rotation(x, z, 0.1, warmup=2, reps=10)
            g = 32.0 * (1 << n) / (ms * 1e-3) / 1e9
            row.append(g); tot[v].append(ms)
        print(f"{name:12s} " + "  ".join(f"v{v}: {g:6.0f}" for v, g in zip(variants, row)), flush=True)
    for v in variants:
        m = sum(tot[v]) / len(tot[v])
        print(f"variant {v}: mean {m:.3f} ms = {32.0*(1<<n)/(m*1e-3)/1e9:.0f} GB/s")
