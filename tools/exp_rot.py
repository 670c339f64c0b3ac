"""pair-sweep launch geometry at 26-30 qubits, first sweep of variants (round-1 tuning)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import pack_string
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
strings = [("XXXY", [0, 1, 2, 3]), ("XXXY", [n - 4, n - 3, n - 2, n - 1]), ("YXXX", [0, 9, 19, n - 1]), ("X" + "Z" * (n - 2) + "Y", list(range(n)))]
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(0, 13))
with Statevector(n) as sv:
    sv.randomize(1)
    res = {v: [] for v in variants}
    for rnd in range(3):
        for v in variants:
            sv.set_option("rot_variant", v)
            ts = []
            for op, qs in strings:
                x, z = pack_string(n, op, qs)
                ts.append(sv.time_pauli_rotation(x, z, 0.1, warmup=2, reps=10))
            res[v].append(ts)
    for v in variants:
        a = np.array(res[v])
        med = np.median(a, axis=0)
        print(f"variant {v:2d}: " + " ".join(f"{32*2**n/(t*1e-3)/1e9:7.0f}" for t in med) + f"   mean GB/s {np.mean(32*2**n/(med*1e-3)/1e9):7.0f}")
