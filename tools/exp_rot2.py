"""pair-sweep launch geometry, second sweep (threads per group, non-temporal accesses)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd.backend import Statevector
n = 30
variants = [int(v) for v in sys.argv[1].split(",")]
cases = [("bit%d" % p, 1 << p) for p in (0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 29)]
cases += [("b29+b0", (1 << 29) | 1), ("b29+b3", (1 << 29) | 8), ("b29..26", 0xF << 26), ("b3..0", 0xF), ("b20+b5", (1 << 20) | 32),
          ("b13..10", 0xF << 10), ("b7..4", 0xF0)]
with Statevector(n) as sv:
    sv.randomize(1)
    print("case      " + " ".join(f"v{v:<5d}" for v in variants))
    tot = {v: [] for v in variants}
    for name, x in cases:
        row = []
        for v in variants:
            sv.set_option("rot_variant", v)
            t = min(sv.time_pauli_rotation(x, x & 0x5, 0.1, warmup=1, reps=6) for _ in range(2))
            g = 32 * 2**n / (t * 1e-3) / 1e9
            row.append(g); tot[v].append(g)
        print(f"{name:9s} " + " ".join(f"{g:6.0f}" for g in row))
    print("mean      " + " ".join(f"{np.mean(tot[v]):6.0f}" for v in variants))
    print("min       " + " ".join(f"{np.min(tot[v]):6.0f}" for v in variants))
