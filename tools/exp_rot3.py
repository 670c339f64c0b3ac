"""pair-sweep launch geometry, third sweep (per-pivot behaviour, mid sizes)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd.backend import Statevector
variants = [0, 16, 13, 12, 17]
for n in (20, 22, 24, 26, 28):
    with Statevector(n) as sv:
        sv.randomize(1)
        for name, x in (("top", 1 << (n - 1)), ("mid", 1 << (n // 2)), ("b4", 16), ("w4", (1 << (n - 1)) | (1 << (n - 3)) | 32 | 2)):
            row = []
            for v in variants:
                sv.set_option("rot_variant", v)
                t = min(sv.time_pauli_rotation(x, x & 5, 0.1, warmup=2, reps=20) for _ in range(2))
                row.append(32 * 2**n / (t * 1e-3) / 1e9)
            print(f"n={n} {name:4s} " + " ".join(f"v{v}:{g:7.0f}" for v, g in zip(variants, row)))
