"""diagonal-string sweeps at 30 qubits: launch geometry experiments (round-1 tuning)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd.backend import Statevector
n = 30
with Statevector(n) as sv:
    sv.randomize(1)
    for v in (0, 100, 101, 102, 103, 104, 105, 106, 107):
        sv.set_option("rot_variant", v)
        ts = [min(sv.time_pauli_rotation(0, z, 0.1, warmup=1, reps=6) for _ in range(2)) for z in ((1 << 30) - 1, 1 << 15)]
        print(f"diag variant {v:3d}: " + " ".join(f"{32*2**n/(t*1e-3)/1e9:7.0f}" for t in ts))
