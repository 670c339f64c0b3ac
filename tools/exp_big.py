"""32- / 33-qubit single-GPU states (64 / 128 GiB): sweep time, inverse and norm (property check + timing)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import synth
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import pack_string
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(n)
idx = rng.integers(0, 1 << n, 2048).astype(np.uint64)
with Statevector(n) as sv:
    scale = sv.randomize(20250227)
    base = sv.get_amplitudes(idx)
    assert np.array_equal(base, synth.amplitudes(20250227, idx) * scale)
    print(f"n={n}: norm {sv.norm2():.12f}", flush=True)
    for op, qs in (("XXXY", [0, 9, 19, n - 1]), ("X" + "Z" * (n - 2) + "Y", list(range(n))), ("Z" * n, list(range(n))), ("XYXX", [n - 4, n - 3, n - 2, n - 1])):
        x, z = pack_string(n, op, qs)
        ms = sv.time_pauli_rotation(x, z, 0.1, warmup=1, reps=4)
        sv.apply_pauli_rotation(x, z, -0.5)   # 5 forward rotations of 0.1 were applied
        dev = np.abs(sv.get_amplitudes(idx) - base).max()
        print(f"  {op[:6]:6s}.. weight {len(qs):2d}: {ms:7.2f} ms = {32.0*(1<<n)/(ms*1e-3)/1e12:.2f} TB/s;  after undoing: max|da| = {dev:.1e}, norm {sv.norm2():.12f}", flush=True)
