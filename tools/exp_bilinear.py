"""<bra|sum_t c_t P_t|ket> with bra != ket (the remote contraction of the sharded <H>, distributed.py): G x-groups of 8 strings on a
2^n register, bra = ket buffers of the same handle (torch tensors); ms per call and bytes / s against (16 + 16 G) 2^n.
python tools/exp_bilinear.py [n=26] [G=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openvqe_amd.backend import Statevector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(11)
xs, zs, cs = [], [], []
for g in range(G):
    x = int(rng.integers(1, 1 << n, dtype=np.uint64))
    for t in range(8):
        z = int(rng.integers(0, 1 << n, dtype=np.uint64))
        if bin(x & z).count("1") & 1:           # an even number of Y: real coefficient stays real
            z ^= x & -x
        xs.append(x); zs.append(z); cs.append(rng.standard_normal())
dev = torch.device("cuda:0")
g0 = torch.Generator(device=dev); g0.manual_seed(1)
bra = torch.randn((1 << n, 2), dtype=torch.float64, device=dev, generator=g0)
ket = torch.randn((1 << n, 2), dtype=torch.float64, device=dev, generator=g0)
torch.cuda.synchronize()
with Statevector(n) as sv:
    v = sv.bilinear(xs, zs, cs, bra.data_ptr(), ket.data_ptr())
    ts = []
    for rep in range(5):
        t = time.perf_counter(); v = sv.bilinear(xs, zs, cs, bra.data_ptr(), ket.data_ptr()); ts.append(time.perf_counter() - t)
    dt = min(ts)
    if n <= 22:   # check against numpy
        b = (bra[:, 0] + 1j * bra[:, 1]).cpu().numpy(); k = (ket[:, 0] + 1j * ket[:, 1]).cpu().numpy()
        idx = np.arange(1 << n, dtype=np.uint64)
        ref = 0j
        for x, z, c in zip(xs, zs, cs):
            ny = bin(x & z).count("1")
            sign = 1 - 2 * (np.bitwise_count(idx & np.uint64(z)) & 1).astype(np.float64)   # P|j> = i^ny (-1)^{z.j} |j^x>
            ref += c * (1j ** ny) * np.vdot(b[idx ^ np.uint64(x)], sign * k)
        print(f"  check: device {v:.10f}  numpy {ref:.10f}")
    print(f"n={n} G={G}: {1e3 * dt:.2f} ms per call; (16 + 16 G) 2^n bytes -> {(16 + 16 * G) * (1 << n) / dt / 1e9:.0f} GB/s; 32 G 2^n (a pass per group) would be {32 * G * (1 << n) / 1e9:.1f} GB")
