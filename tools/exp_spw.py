"""H2O/STO-3G UCCSD on the support-compacted kernel: evaluations per wave (sparse_spw) sweep (timing helper)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from openvqe_amd.backend import Statevector
ham, gens, hf = bench.build_workload()
K = len(gens); B = 65536
th = torch.from_numpy(np.random.default_rng(1).uniform(-0.1, 0.1, (B, K))).cuda()
out = torch.empty(B, dtype=torch.float64, device="cuda")
with Statevector(ham.nbqbits) as sv:
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    sv.energy_batch_device(B, th.data_ptr(), out.data_ptr())
    print(sv.program_info(), flush=True)
    for spw in (1, 2, 4, 2):
        sv.set_option("sparse_spw", spw)
        sv.energy_batch_device(B, th.data_ptr(), out.data_ptr()); torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            sv.energy_batch_device(B, th.data_ptr(), out.data_ptr()); ms.append(sv.last_batch_ms())
        print(f"spw={spw}: kernel {min(ms):.3f} ms -> {B/min(ms)/1e3:.1f} M evals/s  E0={float(out[0]):.10f}", flush=True)
