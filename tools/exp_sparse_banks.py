"""support-compacted kernel (k_sparse_vqe) with and without the bank-aware numbering of the support: kernel-only rate of 65 536
evaluations, single-evaluation latency, colliding lane pairs per evaluation.  usage: exp_sparse_banks.py [H2O|LIH]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
name = sys.argv[1] if len(sys.argv) > 1 else "H2O"
only = [int(a[7:]) for a in sys.argv if a.startswith("--only=")]
mol = chem.molecule(name); mol.rhf()
ham = mol.jw_hamiltonian()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
K = len(gens)
rng = np.random.default_rng(1)
B = 65536
th = rng.uniform(-0.1, 0.1, (B, K))
thd = torch.from_numpy(th).cuda()
out = torch.empty(B, dtype=torch.float64, device="cuda")
ref = None
for renumber, rows in ([(o & 1, o >> 1) for o in only] or ((0, 0), (1, 0), (1, 1))):
    with Statevector(ham.nbqbits) as sv:
        sv.set_option("sparse_renumber", renumber)
        sv.set_option("sparse_rows", rows)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, mol.hf_init())
        sv.energy_batch_device(B, thd.data_ptr(), out.data_ptr())
        ms = []
        for _ in range(5):
            sv.energy_batch_device(B, thd.data_ptr(), out.data_ptr()); ms.append(sv.last_batch_ms())
        e = out.cpu().numpy().copy()
        t0 = time.perf_counter()
        for k in range(200): sv.energy(th[k])
        lat = (time.perf_counter() - t0) / 200
        info = sv.program_info()
        if ref is None: ref = e
        print(f"{name} renumber={renumber} rows={rows}: kernel {min(ms):.3f} ms per {B} = {B / min(ms) / 1e3:.1f} M evals/s, single call {1e6 * lat:.1f} us, "
              f"conflicts {info['sp_conflicts_discovery_order']} -> {info['sp_conflicts']}, support {info['support']}, "
              f"max |dE| vs discovery order {np.abs(e - ref).max():.1e}", flush=True)
