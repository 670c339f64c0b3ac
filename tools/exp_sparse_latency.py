"""latency path of the support-compacted evaluation: one wave per evaluation with staged tables (k_sparse_vqe<1, true>) against
one 1024-thread workgroup per evaluation (k_sparse_vqe_wg): microseconds per call for small batches.  usage: [H2O|LIH|H2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, fermion
from openvqe_amd.backend import Statevector
name = sys.argv[1] if len(sys.argv) > 1 else "H2O"
mol = chem.molecule(name); mol.rhf()
ham = mol.jw_hamiltonian()
gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
K = len(gens)
rng = np.random.default_rng(1)
th = rng.uniform(-0.1, 0.1, (1024, K))
ref = {}
for wg in (0, 1):
    with Statevector(ham.nbqbits) as sv:
        sv.set_option("sparse_wg", wg)
        sv.set_hamiltonian(ham); sv.set_ucc_program(gens, mol.hf_init())
        row = []
        for B in (1, 16, K + 1, 1024):
            sv.energy_batch(th[:B])
            reps = 200 if B <= 16 else 50
            t0 = time.perf_counter()
            for _ in range(reps): e = sv.energy_batch(th[:B])
            dt = (time.perf_counter() - t0) / reps
            d = np.abs(e - ref.setdefault(B, e)).max()
            row.append(f"B={B}: {1e6 * dt:.1f} us (max |dE| {d:.1e})")
        t0 = time.perf_counter()
        for k in range(100): sv.energy_gradient(th[k])
        row.append(f"exact gradient: {1e4 * (time.perf_counter() - t0):.1f} us")
        print(f"{name} sparse_wg={wg}: " + ", ".join(row), flush=True)
