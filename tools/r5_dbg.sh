cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5s
python -m pytest tests/test_gpu_sector.py -x -q -m gpu -k "h2o_uccsd_and_table" > gpurun_out/r5s/dbg1.log 2>&1; head -5 gpurun_out/r5s/dbg1.log | cut -c1-300
OVQE_LIB=testing python - > gpurun_out/r5s/dbg2.log 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from openvqe_amd import chem
from openvqe_amd.backend import Statevector
mol = chem.molecule("H2O"); mol.rhf()
prob = mol.problem(active=False)
ham = prob.jw_hamiltonian()
_, _, gens, theta_mp2, hf = prob.uccsd()
for arr in (0, 1):
    for sw in (2, 3):
        with Statevector(ham.nbqbits) as sv:
            sv.set_option("force_path", 2); sv.set_option("sector_min_qubits", 8)
            sv.set_option("sector_stream_arrange", arr); sv.set_option("sector_sweep", sw); sv.set_option("sector_debug", 6)
            sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
            print("arrange", arr, "sweep", sw, flush=True)
            for i in range(4):
                print(" E", sv.energy(np.array(theta_mp2)), flush=True)
PY
tail -30 gpurun_out/r5s/dbg2.log | cut -c1-300
