"""k_sector_expect (the materialised <H> of the sector path) over launch geometries: HIP-event microseconds per evaluation and
the rate at which the table streams.  usage: exp_hexpect.py [m o]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, len(gens))
with Statevector(2 * m) as sv:
    sv.set_option("sector_min_qubits", 8)
    sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf)
    e0 = sv.energy(theta); sv.energy(theta)
    sv.set_option("sector_profile", 1)
    hb = sv.program_info()["sector_h_stream_bytes"]
    for dbg in (1, 2, 3):
        sv.set_option("sector_h_dbg", dbg)
        us = []
        for _ in range(5):
            sv.energy(theta); us.append(sv.program_info()["sector_expect_us"])
        print(f"dbg {dbg} (1 tile loads only, 2 no tile loads, 3 tile loads + slice metadata only): {min(us[1:])} us", flush=True)
    sv.set_option("sector_h_dbg", 0)
    for threads in (512, 1024):
        for groups in (64, 128, 256, 512):
            sv.set_option("sector_h_threads", threads); sv.set_option("sector_h_groups", groups)
            us = []
            for _ in range(6):
                e = sv.energy(theta); us.append(sv.program_info()["sector_expect_us"])
            print(f"threads {threads} groups {groups}: {min(us[1:])} us = {hb / min(us[1:]) / 1e6:.2f} TB/s, dE {e - e0:.1e}", flush=True)
