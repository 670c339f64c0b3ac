"""30-qubit UCC-type program: LDS-tiled sweeps vs one sweep per same-x run (timing helper)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, o = n // 2, 3
singles, doubles = fermion.uccsd_excitations(m, o)
step = max(1, len(doubles) // 48)
gens = [fermion._excitation_generator(n, [a], [i]) for i, a in singles[::5]]
gens += [fermion._excitation_generator(n, [b, a], [i, j]) for i, j, a, b in doubles[::step]]
hf = fermion.hf_integer(n, 2 * o)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
print(f"n={n} generators={len(gens)} rotations={sum(len(g.terms) for g in gens)}", flush=True)
idx = np.random.default_rng(2).integers(0, 1 << n, 2000).astype(np.uint64)
with Statevector(n) as sv:
    ref = None
    for bits, low in ((0, 4), (11, 4), (12, 4), (11, 5), (11, 6), (12, 6), (11, 3), (11, 2)):
        sv.set_option("tile_low", low); sv.set_option("tile_bits", bits)
        sv.set_ucc_program(gens, hf)
        info = sv.program_info()
        sv.prepare_state(theta)
        t = time.time(); sv.prepare_state(theta); dt = time.time() - t
        amps = sv.get_amplitudes(idx)
        if ref is None: ref = amps
        print(f"tile_bits={bits:2d} low={low}: {dt*1e3:8.2f} ms  sweeps={info['sweeps']:3d} (tiled {info['tiled_sweeps']})  "
              f"{dt*1e3/info['sweeps']:.2f} ms/sweep = {32*2**n/(dt/info['sweeps'])/1e12:.2f} TB/s per sweep  "
              f"max|diff|={np.abs(amps-ref).max():.1e}", flush=True)
