"""LDS-tiled multi-op sweeps vs one sweep per op: state preparation time of (a) the reference's QUCCSD gate templates
and (b) UCCSD Pauli-rotation programs at 2*m qubits (timing helper; `python tools/exp_tile.py 12 5 [nexc]`)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.common_files.circuit import efficient_fermionic_ansatz
from openvqe_amd.qat_compat import AffineParam, Program, lower_circuit

m, o = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12, 5)
nexc = int(sys.argv[3]) if len(sys.argv) > 3 else 300
configs = [(0, 4), (10, 4), (11, 4), (12, 4), (11, 5), (11, 6), (12, 6)]
n = 2 * m
singles, doubles = fermion.uccsd_excitations(m, o)
exci = [[i, a] for i, a in singles] + [[i, j, a, b] for i, j, a, b in doubles]
step = max(1, len(exci) // nexc)
exci = exci[::step]
K = len(exci)
prog = Program(); reg = prog.qalloc(n)
efficient_fermionic_ansatz(reg, prog, exci, [AffineParam(k) for k in range(K)])
_, kind, gates = lower_circuit(prog.to_circ())
hf = fermion.hf_integer(n, 2 * o)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, K)
gens = fermion.uccsd_generators(m, o)[::step]
R = sum(len(g.terms) for g in gens)
print(f"n={n}  excitations={K}  literal gates={len(gates)}  UCC generators={len(gens)} rotations={R}", flush=True)
idx = np.random.default_rng(2).integers(0, 1 << n, 2000).astype(np.uint64)
with Statevector(n) as sv:
    for label, setter in (("QUCCSD gates", lambda: sv.set_gate_program(gates, K, hf)),
                          ("UCCSD rotations", lambda: sv.set_ucc_program(gens, hf))):
        ref = None
        for bits, low in configs:
            sv.set_option("tile_low", low); sv.set_option("tile_bits", bits)
            setter()
            sv.prepare_state(theta)
            t = time.time()
            for _ in range(2): sv.prepare_state(theta)
            dt = (time.time() - t) / 2
            amps = sv.get_amplitudes(idx)
            if ref is None: ref = amps
            print(f"{label:16s} tile_bits={bits:2d} low={low}: {dt*1e3:9.2f} ms   max|diff vs untiled|={np.abs(amps-ref).max():.1e}", flush=True)
