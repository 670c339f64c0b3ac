"""30-qubit UCC-type energy: real-amplitude streaming vs complex (timing helper)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian, Term
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m, o = n // 2, 3
singles, doubles = fermion.uccsd_excitations(m, o)
step = max(1, len(doubles) // 48)
gens = [fermion._excitation_generator(n, [a], [i]) for i, a in singles[::5]]
gens += [fermion._excitation_generator(n, [b, a], [i, j]) for i, j, a, b in doubles[::step]]
hf = fermion.hf_integer(n, 2 * o)
rng = np.random.default_rng(1)
theta = rng.uniform(-0.1, 0.1, len(gens))
terms = [Term(float(rng.normal()), "Z", [int(q)]) for q in range(n)]
for _ in range(40):
    q = sorted(rng.choice(n, 4, replace=False).tolist())
    terms += [Term(0.1, "XXYY", q), Term(-0.1, "YYXX", q)]
    p = sorted(rng.choice(n, 2, replace=False).tolist())
    terms += [Term(0.2, "X" + "Z" * (p[1] - p[0] - 1) + "X", list(range(p[0], p[1] + 1))),
              Term(0.2, "Y" + "Z" * (p[1] - p[0] - 1) + "Y", list(range(p[0], p[1] + 1)))]
H = Hamiltonian(n, terms, 0.0)
print(f"n={n} generators={len(gens)} rotations={sum(len(g.terms) for g in gens)} H terms={len(H.terms)}", flush=True)
with Statevector(n) as sv:
    sv.set_option("force_path", 2)
    sv.set_hamiltonian(H)
    for real, bits in ((0, 12), (1, 12), (0, 11), (1, 11), (1, 10)):
        sv.set_option("real_stream", real); sv.set_option("tile_bits", bits)
        sv.set_ucc_program(gens, hf)
        e = sv.energy(theta)
        t = time.time(); e = sv.energy(theta); dt = time.time() - t
        info = sv.program_info()
        print(f"real_stream={real} tile_bits={bits}: energy {dt*1e3:8.2f} ms  E={e:.12f}  circuit sweeps={info['sweeps']} "
              f"H sweeps={info['h_tile_sweeps']}+{info['h_untiled_groups']}", flush=True)
