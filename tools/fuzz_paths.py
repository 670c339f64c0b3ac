"""Differential fuzzing of the compilation paths (not part of the test suite: minutes of GPU time).
Random UCC / gate programs at 13..17 qubits; every combination of tile size, real-amplitude streaming, table fusion and
Clifford-frame mode must give the same energy (and the reference path's state) within rounding.
usage: python tools/fuzz_paths.py [cases] [seed]"""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian, Term
from tests.util import quccsd_like_gates, random_generators, random_hamiltonian

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
worst = 0.0
svs = {}
for case in range(cases):
    n = int(rng.integers(13, 18))
    kind = str(rng.choice(["ucc_real", "ucc_mixed", "gates_frame", "gates_random"]))
    m = n // 2
    if rng.random() < 0.5:
        H, _, _ = fermion.synthetic_molecule(m, max(1, m // 3), seed=int(rng.integers(1 << 30)))
        if H.nbqbits != n:
            H = Hamiltonian(n, H.terms, H.constant_coeff)
    else:
        H = random_hamiltonian(rng, n, 60)
    hf = int(rng.integers(0, 1 << n))
    sv = svs.setdefault(n, Statevector(n))
    sv.set_option("force_path", 2)
    sv.set_hamiltonian(H)
    if kind.startswith("ucc"):
        gens = fermion.uccsd_generators(m, max(1, m // 3))
        gens = [Hamiltonian(n, g.terms, do_clean_up=False) for g in gens]
        pick = rng.choice(len(gens), min(len(gens), 30), replace=False)
        gens = [gens[i] for i in sorted(pick)]
        if kind == "ucc_mixed":
            gens += random_generators(rng, n, 6)
            gens = [gens[i] for i in rng.permutation(len(gens))]
        K = len(gens)
        load = lambda: sv.set_ucc_program(gens, hf)
        frames = (1,)
    else:
        gates, K = quccsd_like_gates(rng, n, 3, 5, extra_random=0 if kind == "gates_frame" else 15,
                                     disjoint_ladders=kind == "gates_frame")
        load = lambda: sv.set_gate_program(gates, K, hf)
        frames = (0, 1, 2)
    theta = rng.uniform(-0.6, 0.6, K)
    ref = None
    scale = max(1.0, sum(abs(t.coeff) for t in H.terms))
    for bits, real, fusion, frame in itertools.product((0, 10, 11, 12), (0, 1), (0, 1), frames):
        sv.set_option("tile_bits", bits); sv.set_option("real_stream", real); sv.set_option("clifford_frame", frame)
        load()
        sv.set_option("table_fusion", fusion)
        e = sv.energy(theta)
        if ref is None:
            ref = e
        d = abs(e - ref) / scale
        worst = max(worst, d)
        if d > 1e-11:
            print(f"MISMATCH case {case} n={n} {kind} bits={bits} real={real} fusion={fusion} frame={frame}: {e} vs {ref}", flush=True)
    print(f"case {case}: n={n} {kind} K={K} E={ref:.10f} info={sv.program_info()['real_stream']} worst so far {worst:.1e}", flush=True)
print("worst relative deviation", worst)
