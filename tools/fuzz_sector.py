"""Differential fuzzing of the sector path (tests/test_gpu_sector.py runs 40 cases of it; more by hand): random UCC-type programs at 14..20
qubits (subsets and random orders of UCCSD generators, single-string generators on few supports, multi-term generators that
fuse to several patterns), JW two-body or random low-weight Hamiltonians, random tile geometry / workgroup sizes / coding:
energies and gradients of the sector path against the dense-state kernels of the same handle.  With OVQE_LIB=testing the draws include
the per-wave streams of the third sweep form (forced wave counts, lanes arranged or not) and the superseded sweep forms.
usage: python tools/fuzz_sector.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian, Term
from tests.util import random_string

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
worst_e = worst_g = 0.0
used = declined = regular = 0
for case in range(cases):
    m = int(rng.integers(7, 11))
    n = 2 * m
    o = int(rng.integers(2, max(3, m // 2 + 1)))
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=int(rng.integers(1 << 30)))
    kind = str(rng.choice(["uccsd_subset", "uccsd_shuffled", "single_strings", "mixed", "quccsd_templates", "quccsd_templates"]))
    gate_program = None
    if kind == "quccsd_templates":
        # the reference's QUCCSD templates on a random list of (generalised) excitations: their states fill the spin-parity quarter of
        # the register — a REGULAR support: sweeps and backward sweeps from bit arithmetic, selectors, blocks of two ops
        from openvqe_amd.common_files.circuit import quccsd_gate_list
        singles, doubles = fermion.uccsd_excitations(m, o)
        exc = [[i, a] for i, a in singles] + [[i, j, a, b] for i, j, a, b in doubles]
        for _ in range(int(rng.integers(0, 12))):      # generalised ones: any two (four) distinct qubits of matching spins
            p, q = (int(v) for v in rng.choice(m, 2, replace=False))
            sp = int(rng.integers(2))
            if rng.random() < 0.4:
                exc.append([2 * p + sp, 2 * q + sp])
            else:
                r, t2 = (int(v) for v in rng.choice(m, 2, replace=False))
                sp2 = int(rng.integers(2))
                quad = [2 * p + sp, 2 * r + sp2, 2 * q + sp, 2 * t2 + sp2]
                if len(set(quad)) == 4:
                    exc.append(quad)
        order = rng.permutation(len(exc)) if rng.random() < 0.5 else np.arange(len(exc))
        take = int(rng.integers(min(30, len(exc)), min(len(exc), 140) + 1))
        if rng.random() < 0.5:
            order = np.sort(order[:take])                 # a subset in list order: long runs that share indices (blocks of two ops)
        else:
            order = order[:take]
        gates, Kq, hfq = quccsd_gate_list(m, o, 1, excitations=[exc[i] for i in order])
        gate_program = (gates, Kq, hfq)
        gens = [None] * Kq
    if kind == "quccsd_templates":
        pass
    elif kind == "uccsd_subset":
        pick = sorted(rng.choice(len(gens), int(rng.integers(5, min(len(gens), 120) + 1)), replace=False))
        gens = [gens[i] for i in pick]
    elif kind == "uccsd_shuffled":
        gens = [gens[i] for i in rng.permutation(len(gens))[: int(rng.integers(10, min(len(gens), 150) + 1))]]
    else:
        supports = [sorted(rng.choice(n, int(rng.choice([2, 4])), replace=False).tolist()) for _ in range(int(rng.integers(3, 7)))]
        singles = []
        for _ in range(int(rng.integers(6, 20))):
            qs = supports[int(rng.integers(len(supports)))]
            ops = ["X"] * len(qs)
            for k in rng.choice(len(qs), 1 if len(qs) == 2 else int(rng.choice([1, 3])), replace=False):
                ops[k] = "Y"
            zq = [q for q in range(n) if q not in qs and rng.random() < 0.3]
            singles.append(Hamiltonian(n, [Term(float(rng.uniform(0.5, 1.5)), "".join(ops) + "Z" * len(zq), qs + zq)], do_clean_up=False))
        gens = singles if kind == "single_strings" else [g for pair in zip(singles, gens[: len(singles)]) for g in pair]
    if rng.random() < 0.3:   # random low-weight Hamiltonian instead of the JW two-body one
        seen, terms = set(), []
        while len(terms) < 120:
            op, qs = random_string(rng, n, 1, 5)
            if (op, tuple(qs)) not in seen:
                seen.add((op, tuple(qs)))
                terms.append(Term(float(rng.normal()), op, qs))
        ham = Hamiltonian(n, terms, 0.1)
    K = len(gens)
    thetas = [rng.uniform(-0.5, 0.5, K) for _ in range(3)]
    opts = {"sector_bits": int(rng.choice([0, 8, 10, 12, 14])), "sector_threads": int(rng.choice([0, 64, 256, 512, 1024])),
            "sector_dict": int(rng.random() < 0.8), "sector_h_bits": int(rng.choice([0, 6, 9, 12])),
            "sector_tile_cap": int(rng.choice([6500, 6500, 700, 150])), "sector_min_qubits": 8,
            # regular supports (full cosets of the program's Z2 symmetries: the single-string kinds produce them): bit-arithmetic sweeps
            "sector_reg_pairs": int(rng.random() < 0.7), "sector_reg_threads": int(rng.choice([128, 256, 512])),
            "sector_reg_adjoint": int(rng.random() < 0.8), "sector_regular": int(rng.choice([1, 1, 1, 3, 0])),
            "sector_reg_runs": int(rng.random() < 0.75)}
    if os.environ.get("OVQE_LIB") == "testing":   # the testing build also takes the options that pick a kernel form: per-wave streams
        # (third sweep form) forced on tiles the product would leave to the second form, over the waves that share a tile's rows
        opts.update({"sector_stream_waves": int(rng.choice([0, 1, 2, 4, 8, 16])), "sector_stream_arrange": int(rng.random() < 0.5),
                     "sector_sweep": int(rng.choice([3, 3, 3, 2, 1])), "sector_adjoint": int(rng.choice([3, 3, 2, 1]))})
        if rng.random() < 0.6:
            opts["sector_threads"] = int(rng.choice([0, 1024]))      # (the streams are for 1024-thread workgroups)
            opts["sector_regular"] = 0
    scale = max(1.0, float(np.abs(ham.packed()[2]).sum()))
    with Statevector(n) as sv:
        sv.set_option("force_path", 2)
        for k, v in opts.items():
            sv.set_option(k, v)
        sv.set_hamiltonian(ham)
        if gate_program:
            sv.set_gate_program(*gate_program)
        else:
            sv.set_ucc_program(gens, hf)
        es = [sv.energy(t) for t in thetas]
        info = sv.program_info()
        eg = [sv.energy_gradient(t) for t in thetas[1:]]
        sv.set_option("sector", 0)
        ed = [sv.energy(t) for t in thetas]
        egd = [sv.energy_gradient(t) for t in thetas[1:]]
    de = max(abs(a - b) for a, b in zip(es, ed)) / scale
    dg = max(float(np.abs(a[1] - b[1]).max()) for a, b in zip(eg, egd)) / scale
    dg = max(dg, max(abs(a[0] - b[0]) for a, b in zip(eg, egd)) / scale)
    worst_e, worst_g = max(worst_e, de), max(worst_g, dg)
    used += info["sector_support"] > 0
    declined += info["sector_support"] == 0
    regular += info["sector_regular_slot_bits"] > 0
    flag = "" if de < 1e-11 and dg < 1e-10 else "   <-- MISMATCH"
    print(f"case {case}: n={n} o={o} {kind} K={K} opts={opts} support={info['sector_support']} sweeps={info['sector_sweeps']} "
          f"h_sweeps={info['sector_h_sweeps']} regular={info['sector_regular_slot_bits']}/{info['sector_free_bits']} dE={de:.1e} dG={dg:.1e}{flag}", flush=True)
print(f"worst dE/|H|_1 = {worst_e:.2e}, worst dG/|H|_1 = {worst_g:.2e}; sector path used in {used} cases ({regular} of them on bit-arithmetic sweeps), declined in {declined}")
if worst_e >= 1e-11 or worst_g >= 1e-10 or used == 0:
    sys.exit(1)
