"""fuzz of the support-list paths of the ADAPT side (ovqe_pool_gradients, ovqe_apply_exp_pauli_sum, ovqe_get_support) against the
passes over the register: random register sizes, sparse random complex states, random Hamiltonians (molecule-shaped or random
strings) and pools.  python tools/fuzz_screen.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
from openvqe_amd.operators import Hamiltonian, Term
from tests.util import random_hamiltonian, random_string
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.choice([12, 13, 14, 16, 17, 18, 20, 22]))
    if rng.random() < 0.5 and n % 2 == 0:
        H, _, _ = fermion.synthetic_molecule(n // 2, max(1, n // 4), seed=int(rng.integers(1 << 30)))
        if len(H.terms) > 3000:
            H = Hamiltonian(n, H.terms[:3000], do_clean_up=False)
    else:
        H = random_hamiltonian(rng, n, int(rng.integers(3, 60)))
    pool = []
    for _ in range(int(rng.integers(1, 30))):
        terms = []
        for _ in range(int(rng.integers(1, 9))):
            op, qs = random_string(rng, n)
            terms.append(Term(complex(rng.normal(), rng.normal() if rng.random() < 0.5 else 0.0), op, qs))
        pool.append(Hamiltonian(n, terms, do_clean_up=False))
    anti = []   # anti-Hermitian generators for the exponentials: i * (real Pauli sum)
    for _ in range(3):
        terms = []
        for _ in range(int(rng.integers(1, 5))):
            op, qs = random_string(rng, n)
            terms.append(Term(1j * float(rng.normal()), op, qs))
        anti.append(Hamiltonian(n, terms, do_clean_up=False))
    nnz = int(min(1 << n, rng.choice([1, 2, 7, 60, 500, (1 << n) // 16, (1 << n) // 16 + 1, (1 << n) // 3])))
    psi = np.zeros(1 << n, complex)
    where = rng.choice(1 << n, size=nnz, replace=False)
    psi[where] = rng.normal(size=nnz) + 1j * rng.normal(size=nnz)
    psi /= np.linalg.norm(psi)
    out = {}
    for den in (16, 0):
        with Statevector(n) as sv:
            sv.set_option("screen_sparse", den)
            sv.set_hamiltonian(H)
            sv.set_state(psi)
            g = np.array(sv.pool_gradients(pool, 0)); q = np.array(sv.pool_gradients(pool, 1)); walked = sv.last_screen_support()
            sup = sv.get_support(capacity=1 << n)
            for a, th in zip(anti, (0.3, -0.7, 1.1)):
                sv.apply_exp_pauli_sum(a, th)
            out[den] = (g, q, walked, sup, sv.get_state(), sv.last_exp_support())
    scale = max(1.0, np.abs(H.packed()[2]).sum()) * max(sum(abs(t.coeff) for t in op.terms) for op in pool)
    ok = (np.abs(out[16][0] - out[0][0]).max() < 1e-12 * scale and np.abs(out[16][1] - out[0][1]).max() < 1e-12 * scale
          and np.array_equal(out[16][4], out[0][4]) and out[0][2] == -1
          and out[16][2] == (nnz if nnz * 16 <= (1 << n) else -1)
          and np.array_equal(out[16][3][0], np.sort(where).astype(np.uint64)) and np.array_equal(out[16][3][1], psi[np.sort(where)]))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} nnz={nnz} pool={len(pool)} H terms={len(H.terms)} walked={out[16][2]} exp={out[16][5]} "
              f"dg={np.abs(out[16][0] - out[0][0]).max():.2e} dq={np.abs(out[16][1] - out[0][1]).max():.2e} "
              f"dstate={np.abs(out[16][4] - out[0][4]).max():.2e}", flush=True)
print(f"{cases} cases, {bad} mismatches")

# ---- sigma = H psi from the materialised Hamiltonian of psi's symmetry sector (option "screen_sector") against the register /
# tile-cover pass: real states inside the (o alpha, o beta) sector of a molecule-shaped Hamiltonian, sometimes with a few
# amplitudes in ANOTHER particle-number sector (the closure is then a union of sectors) or with an imaginary part (declined)
from openvqe_amd import pools
bad2 = 0
for case in range(max(4, cases // 3)):
    m = int(rng.choice([6, 7, 8, 9, 10]))
    o = int(rng.integers(1, m // 2 + 1))
    n = 2 * m
    H, _, hf = fermion.synthetic_molecule(m, o, seed=int(rng.integers(1 << 30)))
    sector = np.array([i for i in range(1 << n) if bin(i & 0xAAAAAAAA).count("1") == o and bin(i & 0x55555555).count("1") == o], np.int64) \
        if n <= 16 else None
    psi = np.zeros(1 << n, complex)
    if sector is not None:
        k = int(rng.choice([1, 5, len(sector) // 3 + 1, len(sector)]))
        where = rng.choice(sector, size=min(k, len(sector)), replace=False)
        psi[where] = rng.normal(size=len(where))
    else:   # larger registers: a chain of exponentials from the reference determinant
        where = None
    kind = int(rng.integers(0, 4))   # 0, 1: inside the sector; 2: strays into another sector; 3: complex
    _, _, singlets = pools.singlet_sd(2 * o, m)
    pool = [singlets[int(k)] for k in rng.choice(len(singlets), size=min(len(singlets), 12), replace=False)]
    for _ in range(4):
        op, qs = random_string(rng, n)
        pool.append(Hamiltonian(n, [Term(float(rng.normal()), op, qs)], do_clean_up=False))
    out = {}
    for on in (1, 0):
        with Statevector(n) as sv:
            sv.set_option("screen_sector", on)
            sv.set_option("screen_sector_min", 1)
            sv.set_hamiltonian(H)
            if where is None:
                sv.init_basis(hf)
                if on:
                    picks = rng.choice(len(singlets), size=6, replace=False)
                    thetas = rng.uniform(-0.6, 0.6, 6)
                for k, th in zip(picks, thetas):
                    sv.apply_exp_pauli_sum(singlets[int(k)], float(th))
                state = sv.get_state()
            else:
                state = psi.copy()
            if kind == 2:
                state[[1, (1 << n) - 2]] = 0.3, -0.2
            if kind == 3:
                state[np.flatnonzero(state)[0]] *= (0.6 + 0.8j)
            state = state / np.linalg.norm(state)
            sv.set_state(state)
            g = np.array(sv.pool_gradients(pool, 0)); used = sv.last_screen_sector(); q = np.array(sv.pool_gradients(pool, 1))
            out[on] = (g, q, used)
    scale = max(1.0, np.abs(H.packed()[2]).sum()) * max(sum(abs(t.coeff) for t in op.terms) for op in pool)
    ok = np.abs(out[1][0] - out[0][0]).max() < 1e-12 * scale and np.abs(out[1][1] - out[0][1]).max() < 1e-12 * scale and out[0][2] == 0
    if kind == 3: ok = ok and out[1][2] == 0
    if not ok:
        bad2 += 1
        print(f"SECTOR MISMATCH case {case}: n={n} o={o} kind={kind} used={out[1][2]} dg={np.abs(out[1][0] - out[0][0]).max():.2e} "
              f"dq={np.abs(out[1][1] - out[0][1]).max():.2e} scale={scale:.2e}", flush=True)
    else:
        print(f"sector case {case}: n={n} o={o} kind={kind} sector engine on {out[1][2]} determinants", flush=True)
print(f"sector cases: {bad2} mismatches")
