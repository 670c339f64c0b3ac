"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on seeded inputs.

Tolerances: amplitudes max-abs 1e-12, energies 1e-10 relative to |H|_1 scale (north_star: 1e-9 Ha).
"""
from math import comb

import numpy as np
import pytest

from oracle import dense, masks
from tests.util import random_generators, random_hamiltonian, random_state, random_string

pytestmark = pytest.mark.gpu

AMP_TOL = 1e-12


@pytest.fixture(scope="module")
def SV(gpu_lib):
    from openvqe_amd.backend import Statevector
    return Statevector


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 11, 12])
def test_single_pauli_rotation_matches_oracle(SV, n):
    rng = np.random.default_rng(100 + n)
    with SV(n) as sv:
        for _ in range(12):
            psi = random_state(rng, n)
            op, qs = random_string(rng, n)
            phi = float(rng.uniform(-np.pi, np.pi))
            x, z = masks.pack_pauli(n, op, qs)
            ref = masks.rotate(psi, x, z, phi) if n > 8 else dense.pauli_rotation(psi, n, op, qs, phi)
            sv.set_state(psi)
            sv.rotate(op, qs, phi)
            assert np.abs(sv.get_state() - ref).max() < AMP_TOL, (op, qs)


@pytest.mark.parametrize("n", [2, 6, 10, 13])
def test_fused_rotation_sequence(SV, n):
    rng = np.random.default_rng(200 + n)
    with SV(n) as sv:
        psi = random_state(rng, n)
        xs, zs, phis = [], [], []
        ref = psi.copy()
        for _ in range(40):
            if xs and rng.random() < 0.6:  # same x mask, different z: fused into one sweep
                x = xs[-1]
                z = int(rng.integers(0, 1 << n))
            else:
                op, qs = random_string(rng, n)
                x, z = masks.pack_pauli(n, op, qs)
            phi = float(rng.uniform(-1, 1))
            xs.append(x); zs.append(z); phis.append(phi)
            ref = masks.rotate(ref, x, z, phi)
        sv.set_state(psi)
        sv.apply_pauli_rotations(xs, zs, phis)
        assert np.abs(sv.get_state() - ref).max() < AMP_TOL


@pytest.mark.parametrize("n", [2, 4, 7, 11])
def test_gates_match_oracle(SV, n):
    rng = np.random.default_rng(300 + n)
    with SV(n) as sv:
        psi = random_state(rng, n)
        sv.set_state(psi)
        ref = psi.copy()
        for _ in range(30):
            name = str(rng.choice(["X", "H", "RX", "RY", "RZ", "CNOT"]))
            if name == "CNOT":
                c, t = rng.choice(n, 2, replace=False).tolist()
                sv.apply_gate("CNOT", [c, t])
                ref = masks.gate_cnot(ref, n, c, t)
            else:
                q = int(rng.integers(0, n))
                ang = float(rng.uniform(-3, 3)) if name.startswith("R") else None
                sv.apply_gate(name, [q], ang)
                ref = masks.gate_1q(ref, n, q, dense.gate_matrix(name, ang))
        assert np.abs(sv.get_state() - ref).max() < AMP_TOL


@pytest.mark.parametrize("n,nterms", [(1, 2), (3, 10), (6, 60), (10, 200), (12, 300)])
def test_expectation(SV, n, nterms):
    rng = np.random.default_rng(400 + n)
    nterms = min(nterms, 4 ** n - 1)
    H = random_hamiltonian(rng, n, nterms)
    psi = random_state(rng, n)
    xs, zs, cs = H.packed()
    ref = masks.expectation(psi, xs, zs, cs.real, H.constant_coeff)
    if n <= 6:
        assert abs(ref - dense.expectation(H, psi)) < 1e-12
    with SV(n) as sv:
        sv.set_state(psi)
        got = sv.expectation(H)
    assert abs(got - ref) < 1e-11 * max(1.0, np.abs(cs).sum())


@pytest.mark.parametrize("force_path", [1, 2])
@pytest.mark.parametrize("n,k", [(2, 3), (4, 6), (7, 10), (10, 12), (12, 20), (13, 8), (14, 6)])
def test_ucc_energy_and_state(SV, n, k, force_path):
    rng = np.random.default_rng(500 + 10 * n + force_path)
    H = random_hamiltonian(rng, n, min(40, 4 ** n - 1))
    gens = random_generators(rng, n, k)
    hf = int(rng.integers(0, 1 << n))
    thetas = rng.uniform(-0.5, 0.5, size=(5, k))
    xs, zs, cs = H.packed()
    with SV(n) as sv:
        sv.set_option("force_path", force_path)
        sv.set_hamiltonian(H)
        sv.set_ucc_program(gens, hf)
        e_batch = sv.energy_batch(thetas)
        for b in range(thetas.shape[0]):
            ref = masks  # mask oracle, validated against the dense one in tests/test_oracle.py
            psi = np.zeros(1 << n, complex); psi[hf] = 1
            for g, th in zip(gens, thetas[b]):
                for t in g.terms:
                    x, z = masks.pack_pauli(n, t.op, t.qbits)
                    psi = masks.rotate(psi, x, z, th * t.coeff)
            e_ref = masks.expectation(psi, xs, zs, cs.real, H.constant_coeff)
            assert abs(e_batch[b] - e_ref) < 1e-10 * max(1.0, np.abs(cs).sum())
            if b == 0:
                assert abs(sv.energy(thetas[0]) - e_ref) < 1e-10 * max(1.0, np.abs(cs).sum())
                sv.prepare_state(thetas[0])
                assert np.abs(sv.get_state() - psi).max() < AMP_TOL


@pytest.mark.parametrize("force_path", [1, 2])
def test_gate_program_energy(SV, force_path):
    n = 6
    rng = np.random.default_rng(77)
    H = random_hamiltonian(rng, n, 30)
    gates = []
    K = 4
    for _ in range(60):
        name = str(rng.choice(["X", "H", "RX", "RY", "RZ", "CNOT"]))
        if name == "CNOT":
            c, t = rng.choice(n, 2, replace=False).tolist()
            gates.append((name, [c, t], 0.0, 0.0, -1))
        elif name in ("X", "H"):
            gates.append((name, [int(rng.integers(0, n))], 0.0, 0.0, -1))
        else:
            p = int(rng.integers(-1, K))
            gates.append((name, [int(rng.integers(0, n))], float(rng.choice([1.0, -1.0, -2.0])), float(rng.uniform(-1, 1)), p))
    theta = rng.uniform(-1, 1, K)
    hf = 0b101100
    psi = dense.basis_state(n, hf)
    for name, qs, sc, co, p in gates:
        ang = co + (sc * theta[p] if p >= 0 else 0.0)
        psi = dense.apply_gate(psi, n, name, qs, ang)
    e_ref = dense.expectation(H, psi)
    with SV(n) as sv:
        sv.set_option("force_path", force_path)
        sv.set_hamiltonian(H)
        sv.set_gate_program(gates, K, hf)
        assert abs(sv.energy(theta) - e_ref) < 1e-11 * 30
        sv.prepare_state(theta)
        assert np.abs(sv.get_state() - psi).max() < AMP_TOL


@pytest.mark.parametrize("n", [3, 6, 9])
def test_pool_gradients_and_exact_exponential(SV, n):
    from openvqe_amd.operators import Hamiltonian, Term
    rng = np.random.default_rng(600 + n)
    H = random_hamiltonian(rng, n, min(30, 4 ** n - 1))
    # anti-Hermitian pool operators: i * (real Pauli sum)
    pool = []
    for _ in range(7):
        terms = []
        for _ in range(int(rng.integers(1, 5))):
            op, qs = random_string(rng, n)
            terms.append(Term(1j * float(rng.normal()), op, qs))
        pool.append(Hamiltonian(n, terms, do_clean_up=False))
    psi = random_state(rng, n)
    hmat = dense.operator_matrix(H, sparse=True)
    pool_mats = [dense.operator_matrix(a, sparse=True, with_constant=False) for a in pool]
    g_ref = dense.fermionic_pool_gradients(pool_mats, hmat, psi)
    q_ref = [2.0 * abs(np.vdot(psi, hmat @ (m @ psi))) for m in pool_mats]
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        sv.set_state(psi)
        g = sv.pool_gradients(pool, 0)
        q = sv.pool_gradients(pool, 1)
        scale = max(1.0, np.abs(H.packed()[2]).sum())
        assert np.abs(np.array(g) - np.array(g_ref)).max() < 1e-11 * scale
        assert np.abs(np.array(q) - np.array(q_ref)).max() < 1e-11 * scale
        # exact exp(theta A) psi
        ref = dense.exact_exp_state(psi, pool_mats[:3], [0.3, -0.7, 1.9])
        for a, th in zip(pool[:3], [0.3, -0.7, 1.9]):
            sv.apply_exp_pauli_sum(a, th)
        assert np.abs(sv.get_state() - ref).max() < 1e-11


@pytest.mark.parametrize("n,nnz", [(12, 1), (14, 37), (16, 3000), (16, 5000), (20, 60), (20, 1)])
def test_pool_gradients_on_the_support_of_psi(SV, n, nnz):
    """ovqe_pool_gradients over the list of non-zero amplitudes ("screen_sparse": the ADAPT state of a few operators) against
    the pass over the register and against the bit-mask oracle; complex amplitudes, complex pool coefficients, both modes"""
    from openvqe_amd.operators import Hamiltonian, Term, pack_terms
    from oracle import masks
    rng = np.random.default_rng(900 + n + nnz)
    H = random_hamiltonian(rng, n, 40)
    pool = []
    for _ in range(23):
        terms = []
        for _ in range(int(rng.integers(1, 9))):
            op, qs = random_string(rng, n)
            terms.append(Term(complex(rng.normal(), rng.normal()), op, qs))
        pool.append(Hamiltonian(n, terms, do_clean_up=False))
    psi = np.zeros(1 << n, complex)
    where = rng.choice(1 << n, size=nnz, replace=False)
    psi[where] = rng.normal(size=nnz) + 1j * rng.normal(size=nnz)
    psi /= np.linalg.norm(psi)
    hx, hz, hc = H.packed()
    sigma = masks.apply_pauli_sum(psi, hx, hz, hc) + H.constant_coeff * psi
    vals = []
    for op in pool:
        px, pz, pc = pack_terms(n, op.terms)
        vals.append(np.vdot(sigma, masks.apply_pauli_sum(psi, px, pz, pc)))
    vals = np.array(vals)
    scale = max(1.0, np.abs(hc).sum()) * max(np.abs(pack_terms(n, op.terms)[2]).sum() for op in pool)
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        sv.set_state(psi)
        got = {}
        for den in (16, 0):
            sv.set_option("screen_sparse", den)
            got[den] = (np.array(sv.pool_gradients(pool, 0)), np.array(sv.pool_gradients(pool, 1)))
            expect_list = den > 0 and nnz * den <= (1 << n)
            assert sv.last_screen_support() == (nnz if expect_list else -1)
    for den in (16, 0):
        assert np.abs(got[den][0] - 2.0 * vals.real).max() < 1e-12 * scale
        assert np.abs(got[den][1] - 2.0 * np.abs(vals)).max() < 1e-12 * scale
    assert np.abs(got[16][0] - got[0][0]).max() < 1e-13 * scale


@pytest.mark.parametrize("m,o", [(6, 3), (8, 4)])
def test_exact_exponentials_on_the_reachable_support(SV, m, o):
    """ovqe_apply_exp_pauli_sum with its Taylor steps over the closure of the support ("screen_sparse") against the pass over
    the register: bit-identical states; the chain of spin-adapted generators from the Hartree-Fock determinant (the
    prepare_adapt_state of the ADAPT mirrors, ref:openvqe/adapt/fermionic_adapt_vqe.py:12-38) stays inside the particle-number
    sector, and matches scipy's expm_multiply on the oracle's matrices at the smaller size"""
    from openvqe_amd import fermion, pools
    n = 2 * m
    _, _, hf = fermion.synthetic_molecule(m, o, seed=3)
    _, _, pool = pools.singlet_sd(2 * o, m)
    rng = np.random.default_rng(m)
    picks = rng.choice(len(pool), size=6, replace=False)
    thetas = rng.uniform(-0.8, 0.8, len(picks))
    states, reach = {}, {}
    for den in (16, 0):
        with SV(n) as sv:
            sv.set_option("screen_sparse", den)
            sv.init_basis(hf)
            reach[den] = []
            for k, th in zip(picks, thetas):
                sv.apply_exp_pauli_sum(pool[k], th)
                reach[den].append(sv.last_exp_support())
            states[den] = sv.get_state()
    assert all(r == -1 for r in reach[0])
    assert all(0 < r <= (1 << n) // 16 for r in reach[16]), reach[16]
    assert reach[16][0] <= 16 and reach[16][-1] >= reach[16][0]
    assert np.array_equal(states[16], states[0])
    assert abs(np.linalg.norm(states[16]) - 1.0) < 1e-12
    occupied = np.flatnonzero(states[16])   # EVERY non-zero amplitude: cancelling coefficients leave no residues in other sectors
    assert 1 < len(occupied) <= comb(m, o) ** 2 and all(bin(int(i)).count("1") == 2 * o for i in occupied)
    if n <= 12:
        mats = [dense.operator_matrix(pool[k], sparse=True, with_constant=False) for k in picks]
        psi0 = np.zeros(1 << n, complex)
        psi0[hf] = 1.0
        ref = dense.exact_exp_state(psi0, mats, list(thetas))
        assert np.abs(states[16] - ref).max() < 1e-11


@pytest.mark.parametrize("m,o", [(6, 2), (8, 3)])
def test_chain_of_exponentials_keeps_its_support_list(SV, m, o):
    """back-to-back ovqe_apply_exp_pauli_sum calls after ovqe_init_basis (prepare_adapt_state of the ADAPT mirrors) extend the list
    the previous call left instead of scanning the register again; any other entry point in between drops the list.  Both ways,
    and the pass over the whole register, give the same amplitudes bit for bit; a state pointer handed out ends the shortcut"""
    from openvqe_amd import fermion, pools
    n = 2 * m
    _, _, hf = fermion.synthetic_molecule(m, o, seed=5)
    _, _, pool = pools.singlet_sd(2 * o, m)
    rng = np.random.default_rng(10 * m + o)
    picks = rng.choice(len(pool), size=8, replace=False)
    thetas = rng.uniform(-0.9, 0.9, len(picks))
    thetas[3] = 0.0                                   # (a call that leaves the state alone keeps the list)
    states, reach = {}, {}
    for mode in ("chain", "interrupted", "exposed", "register"):
        with SV(n) as sv:
            sv.set_option("screen_sparse", 0 if mode == "register" else 4)
            if mode == "exposed":
                sv.state_ptr()
            sv.init_basis(hf)
            for k, th in zip(picks, thetas):
                sv.apply_exp_pauli_sum(pool[k], th)
                if mode == "interrupted":
                    assert abs(sv.norm2() - 1.0) < 1e-12
            reach[mode] = sv.last_exp_support()
            states[mode] = sv.get_state()
    for mode in ("interrupted", "exposed", "register"):
        assert np.array_equal(states["chain"], states[mode]), mode
    # the kept list is a superset of the scanned one (amplitudes that cancelled to zero stay listed), both inside the cap
    assert reach["register"] == -1 and 0 < reach["interrupted"] <= reach["chain"] <= (1 << n) // 4, reach
    assert reach["exposed"] == reach["interrupted"]
    occupied = np.flatnonzero(states["chain"])
    assert 1 < len(occupied) <= reach["chain"] and all(bin(int(i)).count("1") == 2 * o for i in occupied)


def test_errors_are_reported(SV):
    from openvqe_amd._lib import BackendError
    with SV(3) as sv:
        with pytest.raises(BackendError):
            sv.energy([0.1])  # no program
        with pytest.raises(BackendError):
            sv.apply_pauli_rotation(1 << 5, 0, 0.1)  # mask beyond the register


@pytest.mark.parametrize("n,g", [(6, 1), (9, 2)])
def test_shard_handles_global_z_and_cross_shard_bilinear(SV, n, g):
    """Two/four shard handles on one GPU stand for the ranks of a sharded state: local-x rotations with
    global z bits (rank-dependent sign) and <own|P|partner> for global-x terms (ovqe_bilinear)."""
    rng = np.random.default_rng(900 + n)
    nl = n - g
    psi = random_state(rng, n)
    shards = [SV(nl, n_global=g, shard_index=s) for s in range(1 << g)]
    try:
        for s, sv in enumerate(shards):
            sv.set_state(psi[s << nl:(s + 1) << nl])
        ref = psi.copy()
        for _ in range(10):
            x = int(rng.integers(0, 1 << nl))           # local x
            z = int(rng.integers(0, 1 << n))            # z anywhere, incl. rank bits
            phi = float(rng.uniform(-1, 1))
            ref = masks.rotate(ref, x, z, phi)
            for sv in shards:
                sv.apply_pauli_rotation(x, z, phi)
        got = np.concatenate([sv.get_state() for sv in shards])
        assert np.abs(got - ref).max() < AMP_TOL
        # expectation of terms with global x: sum over shards of <own|P|partner shard>
        T = 12
        xs = rng.integers(0, 1 << n, T).astype(np.uint64)
        zs = rng.integers(0, 1 << n, T).astype(np.uint64)
        cs = rng.normal(size=T)
        want = masks.expectation(ref, xs, zs, cs)
        total = 0j
        for s, sv in enumerate(shards):
            for t in range(T):
                partner = s ^ (int(xs[t]) >> nl)
                total += sv.bilinear(xs[t:t + 1], zs[t:t + 1], cs[t:t + 1], ket_ptr=shards[partner].state_ptr())
        assert abs(total.real - want) < 1e-11 and abs(total.imag) < 1e-11
    finally:
        for sv in shards:
            sv.close()


def test_randomize_is_reproducible_on_host(SV):
    """ovqe_randomize's counter-based generator restated on the host (openvqe_amd/synth.py) — the basis of the
    sampled full-size parity test at 30 qubits"""
    from openvqe_amd import synth
    with SV(10) as sv:
        scale = sv.randomize(4242)
        got = sv.get_state()
        want = synth.amplitudes(4242, np.arange(1 << 10, dtype=np.uint64)) * scale
        assert np.array_equal(got, want)
        assert abs(sv.norm2() - 1.0) < 1e-12


@pytest.mark.parametrize("n,m,o", [(8, 4, 2), (12, 6, 2), (14, 7, 5)])
def test_real_mode_matches_complex_mode_and_oracle(SV, n, m, o):
    """UCCSD generators (odd #Y everywhere) select the real-amplitude fused kernel; it must agree with the
    complex kernel (real_mode = 0), the streaming path and the oracle."""
    from openvqe_amd import fermion
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=77 + n)
    rng = np.random.default_rng(n)
    K = len(gens)
    thetas = rng.uniform(-0.2, 0.2, size=(3, K))
    xs, zs, cs = ham.packed()
    psi = np.zeros(1 << n, complex); psi[hf] = 1
    for g, th in zip(gens, thetas[0]):
        for t in g.terms:
            x, z = masks.pack_pauli(n, t.op, t.qbits)
            psi = masks.rotate(psi, x, z, th * t.coeff)
    e_ref = masks.expectation(psi, xs, zs, cs.real, ham.constant_coeff)
    assert np.abs(psi.imag).max() < 1e-15  # the state really is real
    scale = max(1.0, np.abs(cs).sum())
    res = {}
    with SV(n) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        for label, opts in (("sparse", {"force_path": 3, "real_mode": 1, "table_fusion": 1}),
                            ("auto", {"force_path": 0, "real_mode": 1, "table_fusion": 1}),
                            ("real", {"force_path": 1, "real_mode": 1, "table_fusion": 1}),
                            ("complex", {"force_path": 1, "real_mode": 0, "table_fusion": 1}),
                            ("real_seq", {"force_path": 1, "real_mode": 1, "table_fusion": 0}),
                            ("complex_seq", {"force_path": 1, "real_mode": 0, "table_fusion": 0}),
                            ("stream", {"force_path": 2})):
            for k, v in opts.items():
                sv.set_option(k, v)
            res[label] = sv.energy_batch(thetas)
    for label in res:
        assert abs(res[label][0] - e_ref) < 1e-10 * scale, label
    for label in res:
        assert np.abs(res[label] - res["stream"]).max() < 1e-10 * scale, label


def test_streaming_path_20_qubits_against_c_oracle(SV):
    """UCCSD rotations at 20 qubits on the streaming kernels (fused same-x sweeps, x-grouped expectation)
    against the plain-C oracle (OpenMP, same inputs): energy within 1e-10 * |H|_1, sampled amplitudes 1e-12"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    n = 20
    gens = fermion.uccsd_generators(10, 3)[::7]  # every 7th generator keeps singles and doubles
    hf = fermion.hf_integer(n, 6)
    ham = random_hamiltonian(np.random.default_rng(2020), n, 300)
    rng = np.random.default_rng(20)
    theta = rng.uniform(-0.2, 0.2, len(gens))
    rx, rz, rc, pidx, K = compile_ucc_program(n, gens)
    hx, hz, hc = ham.packed()
    e_ref, psi_ref = cref.ucc_energy(n, hf, rx, rz, rc, pidx, theta, hx, hz, hc.real.copy(), ham.constant_coeff, 0)
    with SV(n) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        e = sv.energy(theta)
        sv.prepare_state(theta)
        idx = rng.integers(0, 1 << n, 5000).astype(np.uint64)
        amps = sv.get_amplitudes(idx)
        n2 = sv.norm2()
    assert abs(e - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum())
    assert np.abs(amps - psi_ref[idx.astype(np.int64)]).max() < 1e-12
    assert abs(n2 - 1.0) < 1e-11


def test_index_streams_match_in_kernel_indices(SV):
    """host-precomputed pair-index streams (index_streams = 1, default) against in-kernel index arithmetic (0)"""
    from openvqe_amd import chem, fermion
    mol = chem.molecule("H2O"); mol.rhf()
    ham = mol.jw_hamiltonian()
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
    thetas = np.random.default_rng(3).uniform(-0.2, 0.2, (4, len(gens)))
    with SV(14) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, mol.hf_init())
        out = {}
        for level in (1, 0):
            sv.set_option("index_streams", level)
            out[level] = sv.energy_batch(thetas)
    assert np.abs(out[1] - out[0]).max() < 1e-10


def test_sharded_statevector_single_rank_hip_engine(gpu_lib):
    """openvqe_amd.distributed.ShardedStatevector on its product engine (torch tensor adopted by the C ABI handle)
    at world size 1: same rotations / expectation as the plain handle and the oracle."""
    import torch
    from openvqe_amd.distributed import ShardedStatevector
    n = 12
    rng = np.random.default_rng(1212)
    R, T = 30, 40
    xs = [int(v) for v in rng.integers(0, 1 << n, R)]
    zs = [int(v) for v in rng.integers(0, 1 << n, R)]
    xs[4] = 0
    xs[9] = xs[8]
    phis = rng.uniform(-1, 1, R)
    hx = [int(v) for v in rng.integers(0, 1 << n, T)]
    hz = [int(v) for v in rng.integers(0, 1 << n, T)]
    hc = rng.normal(size=T)
    hf = int(rng.integers(0, 1 << n))
    sv = ShardedStatevector(n, device=0)
    e = sv.energy(hx, hz, hc, -0.5, xs, zs, phis, hf)
    got = sv.gather_state()
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    assert np.abs(got - psi).max() < AMP_TOL
    assert abs(e - masks.expectation(psi, hx, hz, hc, -0.5)) < 1e-11 * max(1.0, np.abs(hc).sum())
    assert abs(sv.norm2() - 1.0) < 1e-12
    assert sv.engine.tensor.is_cuda and sv.engine.tensor.dtype == torch.complex128
    sv.engine.sv.close()


def test_real_gate_program_uses_real_kernel_and_matches_oracle(SV):
    """X / H / CNOT / RY only: every rotation entry has an odd number of Y, so the fused kernel runs in real mode"""
    n = 7
    rng = np.random.default_rng(707)
    H = random_hamiltonian(rng, n, 40)
    K = 5
    gates = []
    for _ in range(80):
        name = str(rng.choice(["X", "H", "RY", "CNOT"]))
        if name == "CNOT":
            c, t = rng.choice(n, 2, replace=False).tolist()
            gates.append((name, [c, t], 0.0, 0.0, -1))
        elif name == "RY":
            gates.append((name, [int(rng.integers(0, n))], float(rng.choice([1.0, -1.0, -2.0])), float(rng.uniform(-1, 1)),
                          int(rng.integers(-1, K))))
        else:
            gates.append((name, [int(rng.integers(0, n))], 0.0, 0.0, -1))
    theta = rng.uniform(-1, 1, K)
    hf = 0b1011000
    psi = dense.basis_state(n, hf)
    for name, qs, sc, co, p in gates:
        psi = dense.apply_gate(psi, n, name, qs, co + (sc * theta[p] if p >= 0 else 0.0))
    assert np.abs(psi.imag).max() < 1e-15
    e_ref = dense.expectation(H, psi)
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        sv.set_gate_program(gates, K, hf)
        for real_mode in (1, 0):
            sv.set_option("real_mode", real_mode)
            assert abs(sv.energy(theta) - e_ref) < 1e-11 * 40


@pytest.mark.parametrize("n", [15, 16])
def test_batched_path_above_lds_capacity(SV, n):
    """n = 15, 16: batches >= 32 run the fused kernel with per-workgroup global state slices; single evaluations take
    the streaming kernels — both must agree with each other and with the mask oracle"""
    from openvqe_amd import fermion
    rng = np.random.default_rng(n)
    gens = fermion.uccsd_generators(n // 2, 2)[:: max(1, n // 3)] if n % 2 == 0 else \
        [g for g in random_generators(rng, n, 6)]
    ham = random_hamiltonian(rng, n, 60)
    hf = int(rng.integers(0, 1 << n))
    thetas = rng.uniform(-0.3, 0.3, (40, len(gens)))
    xs, zs, cs = ham.packed()
    with SV(n) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        batch = sv.energy_batch(thetas)        # B = 40 >= 32: fused kernel
        single = np.array([sv.energy(t) for t in thetas[:3]])   # streaming kernels
    assert np.abs(batch[:3] - single).max() < 1e-10 * max(1.0, np.abs(cs).sum())
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for g, th in zip(gens, thetas[0]):
        for t in g.terms:
            x, z = masks.pack_pauli(n, t.op, t.qbits)
            psi = masks.rotate(psi, x, z, th * t.coeff)
    assert abs(batch[0] - masks.expectation(psi, xs, zs, cs.real, ham.constant_coeff)) < 1e-10 * max(1.0, np.abs(cs).sum())


def test_device_resident_batches_and_sparse_fallbacks(SV):
    """ovqe_energy_batch_device (theta / energies as torch CUDA tensors) on the support-compacted and the dense fused
    kernel; programs without a compact support (complex gates) must refuse force_path = 3 and fall back otherwise."""
    import torch
    from openvqe_amd import chem, fermion
    from openvqe_amd._lib import BackendError
    mol = chem.molecule("LIH"); mol.rhf()
    ham = mol.jw_hamiltonian()
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
    B = 300
    th = np.random.default_rng(5).uniform(-0.2, 0.2, (B, len(gens)))
    td = torch.from_numpy(th).cuda()
    with SV(12) as sv:
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, mol.hf_init())
        sv.set_option("force_path", 2)
        ref = sv.energy_batch(th[:6])                      # streaming kernels
        out = {}
        for path in (3, 1, 0):
            sv.set_option("force_path", path)
            ed = torch.zeros(B, dtype=torch.float64, device="cuda")
            sv.energy_batch_device(B, td.data_ptr(), ed.data_ptr())
            out[path] = ed.cpu().numpy()
            assert np.abs(out[path][:6] - ref).max() < 1e-10
            assert np.abs(sv.energy_batch(th) - out[path]).max() < 1e-12   # host-buffer entry point, same kernel
        assert np.abs(out[3] - out[1]).max() < 1e-10 and np.array_equal(out[3], out[0])
    with SV(5) as sv:
        rng = np.random.default_rng(8)
        sv.set_hamiltonian(random_hamiltonian(rng, 5, 20))
        gates = [("H", [0], 0.0, 0.0, -1), ("RZ", [0], 1.0, 0.0, 0), ("CNOT", [0, 3], 0.0, 0.0, -1), ("RX", [3], 1.0, 0.1, 1)]
        sv.set_gate_program(gates, 2, 0)
        e_auto = sv.energy([0.3, -0.2])                     # falls back to the dense fused kernel
        sv.set_option("force_path", 2)
        assert abs(sv.energy([0.3, -0.2]) - e_auto) < 1e-12
        sv.set_option("force_path", 3)
        with pytest.raises(BackendError):
            sv.energy([0.3, -0.2])


@pytest.mark.parametrize("n,kind", [(1, "random"), (4, "random"), (8, "molecule"), (10, "random"), (12, "molecule"),
                                    (14, "molecule")])
def test_ground_state_lanczos(SV, n, kind):
    """ovqe_ground_state (device Lanczos, the stand-in for the reference's dense eigh of fermionic_adapt_vqe.py:474)
    against numpy's dense eigh (n <= 10) / the oracle engine's matrix-free ARPACK run above"""
    from openvqe_amd import fermion
    from tests.oracle_backend import OracleStatevector
    rng = np.random.default_rng(4000 + n)
    if kind == "molecule":
        H, _, _ = fermion.synthetic_molecule(n // 2, max(1, n // 4), seed=n)
    else:
        H = random_hamiltonian(rng, n, min(30, 4 ** n - 1))
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        e, res, its = sv.ground_state(tol=1e-11)
        vec = sv.get_state()
        e_state = sv.expectation(H)
    assert abs(np.linalg.norm(vec) - 1.0) < 1e-12
    assert abs(e_state - e) < 1e-9
    if n <= 10:
        w, v = np.linalg.eigh(H.get_matrix())
        e_ref = w[0]
        ground = v[:, np.abs(w - w[0]) < 1e-9]          # the ground SPACE (degenerate levels)
        overlap = np.linalg.norm(ground.conj().T @ vec) ** 2
    elif n <= 12:
        o = OracleStatevector(n)
        o.set_hamiltonian(H)
        e_ref, _, _ = o.ground_state(tol=1e-12)
        overlap = None
    else:  # an eigenpair by its residual; lowest: not above the Hartree-Fock-like basis-state energies
        e_ref, overlap = e, None
        with SV(n) as sv:
            sv.set_hamiltonian(H)
            for idx in (0, (1 << n) - 1, int(rng.integers(0, 1 << n))):
                sv.init_basis(idx)
                assert sv.expectation(H) >= e - 1e-9
    scale = max(1.0, sum(abs(t.coeff) for t in H.terms))
    assert abs(e - e_ref) < 1e-9 * scale, (e, e_ref, its)
    assert res < 1e-6 * scale
    if overlap is not None:
        assert abs(overlap - 1.0) < 1e-8, overlap


def test_ground_state_one_pass_and_two_pass_agree(SV):
    """ovqe_ground_state with its Lanczos vectors kept in HBM (one pass, default) and with the recurrence run twice
    ("lanczos_keep_gb" = 0): same tridiagonal matrix, same Ritz vector"""
    from openvqe_amd import fermion
    H, _, _ = fermion.synthetic_molecule(8, 4, seed=5)
    out = []
    for keep in (96, 0):
        with SV(16) as sv:
            sv.set_option("lanczos_keep_gb", keep)
            sv.set_hamiltonian(H)
            e, res, its = sv.ground_state(tol=1e-11)
            out.append((e, res, its, sv.get_state()))
    (e1, r1, i1, v1), (e2, r2, i2, v2) = out
    assert i1 == i2 and i1 > 8
    assert abs(e1 - e2) < 1e-11
    assert max(r1, r2) < 1e-6
    assert abs(abs(np.vdot(v1, v2)) - 1.0) < 1e-10
    assert np.max(np.abs(v1 - v2)) < 1e-9


def _fd_gradient(fun, theta, step=1e-5):
    g = np.zeros_like(theta)
    for k in range(len(theta)):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += step; tm[k] -= step
        g[k] = (fun(tp) - fun(tm)) / (2 * step)
    return g


@pytest.mark.parametrize("n,k", [(2, 3), (5, 6), (8, 8), (11, 10), (13, 6), (15, 5)])
def test_adjoint_gradient_of_ucc_programs(SV, n, k):
    """ovqe_energy_gradient (adjoint method: one backward pass) against central differences of the ORACLE energy;
    parameters shared by several rotations, diagonal strings, repeated parameters"""
    from oracle import cref
    from openvqe_amd.backend import compile_ucc_program
    rng = np.random.default_rng(7100 + n)
    H = random_hamiltonian(rng, n, min(40, 4 ** n - 1))
    gens = random_generators(rng, n, k)
    hf = int(rng.integers(0, 1 << n))
    theta = rng.uniform(-0.5, 0.5, k)
    rx, rz, rc, pidx, K = compile_ucc_program(n, gens)
    hx, hz, hc = H.packed()

    def e_oracle(th):
        return cref.ucc_energy(n, hf, rx, rz, rc, pidx, th, hx, hz, hc.real.copy(), H.constant_coeff, 0)[0]

    g_ref = _fd_gradient(e_oracle, theta)
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        sv.set_ucc_program(gens, hf)
        e, g = sv.energy_gradient(theta)
        e2 = sv.energy(theta)
    scale = max(1.0, np.abs(hc).sum())
    assert abs(e - e_oracle(theta)) < 1e-10 * scale and abs(e - e2) < 1e-10 * scale
    assert np.abs(g - g_ref).max() < 2e-8 * scale, (g, g_ref)


@pytest.mark.parametrize("frame", [0, 1])
def test_adjoint_gradient_of_gate_programs(SV, frame):
    """literal gate list (X / H / CNOT un-applied on both states, RX / RY / RZ as rotations) and its Clifford-frame
    form give the same exact gradient; checked against central differences of the plain-C gate-level oracle"""
    from oracle import cref
    from openvqe_amd.backend import GATE_OPCODES
    from tests.util import quccsd_like_gates
    n = 8
    rng = np.random.default_rng(8200)
    gates, K = quccsd_like_gates(rng, n, 3, 4, extra_random=0 if frame else 10, disjoint_ladders=True)
    theta = rng.uniform(-0.7, 0.7, K)
    H = random_hamiltonian(rng, n, 50)
    hf = 0b11010010
    hx, hz, hc = H.packed()
    arrs = ([GATE_OPCODES[g[0]] for g in gates], [n - 1 - g[1][0] for g in gates],
            [n - 1 - g[1][1] if len(g[1]) > 1 else 0 for g in gates], [g[2] for g in gates], [g[3] for g in gates],
            [g[4] for g in gates])

    def e_oracle(th):
        return cref.gate_energy(n, hf, *arrs, th, hx, hz, hc.real.copy(), H.constant_coeff)[0]

    g_ref = _fd_gradient(e_oracle, theta)
    with SV(n) as sv:
        sv.set_option("clifford_frame", frame)
        sv.set_hamiltonian(H)
        sv.set_gate_program(gates, K, hf)
        e, g = sv.energy_gradient(theta)
    scale = max(1.0, np.abs(hc).sum())
    assert abs(e - e_oracle(theta)) < 1e-10 * scale
    assert np.abs(g - g_ref).max() < 2e-8 * scale


@pytest.mark.parametrize("n_local,n_global,shard", [(9, 2, 2), (11, 1, 1), (23, 1, 0)])
def test_shard_screen_primitives_against_oracle(gpu_lib, n_local, n_global, shard):
    """ovqe_apply_pauli_sum / ovqe_bilinear_batch on a shard handle with the PARTNER's shard as ket (x masks carrying one
    global part): the building blocks of the sharded ADAPT screen (openvqe_amd/distributed.py), against the bit-mask
    formulas.  23 local qubits exercises the chunked launch of the batched contraction."""
    import torch
    from openvqe_amd.backend import Statevector
    rng = np.random.default_rng(100 * n_local + shard)
    dim = 1 << n_local
    n = n_local + n_global
    lm = dim - 1
    xg = 1 << n_local                                  # rank difference 1: partner = shard ^ 1
    base, pbase = shard << n_local, (shard ^ 1) << n_local
    own = rng.normal(size=dim) + 1j * rng.normal(size=dim)
    partner = rng.normal(size=dim) + 1j * rng.normal(size=dim)
    sigma = rng.normal(size=dim) + 1j * rng.normal(size=dim)
    T = 14
    xs = [(int(v) & lm) | xg for v in rng.integers(0, 1 << 30, T)]
    zs = [int(v) & ((1 << n) - 1) for v in rng.integers(0, 1 << 30, T)]
    cs = rng.normal(size=T) + 1j * rng.normal(size=T)

    def apply_ref(ket, ket_base, x, z):                # (P ket)_i restricted to this shard: i global = base | i
        i = np.arange(dim, dtype=np.uint64)
        j = i ^ np.uint64(x & lm)
        par = (j | np.uint64(ket_base)) & np.uint64(z)
        for s in (32, 16, 8, 4, 2, 1):
            par ^= par >> np.uint64(s)
        return (1j) ** (bin(x & z).count("1") % 4) * (1.0 - 2.0 * (par & np.uint64(1)).astype(float)) * ket[j.astype(np.int64)]

    t_own = torch.from_numpy(own).cuda()
    t_partner = torch.from_numpy(partner).cuda()
    t_sigma = torch.from_numpy(sigma).cuda()
    t_out = torch.zeros(dim, dtype=torch.complex128, device="cuda")
    with Statevector(n_local, n_global=n_global, shard_index=shard) as sv:
        sv.adopt_state(t_own.data_ptr())
        # out = sum c P partner   (then += the local part: x without the global bit, ket = own state)
        sv.apply_pauli_sum(xs, zs, cs, t_out.data_ptr(), t_partner.data_ptr(), accumulate=False)
        xl = [x & lm for x in xs[:5]]
        sv.apply_pauli_sum(xl, zs[:5], cs[:5], t_out.data_ptr(), None, accumulate=True)
        torch.cuda.synchronize()
        want = sum(c * apply_ref(partner, pbase, x, z) for x, z, c in zip(xs, zs, cs))
        want = want + sum(c * apply_ref(own, base, x, z) for x, z, c in zip(xl, zs[:5], cs[:5]))
        assert np.abs(t_out.cpu().numpy() - want).max() < 1e-11 * T
        offsets = np.array([0, 3, 3, 8, T], np.int64)  # an empty operator in the middle
        got = sv.bilinear_batch(offsets, xs, zs, cs, bra_ptr=t_sigma.data_ptr(), ket_ptr=t_partner.data_ptr())
        ref = [sum(c * np.vdot(sigma, apply_ref(partner, pbase, x, z)) for x, z, c in
                   zip(xs[a:b], zs[a:b], cs[a:b])) for a, b in zip(offsets[:-1], offsets[1:])]
        assert np.abs(got - np.array(ref)).max() < 1e-10 * np.sqrt(dim)


def test_empty_pauli_sums(gpu_lib):
    """ovqe_apply_pauli_sum with T = 0 (out = 0, or untouched when accumulating) and ovqe_bilinear_batch over operators without
    any term (all zeros): the empty host vectors must never reach an upload"""
    import torch
    from openvqe_amd.backend import Statevector
    n = 9
    rng = np.random.default_rng(9)
    fill = torch.from_numpy(rng.normal(size=1 << n) + 1j * rng.normal(size=1 << n)).cuda()
    out = fill.clone()
    with Statevector(n) as sv:
        sv.randomize(5)
        sv.apply_pauli_sum([], [], [], out.data_ptr(), None, accumulate=True)
        torch.cuda.synchronize()
        assert torch.equal(out, fill)
        sv.apply_pauli_sum([], [], [], out.data_ptr(), None, accumulate=False)
        torch.cuda.synchronize()
        assert float(out.abs().max()) == 0.0
        got = sv.bilinear_batch(np.zeros(4, np.int64), [], [], [], bra_ptr=fill.data_ptr())
        assert got.shape == (3,) and np.all(got == 0)


@pytest.mark.parametrize("molecule", ["LIH", "H2O"])
def test_fused_gradient_on_the_compact_support(SV, molecule):
    """ovqe_energy_gradient below 17 qubits: forward circuit, lambda = H psi and the backward pass in ONE launch on the compact
    support (k_sparse_grad) — against the streaming adjoint pass of the same handle (option sparse_grad = 0), against central
    differences of the C oracle on sampled parameters, and the energy against ovqe_energy; also with the bank-aware numbering
    of the support switched off (same numbers up to the order of the additions)"""
    from openvqe_amd import chem, fermion
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    mol = chem.molecule(molecule)
    mol.rhf()
    ham = mol.jw_hamiltonian()
    n = ham.nbqbits
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
    hf = mol.hf_init()
    K = len(gens)
    rng = np.random.default_rng(K)
    theta = rng.uniform(-0.2, 0.2, K)
    out = {}
    for label, opts in (("fused", {}), ("fused_discovery_order", {"sparse_renumber": 0}), ("streaming", {"sparse_grad": 0})):
        with SV(n) as sv:
            for k, v in opts.items():
                sv.set_option(k, v)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            e, g = sv.energy_gradient(theta)
            out[label] = (e, g, sv.energy(theta), sv.program_info())
    assert out["fused"][3]["support"] > 0 and out["fused"][3]["sp_conflicts"] <= out["fused"][3]["sp_conflicts_discovery_order"]
    rx, rz, rc, pidx, _ = compile_ucc_program(n, gens)
    hx, hz, hc = ham.packed()
    scale = max(1.0, float(np.abs(hc).sum()))

    def e_oracle(th):
        return cref.ucc_energy(n, hf, rx, rz, rc, pidx, th, hx, hz, hc.real.copy(), ham.constant_coeff, 0)[0]

    e_ref = e_oracle(theta)
    for label, (e, g, e2, _) in out.items():
        assert abs(e - e_ref) < 1e-10 * scale and abs(e2 - e_ref) < 1e-10 * scale, label
        assert np.abs(g - out["streaming"][1]).max() < 1e-11 * scale, label
    h = 1e-4
    for k in (0, K // 2, K - 1):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += h
        tm[k] -= h
        assert abs(out["fused"][1][k] - (e_oracle(tp) - e_oracle(tm)) / (2 * h)) < 2e-7 * scale


@pytest.mark.parametrize("n,g,chunk_bits", [(14, 1, 12), (15, 2, 11), (13, 2, 10), (12, 3, 7), (16, 1, 15)])
def test_planned_cross_shard_sums_against_oracle(SV, n, g, chunk_bits):
    """The ovqe_xsum_* exports straight through the binding, several shard handles on one GPU standing for the ranks: a Pauli sum
    planned once per shard — <H> (d = 0 terms by the tile cover / pair trick, cross terms by k_tile_cross passes over (partner
    chunk, own shard), chunks of 2^chunk_bits amplitudes: tiles of 2^10 / 2^11 / 2^12 and the streaming fall-back over the cases)
    and sigma = (H + c) psi — against the bit-mask oracle on the whole register; then the error paths of the boundary."""
    import torch
    from openvqe_amd._lib import BackendError
    rng = np.random.default_rng(77 * n + g)
    nl = n - g
    W = 1 << g
    T = 60
    xs = np.array([int(v) for v in rng.integers(0, 1 << n, T)], np.uint64)
    xs[:8] = 0                                                   # a diagonal group
    xs[8:16] &= np.uint64((1 << nl) - 1)                         # local off-diagonal groups
    xs[16:20] = xs[16]                                           # a group of four terms
    zs = np.array([int(v) for v in rng.integers(0, 1 << n, T)], np.uint64)
    even = np.array([bin(int(x) & int(z)).count("1") % 2 == 0 for x, z in zip(xs, zs)])
    zs = np.where(even, zs, zs ^ (xs & (~xs + np.uint64(1))))      # an even number of Y: real coefficients, Hermitian terms
    cs = rng.normal(size=T)
    psi = random_state(rng, n)
    want_e = masks.expectation(psi, xs, zs, cs, 0.0)
    want_sigma = 0.3 * psi + masks.apply_pauli_sum(psi, xs, zs, cs)
    shards = [SV(nl, n_global=g, shard_index=s) for s in range(W)]
    bufs = [torch.from_numpy(psi[s << nl:(s + 1) << nl].copy()).cuda() for s in range(W)]
    outs = [torch.zeros(1 << nl, dtype=torch.complex128, device="cuda") for _ in range(W)]
    try:
        for sv, b in zip(shards, bufs):
            sv.adopt_state(b.data_ptr())
        total = 0.0
        csize = 1 << chunk_bits
        for s, sv in enumerate(shards):
            sid = sv.xsum_create(xs, zs, cs, chunk_bits)
            info = sv.xsum_info(sid)
            partners = sv.xsum_partners(sid)
            assert info["partners"] == len(partners) == len({int(x) >> nl for x in xs if int(x) >> nl})
            assert info["streaming_fallback"] == (1 if chunk_bits < 10 else 0)
            total += sv.xsum_expect_local(sid)
            sv.xsum_apply_local(sid, outs[s].data_ptr(), 0.3)
            for d, _ in partners:
                ket = bufs[s ^ d]
                for c in range(1 << (nl - chunk_bits)):
                    chunk = ket[c * csize:(c + 1) * csize]
                    sv.xsum_expect_remote(sid, d, c, chunk.data_ptr())
                    sv.xsum_apply_remote(sid, d, c, chunk.data_ptr(), outs[s].data_ptr())
            v = sv.xsum_expect_finish(sid)
            total += v.real
            assert abs(sv.xsum_expect_finish(sid)) == 0.0            # the accumulator was reset
            # the boundary says no: a rank difference without terms, a chunk beyond the shard, an unknown plan
            missing = next(d for d in range(1, 2 * W) if d not in [p[0] for p in partners])
            with pytest.raises(BackendError, match="no terms for this rank difference|chunk index"):
                sv.xsum_expect_remote(sid, missing, 0, bufs[s].data_ptr())
            with pytest.raises(BackendError, match="chunk index beyond the shard"):
                sv.xsum_expect_remote(sid, partners[0][0], 1 << (nl - chunk_bits), bufs[s].data_ptr())
            sv.xsum_destroy(sid)
            with pytest.raises(BackendError, match="no such planned sum"):
                sv.xsum_expect_local(sid)
        torch.cuda.synchronize()
        l1 = float(np.abs(cs).sum())
        assert abs(total - want_e) < 1e-11 * l1
        got_sigma = np.concatenate([o.cpu().numpy() for o in outs])
        assert np.abs(got_sigma - want_sigma).max() < 1e-11 * l1
        # complex coefficients: sigma only; an expectation value of a non-Hermitian sum is refused
        sid = shards[0].xsum_create(xs[:4], zs[:4], cs[:4] * (1.0 + 0.5j), chunk_bits)
        with pytest.raises(BackendError, match="complex coefficients"):
            shards[0].xsum_expect_local(sid)
        with pytest.raises(BackendError, match="bits beyond the register"):
            shards[0].xsum_create(np.array([1 << n], np.uint64), np.array([0], np.uint64), np.array([1.0]), chunk_bits)
    finally:
        for sv in shards:
            sv.close()


@pytest.mark.parametrize("n", [9, 15, 17])
def test_real_state_option_on_the_handle_against_oracle(SV, n):
    """Option "real_state" straight through the binding: the adopted buffer holds 2^n DOUBLES — ovqe_init_basis, odd-Y rotations (pair
    sweeps at 9 qubits, real tile sweeps of 2^12 doubles from 14 qubits on), ovqe_norm2 and a planned sum's <H> on 8-byte amplitudes
    against the bit-mask oracle; a rotation that would make the amplitudes complex is refused with the handle left intact"""
    import torch
    from openvqe_amd._lib import BackendError
    rng = np.random.default_rng(31 * n)
    R = 24
    xs, zs = [], []
    for _ in range(R):
        bits = [int(b) for b in rng.choice(n, int(rng.integers(2, 5)), replace=False)]
        x = sum(1 << b for b in bits)
        z = (1 << bits[0]) | int(rng.integers(0, 1 << n)) & ~x          # one Y, Z anywhere else
        xs.append(x)
        zs.append(z)
    xs[5], zs[5] = xs[4], zs[4] ^ (int(rng.integers(0, 1 << n)) & ~xs[4])  # a fusable pair (same x, Z elsewhere differs)
    phis = rng.uniform(-1, 1, R)
    hf = int(rng.integers(0, 1 << n))
    T = 30
    hx = np.array([int(v) for v in rng.integers(0, 1 << n, T)], np.uint64)
    hx[:6] = 0
    hz = np.array([int(v) for v in rng.integers(0, 1 << n, T)], np.uint64)
    odd = np.array([bin(int(x) & int(z)).count("1") % 2 == 1 for x, z in zip(hx, hz)])
    hz = np.where(odd, hz ^ (hx & (~hx + np.uint64(1))), hz)               # even number of Y: a real-symmetric sum
    hc = rng.normal(size=T)
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    assert np.abs(psi.imag).max() < 1e-15
    buf = torch.zeros(1 << n, dtype=torch.float64, device="cuda")
    with SV(n) as sv:
        sv.adopt_state(buf.data_ptr())
        sv.set_option("real_state", 1)
        sv.init_basis(hf)
        sv.apply_pauli_rotations(xs, zs, phis)
        torch.cuda.synchronize()
        assert np.abs(buf.cpu().numpy() - psi.real).max() < 1e-12 and abs(sv.norm2() - 1.0) < 1e-12
        sid = sv.xsum_create(hx, hz, hc, min(n, 12))
        got = sv.xsum_expect_local(sid)
        assert abs(got - masks.expectation(psi, hx, hz, hc, 0.0)) < 1e-11 * np.abs(hc).sum()
        with pytest.raises(BackendError, match="even number of Y"):
            sv.apply_pauli_rotations([3], [0], [0.1])                    # XX: complex amplitudes
        with pytest.raises(BackendError, match="complex amplitudes"):
            sv.xsum_apply_local(sid, buf.data_ptr() + 8, 0.0)            # sigma = H psi needs the complex layout
        torch.cuda.synchronize()
        assert np.abs(buf.cpu().numpy() - psi.real).max() < 1e-12        # nothing was touched
        sv.set_option("real_state", 0)
