"""LDS-tiled multi-op sweeps of the streaming path (openvqe_amd/csrc/sv_tile.hpp): parity with the plain-C oracle
and with the one-sweep-per-op kernels, through the C ABI."""
import numpy as np
import pytest

from tests.util import quccsd_like_gates, random_generators, random_hamiltonian

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def SV(gpu_lib):
    from openvqe_amd.backend import Statevector
    return Statevector


def _gate_arrays(n, gates):
    from openvqe_amd.backend import GATE_OPCODES
    return ([GATE_OPCODES[g[0]] for g in gates], [n - 1 - g[1][0] for g in gates],
            [n - 1 - g[1][1] if len(g[1]) > 1 else 0 for g in gates], [g[2] for g in gates], [g[3] for g in gates],
            [g[4] for g in gates])


@pytest.mark.parametrize("n,tile_bits,tile_low", [(13, 11, 4), (14, 12, 4), (15, 10, 4), (16, 11, 4), (16, 12, 5),
                                                  (17, 11, 6), (17, 10, 0), (18, 11, 4)])
def test_tiled_gate_program_matches_c_oracle(SV, n, tile_bits, tile_low):
    """QUCCSD templates (CNOT ladders, controls inside and outside the tile) + random literal gates"""
    from oracle import cref
    rng = np.random.default_rng(1000 * n + 10 * tile_bits + tile_low)
    gates, K = quccsd_like_gates(rng, n, 4, 5, extra_random=25)
    theta = rng.uniform(-1, 1, K)
    H = random_hamiltonian(rng, n, 60)
    hf = int(rng.integers(0, 1 << n))
    hx, hz, hc = H.packed()
    e_ref, psi_ref = cref.gate_energy(n, hf, *_gate_arrays(n, gates), theta, hx, hz, hc.real.copy(), H.constant_coeff)
    idx = rng.integers(0, 1 << n, 4000).astype(np.uint64)
    out = {}
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_hamiltonian(H)
        for bits in (0, tile_bits):
            sv.set_option("clifford_frame", 0)                # the literal list, gate by gate
            sv.set_option("tile_low", tile_low)
            sv.set_option("tile_bits", bits)
            sv.set_gate_program(gates, K, hf)
            assert sv.program_info()["literal_gates"] > 0
            e = sv.energy(theta)
            sv.prepare_state(theta)
            out[bits] = (e, sv.get_amplitudes(idx), sv.norm2())
        # default: the random gates leave the Clifford frame OPEN — the program is the Pauli-rotation sequence, energies use the
        # Hamiltonian conjugated by the net Clifford operator, ovqe_prepare_state applies the Clifford gates behind the rotations
        sv.set_option("clifford_frame", 1)
        sv.set_gate_program(gates, K, hf)
        info = sv.program_info()
        e = sv.energy(theta)
        eg, _ = sv.energy_gradient(theta)
        sv.prepare_state(theta)
        out["open_frame"] = (e, sv.get_amplitudes(idx), sv.norm2())
        e_after = sv.energy(theta)                            # ... and the state calls leave the evaluation path intact
    assert info["literal_gates"] == 0
    assert abs(eg - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum()) and abs(e_after - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum())
    for bits, (e, amps, n2) in out.items():
        assert abs(e - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum()), bits
        assert np.abs(amps - psi_ref[idx.astype(np.int64)]).max() < 1e-12, bits
        assert abs(n2 - 1.0) < 1e-11
    assert np.abs(out[0][1] - out[tile_bits][1]).max() < 1e-14


@pytest.mark.parametrize("n,tile_bits", [(14, 11), (16, 10), (18, 12), (20, 11)])
def test_tiled_rotation_program_matches_c_oracle(SV, n, tile_bits):
    """Pauli-rotation programs: same-x runs, diagonal strings, z chains crossing the tile boundary, wide x masks that
    keep their own sweep"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    rng = np.random.default_rng(77 * n + tile_bits)
    gens = fermion.uccsd_generators(n // 2, 2)[::5][:25] + random_generators(rng, n, 10)
    order = rng.permutation(len(gens))
    gens = [gens[i] for i in order]
    hf = fermion.hf_integer(n, 4)
    H = random_hamiltonian(rng, n, 80)
    theta = rng.uniform(-0.4, 0.4, len(gens))
    rx, rz, rc, pidx, K = compile_ucc_program(n, gens)
    hx, hz, hc = H.packed()
    e_ref, psi_ref = cref.ucc_energy(n, hf, rx, rz, rc, pidx, theta, hx, hz, hc.real.copy(), H.constant_coeff, 0)
    idx = rng.integers(0, 1 << n, 4000).astype(np.uint64)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("tile_bits", tile_bits)
        sv.set_hamiltonian(H)
        sv.set_ucc_program(gens, hf)
        e = sv.energy(theta)
        sv.prepare_state(theta)
        amps = sv.get_amplitudes(idx)
    assert abs(e - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum())
    assert np.abs(amps - psi_ref[idx.astype(np.int64)]).max() < 1e-12


@pytest.mark.parametrize("n,ns,nd", [(4, 2, 1), (8, 4, 6), (12, 6, 10), (14, 6, 12), (16, 5, 8)])
def test_quccsd_templates_compile_to_pauli_rotations(SV, n, ns, nd):
    """Clifford-frame form of the reference's QUCCSD gate list: the frame closes after every template, so NO literal
    gate is left, and state + energy equal the literal gate-by-gate execution (C oracle and clifford_frame = 0)"""
    from oracle import cref
    rng = np.random.default_rng(4242 + n)
    gates, K = quccsd_like_gates(rng, n, ns, nd, disjoint_ladders=True)
    theta = rng.uniform(-1, 1, K)
    H = random_hamiltonian(rng, n, 50)
    hf = int(rng.integers(0, 1 << n))
    hx, hz, hc = H.packed()
    e_ref, psi_ref = cref.gate_energy(n, hf, *_gate_arrays(n, gates), theta, hx, hz, hc.real.copy(), H.constant_coeff)
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        res = {}
        for mode in (0, 1):
            sv.set_option("clifford_frame", mode)
            sv.set_gate_program(gates, K, hf)
            info = sv.program_info()
            e = sv.energy(theta)
            eb = sv.energy_batch(np.stack([theta, 0.5 * theta, -theta]))
            sv.prepare_state(theta)
            res[mode] = (info, e, eb, sv.get_state())
    info0, info1 = res[0][0], res[1][0]
    assert info0["literal_gates"] > 0 and info1["literal_gates"] == 0, (info0, info1)
    assert info1["rotations"] == 2 * ns + 8 * nd
    scale = max(1.0, np.abs(hc).sum())
    for mode in (0, 1):
        assert abs(res[mode][1] - e_ref) < 1e-10 * scale
        assert abs(res[mode][2][0] - e_ref) < 1e-10 * scale
        assert np.abs(res[mode][3] - psi_ref).max() < 1e-12
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-10 * scale


@pytest.mark.parametrize("n,ns,nd", [(8, 4, 6), (14, 6, 12)])
def test_clifford_phase_from_host_simulation_equals_device_run(SV, n, ns, nd):
    """the global phase of a closed frame's Clifford part: sparse simulation on the host (default) against the gates run as a
    literal program on the device (clifford_phase_host = 0) — the prepared states are equal INCLUDING the phase"""
    rng = np.random.default_rng(777 + n)
    gates, K = quccsd_like_gates(rng, n, ns, nd, disjoint_ladders=True)
    theta = rng.uniform(-1, 1, K)
    hf = int(rng.integers(0, 1 << n))
    states = {}
    with SV(n) as sv:
        for host in (1, 0):
            sv.set_option("clifford_phase_host", host)
            sv.set_gate_program(gates, K, hf)
            assert sv.program_info()["literal_gates"] == 0
            sv.prepare_state(theta)
            states[host] = sv.get_state()
    assert np.abs(states[0] - states[1]).max() < 1e-12
    assert abs(np.vdot(states[0], states[0]) - 1.0) < 1e-10


@pytest.mark.parametrize("n,seed", [(3, 0), (5, 1), (6, 2), (9, 3), (13, 4)])
def test_forced_clifford_frame_on_random_circuits(SV, n, seed):
    """the conjugation algebra (X, H, CNOT, RX/RY/RZ(+-pi/2) folded into the frame; every other rotation re-expressed
    in it, the Clifford part appended literally) on random circuits, against the Kronecker / plain-C oracles"""
    from oracle import cref
    rng = np.random.default_rng(900 + seed)
    K = 5
    gates = []
    for _ in range(80):
        name = str(rng.choice(["X", "H", "RX", "RY", "RZ", "CNOT", "Q"]))
        if name == "CNOT":
            c, t = rng.choice(n, 2, replace=False).tolist()
            gates.append((name, [c, t], 0.0, 0.0, -1))
        elif name in ("X", "H"):
            gates.append((name, [int(rng.integers(0, n))], 0.0, 0.0, -1))
        elif name == "Q":  # quarter turns: folded
            gates.append((str(rng.choice(["RX", "RY", "RZ"])), [int(rng.integers(0, n))], 0.0,
                          float(rng.choice([np.pi / 2, -np.pi / 2])), -1))
        else:
            gates.append((name, [int(rng.integers(0, n))], float(rng.choice([1.0, -1.0, -2.0])),
                          float(rng.uniform(-1, 1)), int(rng.integers(-1, K))))
    theta = rng.uniform(-1, 1, K)
    H = random_hamiltonian(rng, n, min(40, 4 ** n - 1))
    hf = int(rng.integers(0, 1 << n))
    hx, hz, hc = H.packed()
    e_ref, psi_ref = cref.gate_energy(n, hf, *_gate_arrays(n, gates), theta, hx, hz, hc.real.copy(), H.constant_coeff)
    with SV(n) as sv:
        sv.set_hamiltonian(H)
        for mode in (2, 1, 0):
            sv.set_option("clifford_frame", mode)
            sv.set_gate_program(gates, K, hf)
            assert abs(sv.energy(theta) - e_ref) < 1e-10 * max(1.0, np.abs(hc).sum()), mode
            sv.prepare_state(theta)
            assert np.abs(sv.get_state() - psi_ref).max() < 1e-12, mode


@pytest.mark.parametrize("m,o,tile_bits,tile_low", [(7, 3, 11, 4), (8, 3, 10, 4), (8, 2, 12, 3), (9, 4, 11, 5)])
def test_tiled_expectation_of_jw_hamiltonian(SV, m, o, tile_bits, tile_low):
    """<psi|H|psi> through the tile cover of the x-groups (weight-0/2/4 masks of a JW two-body Hamiltonian, z chains
    crossing the tiles) against the one-sweep-per-group kernel and the plain-C oracle"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=100 + n)
    gens = gens[::3]
    rng = np.random.default_rng(n)
    theta = rng.uniform(-0.3, 0.3, len(gens))
    rx, rz, rc, pidx, K = compile_ucc_program(n, gens)
    hx, hz, hc = ham.packed()
    e_ref, _ = cref.ucc_energy(n, hf, rx, rz, rc, pidx, theta, hx, hz, hc.real.copy(), ham.constant_coeff, 0)
    es = {}
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("tile_low", tile_low)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        for bits in (0, tile_bits):
            sv.set_option("tile_bits", bits)
            es[bits] = sv.energy(theta)
    scale = max(1.0, np.abs(hc).sum())
    assert abs(es[0] - e_ref) < 1e-10 * scale
    assert abs(es[tile_bits] - e_ref) < 1e-10 * scale
    assert abs(es[tile_bits] - es[0]) < 1e-11 * scale


@pytest.mark.parametrize("n,g", [(14, 0), (16, 0), (15, 1), (16, 2)])
def test_tiled_rotation_lists_incl_shard_handles(SV, n, g):
    """ovqe_apply_pauli_rotations on a random state: lists of low-weight strings (several same-x runs per tile sweep)
    mixed with wide ones, on a plain handle and on shard handles (z on rank bits = per-shard sign)"""
    from oracle import masks
    from tests.util import random_state
    rng = np.random.default_rng(1234 + 10 * n + g)
    nl = n - g
    psi = random_state(rng, n)
    xs, zs, phis = [], [], []
    ref = psi.copy()
    for k in range(36):
        if k % 9 == 8:
            x = int(rng.integers(0, 1 << nl))                      # wide: keeps its own sweep
        else:
            bits = rng.choice(nl, int(rng.choice([1, 2, 4])), replace=False)
            x = int(sum(1 << int(b) for b in bits))
            if rng.random() < 0.15:
                x = 0                                              # diagonal string
        z = int(rng.integers(0, 1 << n))
        phi = float(rng.uniform(-1, 1))
        reps = 1 + int(rng.integers(0, 3))
        for _ in range(reps):                                      # same-x run with different z
            xs.append(x); zs.append(z ^ int(rng.integers(0, 1 << n))); phis.append(phi)
            ref = masks.rotate(ref, xs[-1], zs[-1], phis[-1])
    shards = [SV(nl, n_global=g, shard_index=s) for s in range(1 << g)]
    try:
        for s, sv in enumerate(shards):
            sv.set_state(psi[s << nl:(s + 1) << nl])
            sv.apply_pauli_rotations(xs, zs, phis)
        got = np.concatenate([sv.get_state() for sv in shards])
    finally:
        for sv in shards:
            sv.close()
    assert np.abs(got - ref).max() < 1e-12


@pytest.mark.parametrize("n,g", [(14, 0), (16, 0), (15, 1)])
def test_tiled_adhoc_expectation_incl_shard_handles(SV, n, g):
    """ovqe_expectation / ovqe_bilinear on the handle's own state (Hermitian sum, local x): tile cover, cached between
    calls, z on rank bits as per-shard signs; against the bit-mask oracle on a random state"""
    from openvqe_amd import fermion
    from oracle import masks
    from tests.util import random_state
    rng = np.random.default_rng(99 + n + g)
    nl = n - g
    psi = random_state(rng, n)
    ham, _, _ = fermion.synthetic_molecule(nl // 2, 2, seed=n)          # JW strings on the local qubits ...
    xs, zs, cs = ham.packed()
    cs = cs.real.copy()
    xs = xs.astype(np.uint64)
    zs = (zs ^ (rng.integers(0, 1 << g, len(zs)).astype(np.uint64) << np.uint64(nl))) if g else zs   # ... z also on rank bits
    want = masks.expectation(psi, xs, zs, cs)
    shards = [SV(nl, n_global=g, shard_index=s) for s in range(1 << g)]
    try:
        got = []
        for rep in range(2):                                               # second call: cached cover
            tot = 0.0
            for s, sv in enumerate(shards):
                sv.set_state(psi[s << nl:(s + 1) << nl])
                tot += sv.bilinear(xs, zs, cs).real
            got.append(tot)
    finally:
        for sv in shards:
            sv.close()
    scale = np.abs(cs).sum()
    assert abs(got[0] - want) < 1e-11 * scale and got[0] == got[1]


@pytest.mark.parametrize("m,o,bits", [(8, 3, -1), (9, 2, 10), (10, 4, -1), (8, 2, 12)])
def test_real_amplitude_streaming_matches_complex_path_and_oracle(SV, m, o, bits):
    """UCC programs (every rotation string has an odd number of Y) stream 8 bytes per amplitude (real_stream = 1,
    default): energies against the complex streaming path and the plain-C oracle; wide strings and a Hamiltonian
    group that fit no tile included; prepare_state afterwards still delivers the complex state"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import compile_ucc_program
    from openvqe_amd.operators import Hamiltonian, Term
    from oracle import cref
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=300 + n)
    rng = np.random.default_rng(n + 5)
    gens = gens[::4]
    wide_qs = sorted(rng.choice(n, 9, replace=False).tolist())
    gens.insert(2, Hamiltonian(n, [Term(0.8, "XXXYXXXXX", wide_qs)], do_clean_up=False))   # one Y: real, wide x mask
    ham = Hamiltonian(n, list(ham.terms) + [Term(0.21, "X" * 10, sorted(rng.choice(n, 10, replace=False).tolist()))],
                      ham.constant_coeff)
    theta = rng.uniform(-0.3, 0.3, len(gens))
    rx, rz, rc, pidx, K = compile_ucc_program(n, gens)
    hx, hz, hc = ham.packed()
    e_ref, psi_ref = cref.ucc_energy(n, hf, rx, rz, rc, pidx, theta, hx, hz, hc.real.copy(), ham.constant_coeff, 0)
    idx = rng.integers(0, 1 << n, 3000).astype(np.uint64)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("tile_bits", bits)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        es = {}
        for real in (1, 0, 1, 1):                                           # from its second real evaluation on, an
            sv.set_option("real_stream", real)                             # 18+ qubit program may run on the sector path
            es.setdefault(real, []).append(sv.energy(theta))
        sv.prepare_state(theta)
        amps = sv.get_amplitudes(idx)
    scale = max(1.0, np.abs(hc).sum())
    assert abs(es[1][0] - e_ref) < 1e-10 * scale and abs(es[0][0] - e_ref) < 1e-10 * scale
    assert abs(es[1][0] - es[1][1]) < 1e-13 * scale and es[1][1] == es[1][2]   # same path twice: same bits
    assert np.abs(amps - psi_ref[idx.astype(np.int64)]).max() < 1e-12


@pytest.mark.parametrize("m,o,bits", [(7, 3, -1), (8, 2, 10), (8, 3, 12)])
def test_tiled_operator_application_in_screen_and_lanczos(SV, m, o, bits):
    """sigma = H psi through the tile cover (k_tile_apply): the ADAPT gradient screen against the bit-mask oracle and
    against the gather kernel (tile_bits = 0); Lanczos on top of it reaches an eigenpair (residual)"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import GRAD_FERMIONIC
    from openvqe_amd.operators import pack_terms
    from oracle import masks
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=700 + n)
    pool = fermion.uccsd_pool_antihermitian(m, o)[::9][:40]
    rng = np.random.default_rng(n)
    theta = rng.uniform(-0.3, 0.3, len(gens))
    res = {}
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("apply_min_tiles", 1)   # small registers use the gather kernel by default: force the tile form
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        for b in (bits, 0):
            sv.set_option("tile_bits", b)
            sv.prepare_state(theta)
            psi = sv.get_state()
            res[b] = sv.pool_gradients(pool, GRAD_FERMIONIC)
        sv.set_option("tile_bits", bits)
        e, resid, its = sv.ground_state(tol=1e-10)
    hx, hz, hc = ham.packed()
    sigma = masks.apply_pauli_sum(psi, hx, hz, hc) + ham.constant_coeff * psi
    want = []
    for op in pool:
        px, pz, pc = pack_terms(n, op.terms)
        want.append(2.0 * np.vdot(sigma, masks.apply_pauli_sum(psi, px, pz, pc)).real)
    scale = max(1.0, np.abs(hc).sum())
    assert np.abs(res[bits] - np.array(want)).max() < 1e-10 * scale
    assert np.abs(res[bits] - res[0]).max() < 1e-11 * scale
    assert resid < 1e-6 * scale


@pytest.mark.parametrize("real", [False, True])
def test_dense_register_expectation_with_walsh_hadamard_diagonal_against_oracle(gpu_lib, real):
    """A DENSE 25-qubit state (the synthetic state of the sharded legs; real = True: its real parts on a float64 shard): <H> of a sum with
    a 90-string diagonal group — evaluated per contiguous tile by a Walsh-Hadamard transform (k_tile_diag) while the cover's sweeps skip
    their x = 0 entries — and 40 off-diagonal strings (entries with the trip part of the pair index from the host, unsplit entries of
    1024 pairs) against the bit-mask oracle on the same amplitudes"""
    import torch

    import bench
    from openvqe_amd import synth
    from openvqe_amd.distributed import ShardedStatevector
    n = 25
    rng = np.random.default_rng(2525 + int(real))
    hx, hz, hc = [], [], []
    for _ in range(90):                       # diagonal strings: random Z sets of weight 1 .. 6, some reaching the top (per-tile sign) bits
        qs = rng.choice(n, int(rng.integers(1, 7)), replace=False)
        hx.append(0)
        hz.append(sum(1 << int(q) for q in qs))
        hc.append(float(rng.normal()))
    for _ in range(40):                       # JW-like off-diagonal strings with an even number of Y (a real-symmetric sum)
        x, z = bench.two_body_string(rng, n)
        if bin(x & z).count("1") & 1:
            z ^= x & -x
        hx.append(x)
        hz.append(z)
        hc.append(float(rng.normal()))
    psi = synth.amplitudes(bench.SHARDED_SEED, np.arange(1 << n, dtype=np.uint64))
    if real:
        psi = psi.real.astype(complex)
    psi = psi / np.linalg.norm(psi)
    # the C oracle's x-grouped expectation (OpenMP; the numpy oracle needs a minute per case at this size; the two are pinned against each
    # other in tests/test_oracle.py)
    from oracle import cref
    sx, sz, sc = cref.sort_by_x(np.array(hx, np.uint64), np.array(hz, np.uint64), np.array(hc, np.float64))
    want = cref.lib().orc_expectation_grouped(np.ascontiguousarray(psi), n, len(sx), sx, sz, sc) + 0.25
    sv = ShardedStatevector(n, device=0)      # (one rank without a process group: the shard is the register)
    sv.randomize(bench.SHARDED_SEED)
    if real:
        sv.engine.set_real(True)
        sv.real = True
        sv.engine.tensor.mul_(1.0 / sv.norm2() ** 0.5)
    got = sv.expectation(hx, hz, hc, 0.25)
    again = sv.expectation(hx, hz, hc, 0.25)
    assert sv._storage_real() == real
    l1 = float(np.abs(hc).sum())
    assert abs(got - want) < 1e-11 * l1 and got == again, (got, want)
    del sv
    torch.cuda.empty_cache()
