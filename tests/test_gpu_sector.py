"""Sector path (openvqe_amd/csrc/sv_sector.hpp): circuit + materialised <H> on the support of the program's states.
Parity with the plain-C oracle and with the dense streaming kernels, through the C ABI.  The tables are built at the second
evaluation of a (program, Hamiltonian) pair, so every test evaluates at least three parameter vectors."""
from math import comb

import numpy as np
import pytest

from tests.util import random_hamiltonian, random_string

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def SV(gpu_lib):
    from openvqe_amd.backend import Statevector
    return Statevector


def _oracle_energies(n, gens, hf, H, thetas):
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    rx, rz, rc, pidx, _ = compile_ucc_program(n, gens)
    hx, hz, hc = H.packed()
    return [cref.ucc_energy(n, hf, rx, rz, rc, pidx, th, hx, hz, hc.real.copy(), H.constant_coeff, 0)[0] for th in thetas]


@pytest.mark.parametrize("m,o,bits,threads", [(7, 2, 0, 256), (7, 3, 8, 256), (8, 3, 0, 64), (8, 4, 10, 256), (9, 4, 0, 256),
                                              (10, 4, 12, 256), (10, 5, 0, 256), (10, 4, 18, 0)])
def test_sector_uccsd_matches_c_oracle(SV, m, o, bits, threads):
    """molecule-shaped UCCSD (JW generators in table-fused form, a JW two-body Hamiltonian): the support is the
    (o alpha, o beta) sector; energies of the sector path == oracle == dense kernels.  bits = 18 at 20 qubits: tiles of 17
    index bits (8 tiles of ~5500 amplitudes), beyond the dense slot maps of the table construction (binary-search path)"""
    from openvqe_amd import fermion
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=100 + m)
    rng = np.random.default_rng(10 * m + o)
    thetas = [rng.uniform(-0.3, 0.3, len(gens)) for _ in range(4)]
    thetas.append(np.zeros(len(gens)))                      # |hf> itself: every rotation is the identity
    want = _oracle_energies(n, gens, hf, ham, thetas)
    l1 = float(np.abs(ham.packed()[2]).sum())
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_bits", bits)
        sv.set_option("sector_threads", threads)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        got = [sv.energy(th) for th in thetas]
        info = sv.program_info()
        batch = sv.energy_batch(np.stack(thetas))
        sv.set_option("sector", 0)
        dense = [sv.energy(th) for th in thetas]
        assert sv.program_info()["sector_support"] == 0
    assert info["sector_support"] == comb(m, o) ** 2, info
    assert info["sector_sweeps"] >= 1 and info["sector_h_sweeps"] >= 1 and info["sector_pairs"] > 0
    assert info["sector_h_elements"] > info["sector_support"]
    for e, eb, ed, ew in zip(got, batch, dense, want):
        assert abs(e - ew) < 1e-10 * max(1.0, l1)
        assert abs(eb - ew) < 1e-10 * max(1.0, l1)
        assert abs(e - ed) < 1e-12 * max(1.0, l1)


@pytest.mark.parametrize("m,o,bits", [(8, 3, 0), (9, 4, 10), (10, 4, 0)])
def test_pair_table_builder_forms_give_the_same_tables(SV, m, o, bits):
    """k_sec_pairs2 (waves own contiguous entry ranges, ops staged in LDS) against the first form of the builder: the pairs of an op
    come out in the same order, so energies are EQUAL and the pair counts too"""
    from openvqe_amd import fermion
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=900 + m)
    theta = np.random.default_rng(m).uniform(-0.3, 0.3, len(gens))
    got = {}
    for form in (1, 2):
        with SV(2 * m) as sv:
            sv.set_option("sector_min_qubits", 8)
            sv.set_option("sector_pairs_form", form)
            if bits:
                sv.set_option("sector_bits", bits)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            e = [sv.energy(theta) for _ in range(3)][-1]
            eg, g = sv.energy_gradient(theta)
            info = sv.program_info()
            assert info["sector_support"] > 0
            got[form] = (e, eg, g, info["sector_pairs"], info["sector_sweeps"])
    assert got[1][0] == got[2][0] and got[1][3] == got[2][3] and got[1][4] == got[2][4]
    assert abs(got[1][1] - got[2][1]) < 1e-13 and np.abs(got[1][2] - got[2][2]).max() < 1e-13


def test_dictionary_from_a_sample_falls_back_to_all_values(SV):
    """the dictionary of the coded matrix elements is built from a sample of the stream first; k_sec_encode reports a magnitude the
    sample missed and the dictionary is rebuilt from all values — forced here with a sample of every 4096th value (sector_dict = 3):
    the tables, hence the energies, equal those of the full sort (sector_dict = 2)"""
    from openvqe_amd import fermion
    ham, gens, hf = fermion.synthetic_molecule(9, 4, seed=31)
    thetas = np.random.default_rng(2).uniform(-0.3, 0.3, (2, len(gens)))
    got = {}
    for mode in (2, 3, 1):
        with SV(18) as sv:
            sv.set_option("sector_dict", mode)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            sv.energy(thetas[0])
            got[mode] = [sv.energy(t) for t in thetas] + [sv.energy_gradient(thetas[1])[0]]
            info = sv.program_info()
            assert info["sector_support"] > 0 and info["sector_h_sweeps"] > 0
    assert got[2][:2] == got[3][:2] == got[1][:2]
    assert abs(got[2][2] - got[3][2]) < 1e-13


def test_final_reduction_into_mapped_memory_equals_reduce_and_copies(SV):
    """k_sector_finish (energy + orphan flag written into mapped host memory) sums like k_reduce: equal energies"""
    from openvqe_amd import fermion
    ham, gens, hf = fermion.synthetic_molecule(9, 4, seed=77)
    thetas = np.random.default_rng(9).uniform(-0.3, 0.3, (3, len(gens)))
    got = {}
    for fused in (1, 0):
        with SV(18) as sv:
            sv.set_option("sector_fused_reduce", fused)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            sv.energy(thetas[0])
            got[fused] = [sv.energy(t) for t in thetas]
            assert sv.program_info()["sector_support"] > 0
    assert got[0] == got[1]


@pytest.mark.parametrize("m,o", [(9, 4), (10, 5)])
def test_sector_circuit_with_compact_cover_expectation(SV, m, o):
    """sector_h = 0 (what happens when the materialised <H> would not fit the table budget): the circuit runs on the sector
    tables, <H> is evaluated by the compact cover of sv_tile.hpp from the canonical compact state"""
    from openvqe_amd import fermion
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=200 + m)
    rng = np.random.default_rng(20 * m + o)
    thetas = [rng.uniform(-0.3, 0.3, len(gens)) for _ in range(4)]
    want = _oracle_energies(n, gens, hf, ham, thetas)
    l1 = float(np.abs(ham.packed()[2]).sum())
    with SV(n) as sv:
        sv.set_option("sector_h", 0)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        got = [sv.energy(th) for th in thetas]
        info = sv.program_info()
    assert info["sector_support"] == comb(m, o) ** 2 and info["sector_sweeps"] >= 1, info
    assert info["sector_h_sweeps"] == 0 and info["sector_h_elements"] == 0
    for e, ew in zip(got, want):
        assert abs(e - ew) < 1e-10 * max(1.0, l1)


def test_sector_on_h2o_uccsd_and_table_invalidation(SV):
    """H2O / STO-3G UCCSD (14 qubits) at the MP2 amplitudes through the sector path; then a new Hamiltonian and a new
    program on the same handle: the tables are rebuilt for them"""
    from openvqe_amd import chem, fermion
    mol = chem.molecule("H2O")
    mol.rhf()
    prob = mol.problem(active=False)
    ham = prob.jw_hamiltonian()
    _, _, gens, theta_mp2, hf = prob.uccsd()
    n = ham.nbqbits
    rng = np.random.default_rng(14)
    thetas = [np.array(theta_mp2), np.array(theta_mp2) * 0.5, rng.uniform(-0.2, 0.2, len(gens))]
    want = _oracle_energies(n, gens, hf, ham, thetas)
    ham2 = fermion.synthetic_molecule(7, 5, seed=3)[0]       # another JW two-body operator on the same register
    gens2 = gens[::3]
    thetas2 = [t[::3] for t in thetas]
    want2 = _oracle_energies(n, gens2, hf, ham2, thetas2)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        got = [sv.energy(t) for t in thetas]
        info = sv.program_info()
        sv.set_hamiltonian(ham2)
        mixed = [sv.energy(t) for t in thetas]          # same program, other Hamiltonian
        sv.set_ucc_program(gens2, hf)
        got2 = [sv.energy(t) for t in thetas2]
        info2 = sv.program_info()
    assert info["sector_support"] == 441 and 0 < info2["sector_support"] <= 441
    want_mixed = _oracle_energies(n, gens, hf, ham2, thetas)
    for e, ew in zip(got + mixed + got2, want + want_mixed + want2):
        assert abs(e - ew) < 1e-10


def test_sector_with_single_string_rotations(SV):
    """OP_PAIR ops (single Pauli strings with an odd number of Y, no table fusion): x masks from a small set keep the support
    at a few hundred of 2^16 amplitudes; several rotations share an x mask with different z masks"""
    from openvqe_amd.operators import Hamiltonian, Term
    n = 16
    rng = np.random.default_rng(1616)
    supports = [[0, 5], [3, 9, 12, 15], [1, 2], [7, 8, 10, 11], [4, 13], [0, 5], [3, 9, 12, 15], [6, 14]]
    gens = []
    for qs in supports * 2:
        ops = ["X"] * len(qs)
        for k in rng.choice(len(qs), 1 if len(qs) == 2 else int(rng.choice([1, 3])), replace=False):
            ops[k] = "Y"
        zq = [q for q in range(n) if q not in qs and rng.random() < 0.4]
        gens.append(Hamiltonian(n, [Term(float(rng.uniform(0.5, 1.5)), "".join(ops) + "Z" * len(zq), qs + zq)], do_clean_up=False))
    hf = int(rng.integers(0, 1 << n))
    seen, terms = set(), []
    while len(terms) < 150:                                  # strings of weight <= 5: every x-group fits a sector tile
        op, qs = random_string(rng, n, 1, 5)
        if (op, tuple(qs)) not in seen:
            seen.add((op, tuple(qs)))
            terms.append(Term(float(rng.normal()), op, qs))
    H = Hamiltonian(n, terms, 0.25)
    thetas = [rng.uniform(-1, 1, len(gens)) for _ in range(4)]
    want = _oracle_energies(n, gens, hf, H, thetas)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(H)
        sv.set_ucc_program(gens, hf)
        got = [sv.energy(t) for t in thetas]
        info = sv.program_info()
    assert info["real_stream"] == 1
    assert 0 < info["sector_support"] <= 256, info
    for e, ew in zip(got, want):
        assert abs(e - ew) < 1e-10 * max(1.0, float(np.abs(H.packed()[2]).sum()))


def test_sector_declines_a_dense_support(SV):
    """a program whose states fill the register: no tables are built, the dense kernels keep serving the energies"""
    from openvqe_amd.operators import Hamiltonian, Term
    n = 14
    rng = np.random.default_rng(1414)
    gens = [Hamiltonian(n, [Term(1.0, "Y", [q])], do_clean_up=False) for q in range(n)]
    H = random_hamiltonian(rng, n, 60)
    thetas = [rng.uniform(-1, 1, n) for _ in range(3)]
    want = _oracle_energies(n, gens, 0, H, thetas)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(H)
        sv.set_ucc_program(gens, 0)
        got = [sv.energy(t) for t in thetas]
        info = sv.program_info()
    assert info["sector_support"] == 0 and info["sector_bytes"] == 0
    for e, ew in zip(got, want):
        assert abs(e - ew) < 1e-10 * max(1.0, float(np.abs(H.packed()[2]).sum()))


def test_sector_table_budget(SV):
    """sector_max_gb = 0: the tables do not fit the budget, nothing is kept, energies come from the dense kernels"""
    from openvqe_amd import fermion
    ham, gens, hf = fermion.synthetic_molecule(8, 3, seed=5)
    rng = np.random.default_rng(5)
    thetas = [rng.uniform(-0.3, 0.3, len(gens)) for _ in range(3)]
    want = _oracle_energies(16, gens, hf, ham, thetas)
    with SV(16) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_max_gb", 0)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        got = [sv.energy(t) for t in thetas]
        assert sv.program_info()["sector_support"] == 0
    for e, ew in zip(got, want):
        assert abs(e - ew) < 1e-10 * max(1.0, float(np.abs(ham.packed()[2]).sum()))


@pytest.mark.parametrize("m,o,bits,coded", [(7, 3, 0, 1), (8, 3, 9, 1), (9, 4, 0, 1), (10, 4, 0, 1), (8, 4, 0, 0), (9, 3, 10, 0)])
def test_sector_adjoint_gradient(SV, m, o, bits, coded):
    """ovqe_energy_gradient on the sector tables (forward circuit, lambda = H psi on the support, one backward pass):
    against the dense-state adjoint pass of the same handle and central differences of the C oracle's energy"""
    from openvqe_amd import fermion
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=400 + m)
    rng = np.random.default_rng(40 * m + o)
    K = len(gens)
    th1, th2 = rng.uniform(-0.3, 0.3, K), rng.uniform(-0.3, 0.3, K)
    l1 = float(np.abs(ham.packed()[2]).sum())
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_bits", bits)
        sv.set_option("sector_dict", coded)        # 0: explicit doubles for every matrix element
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        e1, g1 = sv.energy_gradient(th1)          # the first gradient call builds the tables and runs on them
        e2, g2 = sv.energy_gradient(th1)
        e3, g3 = sv.energy_gradient(th2)
        e3b = sv.energy(th2)
        info = sv.program_info()
        sv.set_option("sector_adjoint", 1)        # the first form of the backward sweeps (the default takes the 64-bit tables)
        e5, g5 = sv.energy_gradient(th2)
        sv.set_option("sector", 0)
        e4, g4 = sv.energy_gradient(th2)
    assert info["sector_support"] == comb(m, o) ** 2 and info["sector_h_elements"] > 0, info
    assert abs(e2 - e1) < 1e-12 * l1 and abs(e3 - e4) < 1e-12 * l1 and abs(e3 - e3b) < 1e-12 * l1
    assert np.abs(g2 - g1).max() < 1e-11 * l1 and np.abs(g3 - g4).max() < 1e-11 * l1
    assert abs(e5 - e3) < 1e-13 * l1 and np.abs(g5 - g3).max() < 1e-12 * l1   # (lambda = H psi adds with atomics: to rounding)
    picks = rng.choice(K, 4, replace=False)
    h = 1e-5
    for k in picks:
        tp, tm = th2.copy(), th2.copy()
        tp[k] += h
        tm[k] -= h
        ep, em = _oracle_energies(n, gens, hf, ham, [tp, tm])
        assert abs((ep - em) / (2 * h) - g3[k]) < 1e-6 * max(1.0, l1), k


@pytest.mark.parametrize("m,o", [(7, 3), (8, 3), (9, 4)])
def test_sector_on_the_reference_quccsd_templates(SV, m, o):
    """the reference's QUCCSD gate list (ref:openvqe/common_files/circuit.py templates, Clifford-frame form): its double
    templates do not conserve the particle number, so the support is a quarter of the register (spin-parity sectors), ops
    with up to 8 active patterns — energies against the C oracle's gate-by-gate execution, gradients against the
    dense-state adjoint pass"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import GATE_OPCODES
    from openvqe_amd.common_files.circuit import quccsd_gate_list
    from oracle import cref
    n = 2 * m
    ham, _, hf = fermion.synthetic_molecule(m, o, seed=500 + m)
    gates, K, hf2 = quccsd_gate_list(m, o, 2)
    assert hf2 == hf
    rng = np.random.default_rng(50 * m + o)
    thetas = [rng.uniform(-0.4, 0.4, K) for _ in range(3)]
    hx, hz, hc = ham.packed()
    opc = [GATE_OPCODES[g[0]] for g in gates]
    b0 = [n - 1 - g[1][0] for g in gates]
    b1 = [n - 1 - g[1][1] if len(g[1]) > 1 else 0 for g in gates]
    want = [cref.gate_energy(n, hf, opc, b0, b1, [g[2] for g in gates], [g[3] for g in gates], [g[4] for g in gates], th, hx, hz,
                             hc.real.copy(), ham.constant_coeff)[0] for th in thetas]
    l1 = float(np.abs(hc).sum())
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(ham)
        sv.set_gate_program(gates, K, hf)
        assert sv.program_info()["literal_gates"] == 0
        got = [sv.energy(th) for th in thetas]
        info = sv.program_info()
        eg = [sv.energy_gradient(th) for th in thetas[1:]]
        sv.set_option("sector_reg_adjoint", 0)               # the backward sweeps on the pair words of the same tables
        eg_words = [sv.energy_gradient(th) for th in thetas[1:]]
        sv.set_option("sector_reg_adjoint", 1)
        # the support is the full coset of the two spin parities: the sweeps run from bit arithmetic (k_sector_sweep_reg) —
        # against the pair-word sweeps of the same tables, other workgroup sizes and other tile sizes
        variants = {}
        for name, opts in (("pair_words", {"sector_regular": 0}), ("single_ops", {"sector_regular": 1, "sector_reg_pairs": 0}),
                           ("canonical_orders", {"sector_regular": 3, "sector_reg_pairs": 1}),
                           ("threads_512", {"sector_regular": 1, "sector_reg_threads": 512}),
                           ("threads_128", {"sector_reg_threads": 128}), ("bits_10", {"sector_reg_threads": 256, "sector_bits": 10}),
                           ("bits_8", {"sector_bits": 8})):
            for k, v in opts.items():
                sv.set_option(k, v)
            variants[name] = ([sv.energy(th) for th in thetas], sv.program_info())
        sv.set_option("sector_bits", 0)
        sv.set_option("sector", 0)
        eg_dense = [sv.energy_gradient(th) for th in thetas[1:]]
    assert 0 < info["sector_support"] <= (1 << n) // 4 and info["sector_h_elements"] > 0, info
    assert info["sector_support"] == (1 << n) // 4 and info["sector_free_bits"] == 2 and info["sector_regular_slot_bits"] > 0, info
    assert variants["pair_words"][1]["sector_regular_slot_bits"] == 0 and variants["bits_8"][1]["sector_regular_slot_bits"] == 6
    for name, (es, _) in variants.items():
        assert np.abs(np.array(es) - np.array(got)).max() < 1e-13 * max(1.0, l1), name
    for e, ew in zip(got, want):
        assert abs(e - ew) < 1e-10 * max(1.0, l1)
    for (e, g), (ew_, gw_) in zip(eg, eg_words):
        assert abs(e - ew_) < 1e-13 * max(1.0, l1) and np.abs(g - gw_).max() < 1e-12 * max(1.0, l1)
    for (e, g), (ed, gd), ew in zip(eg, eg_dense, want[1:]):
        assert abs(e - ew) < 1e-10 * max(1.0, l1) and abs(ed - ew) < 1e-10 * max(1.0, l1)
        assert np.abs(g - gd).max() < 1e-11 * max(1.0, l1)


@pytest.mark.parametrize("m,o", [(7, 2), (7, 3), (8, 4)])
def test_sector_ground_state_is_the_fci_energy_of_the_sector(SV, m, o):
    """ovqe_sector_ground_state (Lanczos on the materialised Hamiltonian of the support) against the dense diagonalisation of
    the Hamiltonian's block on the (o alpha, o beta) determinants (bit-mask oracle), and against ovqe_ground_state's
    whole-register minimum (never above it... the sector may or may not hold the global minimum: >=)"""
    import itertools
    from openvqe_amd import fermion
    from oracle import masks
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=600 + m)
    # the sector: o electrons on the even (alpha) and o on the odd (beta) qubits, qubit 0 = most significant bit
    def strings(qubits):
        return [sum(1 << (n - 1 - q) for q in c) for c in itertools.combinations(qubits, o)]
    dets = np.array(sorted(a | b for a in strings(range(0, n, 2)) for b in strings(range(1, n, 2))), dtype=np.uint64)
    assert hf in set(dets.tolist())
    hx, hz, hc = ham.packed()
    block = np.zeros((len(dets), len(dets)))
    index = {int(d): k for k, d in enumerate(dets)}
    for x, z, c in zip(hx.tolist(), hz.tolist(), hc.real.tolist()):     # <i|P|j> = i^ny (-1)^{|j & z|} delta(i, j ^ x)
        ph = (1j) ** (bin(x & z).count("1") % 4)
        for k, j in enumerate(dets.tolist()):
            i = j ^ x
            if i in index:
                block[index[i], k] += (c * ph * (-1.0 if bin(j & z).count("1") & 1 else 1.0)).real
    assert np.abs(block - block.T).max() < 1e-12
    e_fci = float(np.linalg.eigvalsh(block)[0]) + ham.constant_coeff
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        e, res, its = sv.sector_ground_state(tol=1e-12)        # builds the tables itself
        info = sv.program_info()
        e_all, _, _ = sv.ground_state(tol=1e-10)
        rng = np.random.default_rng(m)
        e_var = sv.energy(rng.uniform(-0.3, 0.3, len(gens)))
    assert info["sector_support"] == len(dets)
    assert abs(e - e_fci) < 1e-9 and res < 1e-6, (e, e_fci, res, its)
    assert e_all <= e + 1e-9 and e <= e_var + 1e-12            # global minimum <= sector minimum <= any ansatz energy
    with SV(n) as sv:                                           # WITHOUT a program: the sector of the state in the buffer
        sv.set_hamiltonian(ham)
        sv.init_basis(hf)
        e_free, res_free, _ = sv.sector_ground_state(tol=1e-12)
        vec = sv.get_state()
        sv.init_basis(int(dets[len(dets) // 2]))                # another determinant of the same sector: same tables, same number
        e_again, _, _ = sv.sector_ground_state(tol=1e-12)
    assert abs(e_free - e_fci) < 1e-9 and res_free < 1e-6 and abs(e_again - e_fci) < 1e-9
    occupied = np.flatnonzero(vec)
    assert abs(np.linalg.norm(vec) - 1.0) < 1e-12 and set(occupied.tolist()) <= set(dets.tolist())
    assert abs(np.real(np.vdot(vec[dets.astype(np.int64)], block @ vec[dets.astype(np.int64)])) + ham.constant_coeff - e_fci) < 1e-9


def test_problem_fci_energy_on_h2o(SV):
    """chem.Problem.fci_energy (Lanczos on the sector tables of the problem's UCCSD program) against the determinant-space
    FCI of the SCF front-end, H2O / STO-3G (14 qubits, 441 determinants in the (5 alpha, 5 beta) sector)"""
    from openvqe_amd import chem
    mol = chem.molecule("H2O")
    mol.rhf()
    e_fci = mol.ci_ground_state()[0]
    assert abs(mol.problem(active=False).fci_energy() - e_fci) < 1e-9


def test_spin_adapted_generators_on_sector_tables(SV):
    """an ansatz of spin-adapted singlet generators (the ADAPT pool of ref:openvqe/common_files/generator_excitations.py:274-359,
    Trotterised: strings sharing one parameter) on the sector tables, whichever probe they come from (the second-probe case is
    pinned at 24 qubits in tests/test_gpu_fullsize.py): energies and gradients equal the dense kernels' """
    from openvqe_amd import fermion, pools
    m, o = 9, 3
    ham, _, hf = fermion.synthetic_molecule(m, o, seed=11)
    _, _, singlets = pools.singlet_sd(2 * o, m)
    rng = np.random.default_rng(9)
    picks = rng.permutation(len(singlets))[:30]
    gens = [1j * singlets[k] for k in picks]
    thetas = [rng.uniform(-0.25, 0.25, len(gens)) for _ in range(5)]
    out = {}
    for sector in (1, 0):
        with SV(2 * m) as sv:
            sv.set_option("sector", sector)
            sv.set_option("sector_min_qubits", 8)
            sv.set_option("sparse", 0)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            es = [sv.energy(t) for t in thetas]
            eg, g = sv.energy_gradient(thetas[-1])
            out[sector] = (es, eg, g, sv.program_info())
    info = out[1][3]
    l1 = float(np.abs(ham.packed()[2]).sum())
    for a, b in zip(out[1][0], out[0][0]):
        assert abs(a - b) < 1e-11 * l1
    assert abs(out[1][1] - out[0][1]) < 1e-11 * l1
    assert np.abs(out[1][2] - out[0][2]).max() < 1e-9 * l1
    assert info["sector_support"] > 0, info            # tables in use at the end (first or second probe)
    assert out[0][3]["sector_support"] == 0


@pytest.mark.parametrize("m,o", [(8, 4), (9, 4), (10, 4)])
def test_batched_evaluations_on_the_sector_tables_match_c_oracle(SV, m, o):
    """ovqe_energy_batch at 16 - 20 qubits: B parameter vectors per pass of the sector tables (the sweeps with one workgroup per
    (tile, state), <H> with two states per tile; ref:openvqe/ucc_family/get_energy_ucc.py:158-175 — BFGS with jac=None asks
    for K + 1 evaluations per gradient) for B = 1, 3, 8, 141 — a first call on a fresh handle included, which builds the tables
    at once — against the C oracle and against one evaluation at a time"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import compile_ucc_program
    from oracle import cref
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=300 + m)
    K = len(gens)
    rng = np.random.default_rng(m * o)
    thetas = rng.uniform(-0.3, 0.3, (141, K))
    thetas[5] = 0.0
    rx, rz, rc, pidx, _ = compile_ucc_program(n, gens)
    hx, hz, hc = ham.packed()
    # the oracle sees every vector at 16 qubits, the first 32 at 18 and the first 16 at 20 (141 evaluations there = three minutes of the suite's budget),
    # the rest of the big batch is held against one evaluation at a time on the same handle (the serial path has its own oracle tests)
    n_oracle = 141 if m < 9 else (32 if m < 10 else 16)      # (16: one evaluation per core of the GPU box's host, one round)
    want = cref.ucc_energy_batch(n, hf, rx, rz, rc, pidx, thetas[:n_oracle], hx, hz, hc.real.copy(), ham.constant_coeff)
    l1 = float(np.abs(hc).sum())
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        first = sv.energy_batch(thetas[:8])                  # fresh handle: the batch builds the tables itself
        info = sv.program_info()
        got = {B: sv.energy_batch(thetas[:B]) for B in (1, 3, 8, 141)}
        assert "sweep_wide" in sv.sector_forms()             # batches: one workgroup per (tile, state) on the 64-bit words (k_sector_sweep2)
        serial = np.array([sv.energy(t) for t in thetas[:9]])
        serial_rest = np.array([sv.energy(t) for t in thetas[n_oracle:]])
        # parameters and energies resident on the device (ovqe_energy_batch_device beyond the fused kernels' 16 qubits): the same
        # batched passes with nothing crossing PCIe; an odd batch (ragged last pair of states) and, with batches switched off, the
        # route through the host
        import torch
        th_dev = torch.from_numpy(thetas).cuda()
        en_dev = torch.zeros(141, dtype=torch.float64, device="cuda")
        sv.energy_batch_device(141, th_dev.data_ptr(), en_dev.data_ptr())
        torch.cuda.synchronize()
        resident = en_dev.cpu().numpy()
        en7 = torch.zeros(7, dtype=torch.float64, device="cuda")
        sv.energy_batch_device(7, th_dev[10:17].contiguous().data_ptr(), en7.data_ptr())
        resident7 = en7.cpu().numpy()
        sv.set_option("sector_batch", 0)
        unbatched = sv.energy_batch(thetas[:9])
        en5 = torch.zeros(5, dtype=torch.float64, device="cuda")
        sv.energy_batch_device(5, th_dev.data_ptr(), en5.data_ptr())
        resident5 = en5.cpu().numpy()
    assert np.abs(resident - got[141]).max() < 1e-13 * max(1.0, l1)
    assert np.abs(resident7[:6] - want[10:16]).max() < 1e-10 * max(1.0, l1) and abs(resident7[6] - got[141][16]) < 1e-13 * max(1.0, l1) and np.abs(resident5 - serial[:5]).max() < 1e-13 * max(1.0, l1)
    assert info["sector_support"] == comb(m, o) ** 2
    assert np.abs(first - want[:8]).max() < 1e-10 * max(1.0, l1)
    for B, e in got.items():
        assert e.shape == (B,) and np.abs(e[:n_oracle] - want[:B]).max() < 1e-10 * max(1.0, l1), B
    assert np.abs(got[141][n_oracle:] - serial_rest).max(initial=0.0) < 1e-12 * max(1.0, l1)
    assert np.abs(got[141][:9] - serial).max() < 1e-12 * max(1.0, l1)
    assert np.abs(unbatched - serial).max() < 1e-13 * max(1.0, l1)


@pytest.mark.parametrize("m,o,steps", [(8, 3, 12), (9, 4, 20)])
def test_adapt_screen_sigma_from_the_materialised_sector_hamiltonian(SV, m, o, steps):
    """ovqe_pool_gradients with sigma = H psi from the row-format tables of psi's symmetry sector (option "screen_sector": built once
    per Hamiltonian on the closure of the support under its x-groups, no circuit) against the tile cover / register pass of the
    same handle and, at 16 qubits, against the bit-mask oracle; the state is a chain of exact exponentials of spin-adapted
    generators from the Hartree-Fock determinant (ref:openvqe/adapt/fermionic_adapt_vqe.py:77-122) and stays inside the sector"""
    from openvqe_amd import fermion, pools
    from openvqe_amd.backend import GRAD_FERMIONIC, GRAD_QUBIT
    from openvqe_amd.operators import pack_terms
    from oracle import masks
    n = 2 * m
    ham, _, hf = fermion.synthetic_molecule(m, o, seed=700 + m)
    _, _, pool = pools.singlet_sd(2 * o, m)
    rng = np.random.default_rng(70 * m + o)
    picks = rng.choice(len(pool), size=steps, replace=False)
    thetas = rng.uniform(-0.5, 0.5, steps)
    l1 = float(np.abs(ham.packed()[2]).sum())
    with SV(n) as sv:
        sv.set_hamiltonian(ham)
        sv.init_basis(hf)
        for k, th in zip(picks, thetas):
            sv.apply_exp_pauli_sum(pool[k], th)
        psi = sv.get_state()
        sv.set_option("screen_sector_min", 1)
        g1 = np.array(sv.pool_gradients(pool, GRAD_FERMIONIC))
        used, listed = sv.last_screen_sector(), sv.last_screen_support()
        q1 = np.array(sv.pool_gradients(pool[::3], GRAD_QUBIT))
        sv.set_option("screen_sector", 0)
        g0 = np.array(sv.pool_gradients(pool, GRAD_FERMIONIC))
        assert sv.last_screen_sector() == 0
        q0 = np.array(sv.pool_gradients(pool[::3], GRAD_QUBIT))
    nz = np.flatnonzero(psi)
    assert listed == len(nz) and all(bin(int(i)).count("1") == 2 * o for i in nz)
    assert used == comb(m, o) ** 2, (used, comb(m, o) ** 2)
    scale = l1 * max(np.abs(pack_terms(n, op.terms)[2]).sum() for op in pool)
    assert np.abs(g1).max() > 1e-3 and np.abs(g1 - g0).max() < 1e-12 * scale and np.abs(q1 - q0).max() < 1e-12 * scale
    if n <= 16:
        hx, hz, hc = ham.packed()
        sigma = masks.apply_pauli_sum(psi, hx, hz, hc) + ham.constant_coeff * psi
        want = np.array([2.0 * np.vdot(sigma, masks.apply_pauli_sum(psi, *pack_terms(n, op.terms))).real for op in pool])
        assert np.abs(g1 - want).max() < 1e-11 * scale


def test_sector_ground_state_on_a_hopping_chain_needs_more_than_64_rounds(SV):
    """a SHORT-RANGE Hamiltonian: spinless nearest-neighbour hopping chain (Jordan-Wigner: (X_i X_{i+1} + Y_i Y_{i+1}) / 2 plus
    on-site Z terms), 20 sites, 10 particles all on one end.  The determinant graph of the sector (C(20,10) = 184 756) has
    diameter 10 * 10 = 100 hops from that determinant, so the search for the block of H connected to it takes ~100 matvec
    rounds — a search cut at 64 rounds diagonalised a truncated block and reported an energy above the minimum.  Free fermions:
    the exact ground energy is the sum of the 10 lowest single-particle levels."""
    from math import comb
    from openvqe_amd.operators import Hamiltonian, Term
    n, p = 20, 10
    rng = np.random.default_rng(2024)
    t = rng.uniform(0.6, 1.4, n - 1)
    eps = rng.uniform(-0.5, 0.5, n)
    terms = []
    for i in range(n - 1):
        terms += [Term(0.5 * t[i], "XX", [i, i + 1]), Term(0.5 * t[i], "YY", [i, i + 1])]
    terms += [Term(float(eps[i]), "Z", [i]) for i in range(n)]
    ham = Hamiltonian(n, terms, 0.125)
    # Z_i = 1 - 2 n_i; (XX + YY)/2 = hopping with amplitude +-t_i (the sign is a gauge choice on a chain)
    h1 = np.diag(-2.0 * eps) + np.diag(t, 1) + np.diag(t, -1)
    e_exact = 0.125 + eps.sum() + np.sort(np.linalg.eigvalsh(h1))[:p].sum()
    with SV(n) as sv:
        sv.set_hamiltonian(ham)
        sv.init_basis(((1 << p) - 1) << (n - p))
        e, res, its = sv.sector_ground_state(tol=1e-11, max_iter=3000)
        rounds = sv.last_fci_rounds()
        vec = sv.get_state()
    assert rounds > 64, rounds
    assert abs(e - e_exact) < 1e-8, (e, e_exact, res, its)
    occ = np.flatnonzero(np.abs(vec) > 0)
    assert len(occ) <= comb(n, p) and all(bin(int(i)).count("1") == p for i in occ[:: max(1, len(occ) // 500)])
    assert abs(np.linalg.norm(vec) - 1.0) < 1e-10


def test_differential_fuzz_of_the_sector_path(gpu_lib):
    """tools/fuzz_sector.py, 30 cases: random UCC-type programs and QUCCSD template lists at 14-20 qubits under random tile geometry,
    workgroup sizes, coding and sweep forms (pair words / bit arithmetic / blocks of two ops / runs without barriers): energies and all
    gradient components of the sector path against the dense-state kernels of the same handle (1e-11 / 1e-10 |H|_1).  The superseded
    kernel forms that no default selects are reached only through these option draws."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_sector.py"), "30", "2025"], capture_output=True, text=True,
                       timeout=900, cwd=root)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:]
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, tail
    assert "bit-arithmetic sweeps" in tail
    regular = int(tail.split("(")[1].split(" ")[0])
    assert regular >= 3, tail      # the regular-support kernels were among the draws


def test_differential_fuzz_with_the_per_wave_streams_forced(gpu_lib):
    """the same tool on the testing build (OVQE_LIB=testing), 22 cases: the draws add the third sweep form's streams forced on small
    tiles — 1 to 16 waves per tile, lanes of a row arranged for the LDS banks or in list order, words without partner, ops of several
    patterns (single-string and multi-term generators) — and the first / second forms, forwards and backwards"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_sector.py"), "22", "77"], capture_output=True, text=True,
                       timeout=900, cwd=root, env=dict(os.environ, OVQE_LIB="testing"))
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:]
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, tail


def test_gate_list_takes_the_coset_without_a_probe_and_is_checked(SV):
    """A gate list in frame form gets its sector tables on the coset of its Z2 symmetries WITHOUT a probe run (the reference's QUCCSD
    templates fill that coset: ref:openvqe/common_files/circuit.py:13-106); the build then checks with one evaluation at generic
    angles that the coset is populated.  (a) QUCCSD templates at 18 qubits: support = the spin-parity quarter, sweeps from bit arithmetic;
    (b) a NUMBER-CONSERVING gate list — UCCSD generators synthesised as CNOT staircases (what myQLM's build_ucc_ansatz emits) — populates
    (9 choose 4)^2 of the 2^16 coset members: the check rejects the coset and the tables come from a probe.  Energies of both against
    the dense-state kernels of the same handle."""
    import math

    from openvqe_amd import fermion
    from openvqe_amd.common_files.circuit import quccsd_gate_list
    m, o = 9, 4
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=909)
    l1 = float(np.abs(ham.packed()[2]).sum())
    rng = np.random.default_rng(99)
    # (b) literal gates of every 6th generator: basis changes, CNOT staircase, RZ(2 c theta), uncompute
    picked = gens[::6]
    gates = []
    for k, g in enumerate(picked):
        for t in g.terms:
            act = [(q, p) for q, p in zip(t.qbits, t.op) if p != "I"]
            pre = [("H", [q], 0.0, 0.0, -1) if p == "X" else ("RX", [q], 0.0, math.pi / 2, -1) for q, p in act if p in "XY"]
            post = [("H", [q], 0.0, 0.0, -1) if p == "X" else ("RX", [q], 0.0, -math.pi / 2, -1) for q, p in act if p in "XY"]
            ladder = [("CNOT", [a[0], b[0]], 0.0, 0.0, -1) for a, b in zip(act[:-1], act[1:])]
            gates += pre + ladder + [("RZ", [act[-1][0]], 2.0 * float(np.real(t.coeff)), 0.0, k)] + ladder[::-1] + post
    theta_b = rng.uniform(-0.4, 0.4, len(picked))
    qgates, Kq, hfq = quccsd_gate_list(m, o, 1)
    theta_a = rng.uniform(-0.3, 0.3, Kq)
    out = {}
    for label, glist, K, hf0, theta in (("quccsd", qgates, Kq, hfq, theta_a), ("staircase_uccsd", gates, len(picked), hf, theta_b)):
        with SV(n) as sv:
            sv.set_option("force_path", 2)
            sv.set_hamiltonian(ham)
            sv.set_gate_program(glist, K, hf0)
            assert sv.program_info()["literal_gates"] == 0
            e = [sv.energy(theta) for _ in range(3)]
            info = sv.program_info()
            sv.set_option("sector", 0)
            e_dense = sv.energy(theta)
        out[label] = (e, e_dense, info)
    for label, (e, e_dense, info) in out.items():
        assert np.abs(np.array(e) - e_dense).max() < 1e-11 * max(1.0, l1), (label, e, e_dense)
    assert out["quccsd"][2]["sector_support"] == 1 << (n - 2) and out["quccsd"][2]["sector_regular_slot_bits"] > 0
    # (every 6th generator does not reach all of the (4 alpha, 4 beta) sector: a probed support inside it, far below the 2^16 coset)
    assert 1000 < out["staircase_uccsd"][2]["sector_support"] <= comb(m, o) ** 2 and out["staircase_uccsd"][2]["sector_regular_slot_bits"] == 0
    # the same rotation sequence as a Pauli-rotation program gives the same energy (the frame compiler saw through the staircases)
    with SV(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(picked, hf)
        assert abs(sv.energy(theta_b) - out["staircase_uccsd"][1]) < 1e-11 * max(1.0, l1)


@pytest.fixture
def testing_lib(gpu_lib, monkeypatch):
    """the OVQE_TESTING build of the same source: it also takes the options that pick a kernel form"""
    from openvqe_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", _lib.TESTING_LIB_PATH)
    monkeypatch.setattr(_lib, "_lib", None)
    return _lib.lib()


@pytest.mark.parametrize("m,o,waves,arrange", [(8, 3, 0, 1), (8, 3, 1, 0), (8, 4, 2, 1), (9, 4, 0, 1), (9, 4, 4, 0), (9, 4, 16, 1), (10, 5, 8, 1)])
def test_per_wave_streams_equal_the_second_sweep_form_bit_for_bit(testing_lib, m, o, waves, arrange):
    """third form of the circuit sweeps (k_sector_sweep3 / k_sector_adjoint3: pair words in per-wave streams, barriers at run boundaries
    only) against the second form ON THE SAME TABLES: a pair is rotated by the same arithmetic in both, every slot sees its ops in
    program order, so the energies are equal bit for bit, whatever the number of waves that share a tile's rows and whether the lanes
    of a row were arranged for the LDS banks; the gradient's partial sums are added in another order (1e-12); both against the oracle"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import Statevector
    n = 2 * m
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=300 + m)
    rng = np.random.default_rng(31 * m + o)
    thetas = [rng.uniform(-0.3, 0.3, len(gens)) for _ in range(3)]
    want = _oracle_energies(n, gens, hf, ham, thetas)
    l1 = float(np.abs(ham.packed()[2]).sum())
    with Statevector(n) as sv:
        sv.set_option("force_path", 2)
        sv.set_option("sector_min_qubits", 8)
        sv.set_option("sector_stream_waves", waves)
        sv.set_option("sector_stream_arrange", arrange)
        sv.set_hamiltonian(ham)
        sv.set_ucc_program(gens, hf)
        for th in thetas:
            sv.energy(th)                       # (the tables are built at the second evaluation)
        got, grad = {}, {}
        for form in (3, 2):
            sv.set_option("sector_sweep", form)
            sv.set_option("sector_adjoint", form)
            got[form] = [sv.energy(th) for th in thetas]
            grad[form] = [sv.energy_gradient(th) for th in thetas]
        sv.set_option("sector_sweep", 4)        # (the streams for the states of a batch too: the product gives batches the second form)
        batch = sv.energy_batch(np.stack(thetas))
    assert got[3] == got[2]
    for (e3, g3), (e2, g2), ew in zip(grad[3], grad[2], want):
        assert abs(e3 - ew) < 1e-10 * max(1.0, l1) and abs(e3 - e2) < 1e-13 * max(1.0, l1)    # (the gradient's energy is a dot product with lambda)
        assert np.abs(g3 - g2).max() < 1e-12 * max(1.0, l1)
    assert np.abs(batch - np.array(want)).max() < 1e-10 * max(1.0, l1)


def test_packed_hamiltonian_elements_equal_the_32_bit_words_bit_for_bit(testing_lib):
    """<H> tables of a sweep whose dictionary has at most 1023 magnitudes keep their coded words as 24-bit elements (k_sec_pack24: a
    quarter fewer bytes for the kernels that stream them): same elements in the same order, so energies, batches and gradients equal
    those of the 32-bit words (energies and batches bit for bit; H2O / STO-3G UCCSD, 14 qubits; a 16-qubit synthetic molecule), and the oracle"""
    from openvqe_amd import chem, fermion
    from openvqe_amd.backend import Statevector
    mol = chem.molecule("H2O")
    mol.rhf()
    prob = mol.problem(active=False)
    _, _, gens_w, theta_mp2, hf_w = prob.uccsd()
    cases = [(prob.jw_hamiltonian(), gens_w, hf_w, np.array(theta_mp2))]
    ham, gens, hf = fermion.synthetic_molecule(8, 3, seed=77)
    cases.append((ham, gens, hf, np.random.default_rng(8).uniform(-0.3, 0.3, len(gens))))
    for ham, gens, hf, th in cases:
        n = ham.nbqbits
        thetas = [th, 0.5 * th, -0.7 * th]
        want = _oracle_energies(n, gens, hf, ham, thetas)
        l1 = float(np.abs(ham.packed()[2]).sum())
        got = {}
        for pack in (1, 0):
            with Statevector(n) as sv:
                sv.set_option("force_path", 2)
                sv.set_option("sector_min_qubits", 8)
                sv.set_option("sector_h_pack", pack)
                sv.set_hamiltonian(ham)
                sv.set_ucc_program(gens, hf)
                es = [sv.energy(t) for t in thetas]
                es = [sv.energy(t) for t in thetas]           # (from the tables)
                grads = [sv.energy_gradient(t)[1] for t in thetas]
                got[pack] = (es, sv.energy_batch(np.stack(thetas)), grads, sv.program_info()["sector_bytes"])
        assert got[1][0] == got[0][0] and np.array_equal(got[1][1], got[0][1])
        assert all(np.abs(a - b).max() < 1e-12 * max(1.0, l1) for a, b in zip(got[1][2], got[0][2]))   # (lambda = H psi adds with atomics: rounding)
        assert got[1][3] < got[0][3]                                   # the packed tables are smaller
        assert max(abs(e - w) for e, w in zip(got[1][0], want)) < 1e-10 * max(1.0, l1)
