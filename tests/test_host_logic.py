"""CPU tests of the host side: operator containers, JW front-end, program compilation, the qat stand-ins
and the L1 mirrors (EnergyUCC / fermionic_adapt_vqe / qubit_adapt_vqe).  The numerical engine is replaced
by the oracle-backed stand-in of tests/oracle_backend.py (there is no GPU here); where /root/reference is
present the reference's OWN L1 modules are imported unchanged on top of the qat stand-ins and must produce
the same trajectories as the mirrors."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

from openvqe_amd import fermion
from openvqe_amd.operators import Hamiltonian, Term, pack_string
from oracle import dense
from tests.oracle_backend import OracleStatevector

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference"


@pytest.fixture()
def oracle_engine(monkeypatch):
    import openvqe_amd.adapt.fermionic_adapt_vqe as fa
    import openvqe_amd.adapt.qubit_adapt_vqe as qa
    import openvqe_amd.backend as be
    import openvqe_amd.evaluator as ev
    import openvqe_amd.qat_compat as qc
    for mod in (be, ev, fa, qa):
        monkeypatch.setattr(mod, "Statevector", OracleStatevector)
    ev._BACKENDS.clear()
    ev._Evaluator._owner.clear()
    fa._screens.clear(); fa._evaluators.clear()
    qa._screens.clear(); qa._evaluators.clear()
    monkeypatch.setattr(qc, "_default_qpu", None)
    yield
    ev._BACKENDS.clear()
    ev._Evaluator._owner.clear()
    fa._screens.clear(); fa._evaluators.clear()
    qa._screens.clear(); qa._evaluators.clear()
    qc._default_qpu = None


@pytest.fixture(scope="module")
def h2():
    k1 = json.load(open(os.path.join(GOLD, "k1_h2_sto3g.json")))
    H = Hamiltonian(4, [Term(c, o, q) for c, o, q in k1["terms"]], k1["constant_coeff"])
    e0 = np.linalg.eigvalsh(H.get_matrix())[0]
    return H, k1["hf_init"], e0


def reference_module(name):
    if not os.path.isdir(REF):
        pytest.skip("reference tree not present (GPU box)")
    sys.dont_write_bytecode = True  # /root/reference is read-only by contract
    from openvqe_amd import qat_compat
    qat_compat.install(force=True)
    if REF not in sys.path:
        sys.path.append(REF)
    return importlib.import_module(name)


# ------------------------------------------------------------------------------------ containers
def test_hamiltonian_algebra_and_matrix():
    a = Hamiltonian(3, [Term(0.5, "XY", [0, 2]), Term(-1.5, "Z", [1])], 0.25)
    b = Hamiltonian(3, [Term(2.0, "YZ", [0, 1])])
    ma, mb = a.get_matrix(), b.get_matrix()
    assert np.allclose(ma, dense.operator_matrix(a, sparse=False))
    assert np.allclose((a * b).get_matrix(), ma @ mb)
    assert np.allclose((a + b).get_matrix(), ma + mb)
    assert np.allclose((a * 1j).get_matrix(), 1j * ma)
    assert np.allclose((1j * a).get_matrix(), 1j * ma)
    assert np.allclose((a / 2).get_matrix(), ma / 2)
    assert np.allclose(a.get_matrix(sparse=True).toarray(), ma)
    merged = Hamiltonian(2, [Term(1.0, "XZ", [0, 1]), Term(0.5, "ZX", [1, 0]), Term(2.0, "II", [0, 1])])
    assert len(merged.terms) == 1 and merged.terms[0].coeff == 1.5 and merged.constant_coeff == 2.0


def test_packing_convention_qubit0_is_msb():
    assert pack_string(4, "X", [0]) == (0b1000, 0)
    assert pack_string(4, "Y", [3]) == (1, 1)
    assert pack_string(4, "ZX", [1, 2]) == (0b0010, 0b0100)
    with pytest.raises(ValueError):
        pack_string(3, "X", [3])


def test_jw_pool_sizes_and_hf_integer():
    # UCCSD operator counts: singles 2ov, doubles 2C(o,2)C(v,2)+(ov)^2 — 26 for H4 (ref:tests/test_main_quccsd.py:15),
    # 8 for the H4 active space (ref:tests/test_main_quccsd_active_space.py:15)
    for (m, o), want in (((4, 2), 26), ((2, 1), 3), ((7, 5), 140), ((6, 2), 92)):
        s, d = fermion.uccsd_excitations(m, o)
        assert len(s) + len(d) == want
    s, d = fermion.uccsd_excitations(3, 1)  # 2 electrons in 3 orbitals (H4 active: 8 operators)
    assert len(s) + len(d) == 8
    assert fermion.hf_integer(8, 2) == 192  # H2/6-31G reference ket index, ref:notebooks/demo_fermionic_adapt.ipynb
    assert fermion.hf_integer(4, 2) == 12
    gens = fermion.uccsd_generators(7, 5)
    assert sum(len(g.terms) for g in gens) == 1000
    for g in fermion.uccsd_generators(3, 1):
        assert np.allclose(g.get_matrix(), g.get_matrix().conj().T)


def test_jw_hamiltonian_is_hermitian_and_number_conserving():
    h, g = fermion.synthetic_integrals(3, seed=5)
    hpq, hpqrs = fermion.spin_orbital_integrals(h, g)
    ham = fermion.jw_molecular_hamiltonian(hpq, hpqrs)
    mat = ham.get_matrix()
    assert np.allclose(mat, mat.conj().T)
    num = sum(fermion.psum_to_hamiltonian(6, fermion.jw_product([(p, True), (p, False)])).get_matrix() for p in range(6))
    assert np.allclose(mat @ num, num @ mat)


def test_compile_ucc_program_zip_truncation_and_real_check():
    from openvqe_amd.backend import compile_ucc_program
    gens = fermion.uccsd_generators(2, 1)
    xs, zs, cs, ps, K = compile_ucc_program(4, gens, n_params=2)
    assert K == 2 and len(xs) == 4 and ps.tolist() == [0, 0, 1, 1]
    with pytest.raises(ValueError):
        compile_ucc_program(4, fermion.uccsd_pool_antihermitian(2, 1))


# ------------------------------------------------------------------------------------ qat stand-ins
def test_circuit_ops_gate_counts():
    from openvqe_amd.common_files.circuit import count
    from openvqe_amd.qat_compat import Program, build_ucc_ansatz
    g = Hamiltonian(5, [Term(0.5, "XZZY", [0, 1, 2, 3]), Term(-0.5, "YX", [1, 4])], do_clean_up=False)
    prog = Program()
    reg = prog.qalloc(5)
    prog.apply(build_ucc_ansatz([g], 0b11000, n_steps=1)([0.3]), reg)
    ops = prog.to_circ().ops
    assert count("CNOT", ops) == 2 * 3 + 2 * 1      # 2(w-1) per string
    assert count("H", ops) == 2 * 1 + 2 * 1          # 2 per X
    assert count("RX", ops) == 2 * 1 + 2 * 1         # 2 per Y (RX(+-pi/2))
    assert count("X", ops) == 2 and count("rz", ops) == 2


def test_qpu_submit_obs_and_sample(oracle_engine, h2):
    from openvqe_amd.qat_compat import CNOT, RY, H as Hgate, Program, X, build_ucc_ansatz, get_default_qpu
    ham, hf, _ = h2
    gens = fermion.uccsd_generators(2, 1)
    prog = Program()
    reg = prog.qalloc(4)
    for k, (g, th) in enumerate(zip(gens, [0.1, -0.2, 0.3])):
        prog.apply(build_ucc_ansatz([g], hf if k == 0 else 0, n_steps=1)([th]), reg)
    circ = prog.to_circ()
    val = get_default_qpu().submit(circ.to_job(job_type="OBS", observable=ham)).value
    assert abs(val - dense.ucc_energy(ham, gens, hf, [0.1, -0.2, 0.3])) < 1e-12
    res = get_default_qpu().submit(circ.to_job())
    psi = np.zeros(16, complex)
    for s in res:
        psi[s.state.int] = s.amplitude
    assert np.abs(psi - dense.ucc_state(4, hf, gens, [0.1, -0.2, 0.3])).max() < 1e-12
    # literal gates after the X preparation
    p2 = Program()
    q = p2.qalloc(3)
    p2.apply(X, q[0]); p2.apply(Hgate, q[1]); p2.apply(CNOT, q[1], q[2]); p2.apply(RY(0.4), q[0])
    got = get_default_qpu().submit(p2.to_circ().to_job())
    ref = dense.gate_circuit_state(3, 0, [("X", [0], None), ("H", [1], None), ("CNOT", [1, 2], None), ("RY", [0], 0.4)])
    psi = np.zeros(8, complex)
    for s in got:
        psi[s.state.int] = s.amplitude
    assert np.abs(psi - ref).max() < 1e-12


def test_qpu_caches_literal_gate_circuits_by_structure(oracle_engine, h2, monkeypatch):
    """Route A: a circuit of literal gates (what ref:openvqe/common_files/circuit.py's templates submit per evaluation) is compiled
    once per STRUCTURE — gate names, qubits, quarter-turn angles — and later submissions pass their angles only"""
    import math
    from tests.oracle_backend import OracleStatevector
    from openvqe_amd.qat_compat import CNOT, RX, RY, RZ, H as Hgate, Program, X, get_default_qpu
    ham, _, _ = h2
    compiled = []
    real = OracleStatevector.set_gate_program
    monkeypatch.setattr(OracleStatevector, "set_gate_program", lambda self, gates, k, hf: (compiled.append(k), real(self, gates, k, hf))[1])

    def template(a, b, c):
        prog = Program()
        q = prog.qalloc(4)
        prog.apply(X, q[0]); prog.apply(X, q[1])
        prog.apply(Hgate, q[2]); prog.apply(RX(math.pi / 2), q[0]); prog.apply(CNOT, q[0], q[2])
        prog.apply(RZ(a), q[2]); prog.apply(CNOT, q[0], q[2]); prog.apply(RX(-math.pi / 2), q[0]); prog.apply(Hgate, q[2])
        prog.apply(RY(b), q[1]); prog.apply(CNOT, q[1], q[3]); prog.apply(RY(c), q[3]); prog.apply(CNOT, q[1], q[3])
        gates = [("X", [0], None), ("X", [1], None), ("H", [2], None), ("RX", [0], math.pi / 2), ("CNOT", [0, 2], None), ("RZ", [2], a),
                 ("CNOT", [0, 2], None), ("RX", [0], -math.pi / 2), ("H", [2], None), ("RY", [1], b), ("CNOT", [1, 3], None),
                 ("RY", [3], c), ("CNOT", [1, 3], None)]
        return prog.to_circ(), gates

    qpu = get_default_qpu()
    for k, angles in enumerate([(0.3, -0.2, 0.7), (0.31, 0.5, -0.1), (-1.2, 0.05, 0.9)]):
        circ, gates = template(*angles)
        psi = dense.gate_circuit_state(4, 0, gates)
        want = float(np.real(np.vdot(psi, ham.get_matrix() @ psi)))
        assert abs(qpu.submit(circ.to_job(job_type="OBS", observable=ham)).value - want) < 1e-12
        got = np.zeros(16, complex)
        for smp in qpu.submit(circ.to_job()):
            got[smp.state.int] = smp.amplitude
        assert np.abs(got - psi).max() < 1e-12
    assert compiled == [3]                      # one compilation with three parameters for six submissions
    circ, gates = template(0.3, 0.0, 0.7)       # an angle that IS a quarter turn changes the structure: compiled again, still right
    psi = dense.gate_circuit_state(4, 0, gates)
    assert abs(qpu.submit(circ.to_job(job_type="OBS", observable=ham)).value - float(np.real(np.vdot(psi, ham.get_matrix() @ psi)))) < 1e-12
    assert compiled == [3, 2]


# ------------------------------------------------------------------------------------ L1 mirrors
def _pool_generator(n=4):
    return [Hamiltonian(n, [Term(1.0, s, [0, 1, 2, 3])], do_clean_up=False) for s in ("XXXY", "YXXX", "XYXX")]


def test_energy_ucc_get_energies(oracle_engine, h2, capsys):
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    ham, hf, e0 = h2
    gens = fermion.uccsd_generators(2, 1)
    it, res = EnergyUCC().get_energies(ham, gens, _pool_generator(), hf, [0.0] * 3, [0.0] * 3, e0)
    out = capsys.readouterr().out
    assert "tolerance=  0.0001" in out and "method=  BFGS" in out
    assert abs(it["minimum_energy_result1_guess"][0] - e0) < 1e-7
    assert abs(it["minimum_energy_result2_guess"][0] - e0) < 1e-7
    assert res["CNOT1"] == res["CNOT2"] == 2 * 2 * 4 + 8 * 6
    assert res["len_op1"] == 3 and res["len_op2"] == 3
    assert res["energies_1"][0] == pytest.approx(-1.0716472822963232, abs=1e-13)
    assert set(res) == {"CNOT1", "CNOT2", "len_op1", "len_op2", "energies1_substracted_from_FCI",
                        "energies2_substracted_from_FCI", "energies_1", "energies_2"}


def test_gradient_options_of_the_ucc_mirrors(oracle_engine, h2):
    """host logic of the opt-in Jacobians (batched forward differences / adjoint): both reach the minimum of the plain
    jac=None run with fewer objective calls; the QUCCSD mirror likewise (engine = oracle, whose energy_gradient is a
    central difference)"""
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC as EnergyQUCC
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    ham, hf, e0 = h2
    gens = fermion.uccsd_generators(2, 1)
    runs = {}
    for flag in (None, "batched_gradient", "adjoint_gradient"):
        ucc = EnergyUCC()
        if flag:
            setattr(ucc, flag, True)
        sink = []
        runs[flag] = (ucc._minimize(ham, gens, hf, [0.0] * 3, sink, "BFGS", 1e-4), len(sink))
    for flag in ("batched_gradient", "adjoint_gradient"):
        assert abs(runs[flag][0].fun - runs[None][0].fun) < 1e-8
        assert runs[flag][1] < runs[None][1]
    ops = _qucc_cluster_ops()
    out = {}
    for flag in (False, True):
        q = EnergyQUCC()
        q.adjoint_gradient = flag
        out[flag] = q.get_energies(ham, ops, hf, [0.01] * len(ops), [0.0] * len(ops), e0)
    assert abs(out[True][0]["minimum_energy_result1_guess"][0] - out[False][0]["minimum_energy_result1_guess"][0]) < 1e-7
    assert len(out[True][1]["energies_1"]) < len(out[False][1]["energies_1"])


def test_energy_ucc_matches_reference_module(oracle_engine, h2):
    ham, hf, e0 = h2
    ref = reference_module("openvqe.ucc_family.get_energy_ucc")
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    gens = fermion.uccsd_generators(2, 1)
    it_r, res_r = ref.EnergyUCC().get_energies(ham, gens, _pool_generator(), hf, [0.0] * 3, [0.01] * 3, e0)
    it_m, res_m = EnergyUCC().get_energies(ham, gens, _pool_generator(), hf, [0.0] * 3, [0.01] * 3, e0)
    assert len(res_r["energies_1"]) == len(res_m["energies_1"])
    assert np.abs(np.array(res_r["energies_1"]) - np.array(res_m["energies_1"])).max() < 1e-12
    assert np.abs(np.array(res_r["energies_2"]) - np.array(res_m["energies_2"])).max() < 1e-12
    assert res_r["CNOT1"] == res_m["CNOT1"] and res_r["CNOT2"] == res_m["CNOT2"]
    assert np.allclose(it_r["theta_optimized_result1"], it_m["theta_optimized_result1"], atol=1e-10)


def _qucc_cluster_ops():
    # only op.terms[0].qbits is consumed (get_energy_qucc.py:46-49)
    mk = lambda qs: Hamiltonian(4, [Term(1.0, "X" * len(qs), qs)], do_clean_up=False)  # noqa: E731
    return [mk([0, 2]), mk([1, 3]), mk([0, 1, 2, 3])]


def test_quccsd_energy_mirror_and_reference(oracle_engine, h2):
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
    ham, hf, e0 = h2
    ops = _qucc_cluster_ops()
    theta = [0.11, -0.07, 0.23]
    mine = EnergyUCC()
    e = mine.action_quccsd(theta, ham, ops, hf, [])
    # literal circuit through the dense oracle
    circ = mine.prepare_state_ansatz(ham, hf, ops, theta)
    gates = [(o.gate, o.qbits, o.angle) for o in circ.ops]
    psi = dense.gate_circuit_state(4, 0, gates)
    assert abs(e - dense.expectation(ham, psi)) < 1e-12
    it, res = mine.get_energies(ham, ops, hf, [0.0] * 3, [0.01] * 3, e0)
    assert it["minimum_energy_result1_guess"][0] < -1.0716 and res["len_op1"] == 3
    if os.path.isdir(REF):
        ref = reference_module("openvqe.ucc_family.get_energy_qucc")
        e_ref = ref.EnergyUCC().action_quccsd(theta, ham, ops, hf, [])
        assert abs(e_ref - e) < 1e-12
        it_r, res_r = ref.EnergyUCC().get_energies(ham, ops, hf, [0.0] * 3, [0.01] * 3, e0)
        assert res_r["CNOT1"] == res["CNOT1"]
        assert np.abs(np.array(res_r["energies_1"]) - np.array(res["energies_1"])).max() < 1e-11


def test_fermionic_adapt_mirror_and_reference(oracle_engine, h2):
    from openvqe_amd.adapt.fermionic_adapt_vqe import fermionic_adapt_vqe
    ham, hf, e0 = h2
    pool = fermion.uccsd_pool_antihermitian(2, 1)
    args = dict(n_max_grads=1, fci=e0, optimizer="COBYLA", tolerance=1e-6, type_conver="norm",
                threshold_needed=1e-2, max_external_iterations=10)
    it, res = fermionic_adapt_vqe(None, None, None, ham, pool, hf, **args)
    assert res["indices"] == [2] and abs(res["final_energy_last_iteration"] - e0) < 1e-6
    assert it["CNOTs"] == [48] and it["Hadamard"] == [32] and it["RX"] == [32]
    if os.path.isdir(REF):
        ref = reference_module("openvqe.adapt.fermionic_adapt_vqe")
        sparse_pool = [a.get_matrix(sparse=True) for a in pool]
        ket = np.zeros((16, 1), complex); ket[hf] = 1
        import scipy.sparse
        it_r, res_r = ref.fermionic_adapt_vqe(ham.get_matrix(sparse=True), sparse_pool, scipy.sparse.csr_matrix(ket),
                                              ham, pool, hf, **args)
        assert res_r["indices"] == res["indices"]
        assert np.abs(np.array(it_r["energies"]) - np.array(it["energies"])).max() < 1e-9
        assert np.abs(np.array(it_r["norms"]) - np.array(it["norms"])).max() < 1e-9
        assert it_r["CNOTs"] == it["CNOTs"] and it_r["Hadamard"] == it["Hadamard"]
        assert np.abs(np.array(it_r["fidelity"]) - np.array(it["fidelity"])).max() < 1e-9


def test_qubit_adapt_mirror_and_reference(oracle_engine, h2):
    from openvqe_amd.adapt.qubit_adapt_vqe import qubit_adapt_vqe
    ham, hf, e0 = h2
    strings = ["YXXX", "XYXX", "XXYX", "XXXY"]
    pool = [Hamiltonian(4, [Term(-1.0, s, [0, 1, 2, 3])], do_clean_up=False) for s in strings]
    pool += [Hamiltonian(4, [Term(-1.0, "YX", [0, 2])], do_clean_up=False),
             Hamiltonian(4, [Term(-1.0, "XY", [1, 3])], do_clean_up=False)]
    kw = dict(n_max_grads=1, adapt_conver="norm", adapt_thresh=1e-5, adapt_maxiter=6, tolerance_sim=1e-9,
              method_sim="BFGS")
    it, _, res, _ = qubit_adapt_vqe(ham, None, None, 4, pool, hf, e0, **kw)
    assert abs(res["final_energy"] - e0) < 1e-8 and res["indices"][0] == 0
    assert it["CNOTs"][0] == 6 and it["Hadamard"][0] == 6
    if os.path.isdir(REF):
        ref = reference_module("openvqe.adapt.qubit_adapt_vqe")
        import scipy.sparse
        ket = np.zeros((16, 1), complex); ket[hf] = 1
        it_r, _, res_r, _ = ref.qubit_adapt_vqe(ham, ham.get_matrix(sparse=True), scipy.sparse.csr_matrix(ket), 4, pool,
                                                hf, e0, **kw)
        assert res_r["indices"] == res["indices"]
        assert np.abs(np.array(it_r["energies"]) - np.array(it["energies"])).max() < 1e-9
        assert np.abs(np.array(it_r["norms"]) - np.array(it["norms"])).max() < 1e-8
        assert it_r["CNOTs"] == it["CNOTs"]


def test_result_of_a_state_vector_job_keeps_arrays_and_yields_samples():
    """qat stand-in Result: samples given as (indices, amplitudes) arrays are turned into Sample objects on iteration"""
    from openvqe_amd.qat_compat import Result, Sample
    idx = np.array([3, 5, 12], dtype=np.int64)
    amp = np.array([0.6, 0.8j, 0.0 + 0.0j])
    res = Result(indices=idx, amplitudes=amp, nbqbits=4)
    assert len(res) == 3
    got = [(s.state.int, s.amplitude, s.probability) for s in res]
    assert got == [(3, 0.6 + 0j, 0.36), (5, 0.8j, 0.6400000000000001), (12, 0j, 0.0)]
    assert str(res.raw_data[1].state) == str(Sample(5, 0.8j, 4).state)
    assert len(Result(value=1.5)) == 0 and Result(value=1.5).indices is None
    legacy = Result(samples=[Sample(1, 1.0, 2)])
    assert [s.state.int for s in legacy] == [1] and legacy.indices is None


def test_qpu_cache_keys_follow_operator_contents():
    """HipQPU skips recompilation / re-upload for the SAME operator objects only while their contents are unchanged: the
    reference rebuilds everything per submission, so in-place edits between submissions are legal there"""
    from openvqe_amd.operators import Hamiltonian, Term
    from openvqe_amd.qat_compat import HipQPU
    ham = Hamiltonian(4, [Term(0.1 * (k + 1), "XZ", [k % 3, 3]) for k in range(40)], 0.5, do_clean_up=False)
    f0 = HipQPU._fingerprint(ham)
    assert HipQPU._fingerprint(ham) == f0 == HipQPU._fingerprint(ham.copy())
    ham.terms[17].coeff *= 2
    f1 = HipQPU._fingerprint(ham)
    assert f1 != f0
    ham.terms.append(Term(0.0, "Z", [0]))
    f2 = HipQPU._fingerprint(ham)
    assert f2 != f1
    ham.constant_coeff = 0.25
    assert HipQPU._fingerprint(ham) != f2


def test_remaining_qubit_pool_kinds_have_the_expected_shape():
    """'two' / 'four' / 'minimal' / 'pure_with_symmetry' / 'eight' / 'without_Z_from_generator' of pools.qubit_pool (the dispatcher
    kinds of ref:openvqe/common_files/qubit_pool.py:1249-1266; operator-by-operator equality with the reference's module is
    checked in tests/test_reference_quccsd.py where the reference tree is present): sizes, Hermiticity of i x operator, and the
    defining property of the projected families — every string of an operator is the base string times Z's on its own qubits"""
    import numpy as np
    from openvqe_amd import pools
    n = 8
    n2, two = pools.qubit_pool("two", n)
    n4, four = pools.qubit_pool("four", n)
    assert n2 == 50 == len(two) and all(len(op.terms) == 2 for op in two)
    assert n4 == 54 and all(len(op.terms) in (2, 4) for op in four)
    for op in two + four:
        support = set(op.terms[-1].qbits)
        assert all(set(t.qbits) == support for t in op.terms)
        assert all(abs(complex(t.coeff).imag) < 1e-15 and abs(abs(complex(t.coeff).real) - 1.0) < 1e-15 for t in op.terms)
        assert all(sum(c in "Y" for c in t.op) % 2 == 1 for t in op.terms)       # odd number of Y: i x operator is real antisymmetric
    nm, minimal = pools.qubit_pool("minimal", n)
    assert nm == 2 * n - 2 and all(len(op.terms) == 1 and op.terms[0].op[0] == "Y" for op in minimal)
    assert pools.qubit_pool("pure_with_symmetry", 8, molecule_symbol="H4")[0] == 11
    _, _, source = pools.singlet_sd(2, 4, "JW")
    ne, eight = pools.qubit_pool("eight", n, source_pool=source)
    nw, plain = pools.qubit_pool("without_Z_from_generator", n, source_pool=source)
    assert ne <= nw == sum(1 for op in source if op.terms)
    assert all("Z" not in t.op for op in plain for t in op.terms)
    with pytest.raises(ValueError):
        pools.qubit_pool("eight", n)
    with pytest.raises(KeyError):
        pools.qubit_pool("no such pool", n)


def test_pack_terms_refuses_ragged_and_non_pauli_input():
    """the vectorised packer flattens all qubits side by side — a term with more qubits than characters would shift
    every later term's masks; characters outside Latin-1 must end in the same ValueError as any other non-Pauli character"""
    from types import SimpleNamespace as T

    from openvqe_amd.operators import pack_terms
    with pytest.raises(ValueError, match="term 0: 2 Pauli characters on 1 qubits"):
        pack_terms(4, [T(coeff=1.0, op="XY", qbits=[0]), T(coeff=1.0, op="Z", qbits=[1, 2])])
    for text in ("X中", "Xé", "XQ"):
        with pytest.raises(ValueError, match="unknown Pauli"):
            pack_terms(4, [T(coeff=1.0, op=text, qbits=[0, 1])])
    xs, zs, cs = pack_terms(4, [T(coeff=1.0, op="XY", qbits=[0, 3]), T(coeff=2.0, op="", qbits=[]), T(coeff=0.5, op="Z", qbits=[2])])
    assert xs.tolist() == [0b1001, 0, 0] and zs.tolist() == [0b0001, 0, 0b0010]
