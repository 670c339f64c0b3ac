"""The gate-level QUCCSD path and the front-end rows §8f-1/2 pinned on numbers the reference stores:

  K5   ref:notebooks/demo_quccsd.ipynb               H4/STO-3G, 26 cluster operators: E(theta_MP2) and E(0.01) with the 26
       forward-difference evaluations of scipy's BFGS behind each (2 x 27 stored energies), both stored optima with their
       26 parameters, CNOT count 292, printed NOONs, MP2 energy;
  K5a  ref:notebooks/demo_quccsd_active_space.ipynb  the NOON-selected active space (6 qubits, 8 operators): the same set
       of numbers, CNOT count 70, printed thresholds;
  K6b  ref:notebooks/demo_puccgsd.ipynb, second run: the qubit pool DERIVED from the cluster operators
       ('reduced_without_Z', ref:openvqe/algorithms/ucc.py:10-21) as generators — E(0.01) + 18 forward differences;
  K3a  ref:notebooks/demo_fermionic_adapt.ipynb, active run: reference energy in the natural-orbital basis, selected
       pool indices [23, 32, 38], gradient norms, energies, gate counts.

What is fitted and what is predicted: operator ORDER and tuple FORM of myQLM's cluster operators and the
term order of its JW transform are not in the reference tree; they were inferred from the stored first-order gradients /
CNOT counts and are then CHECKED by every other stored number (energies to <= 1e-8 absolute — the reference's PySCF runs
with default SCF / CISD thresholds —, energy differences to 1e-12).  The molecular-orbital phase gauge is PySCF's
documented rule (largest AO coefficient positive), not a fit."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from openvqe_amd import chem, pools
from tests.oracle_backend import OracleStatevector

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FD = float(np.sqrt(np.finfo(float).eps))   # scipy's forward-difference step of BFGS with jac=None


@pytest.fixture(scope="module")
def runs():
    return json.load(open(os.path.join(GOLD, "k5_k7_notebook_runs.json")))


@pytest.fixture(scope="module")
def h4():
    mol = chem.molecule("H4")
    mol.rhf()
    return mol


@contextlib.contextmanager
def engine(cls):
    """run the L1 mirrors on the oracle-backed engine (CPU) or, with cls = None, on the HIP library"""
    import openvqe_amd.adapt.fermionic_adapt_vqe as fa
    import openvqe_amd.backend as be
    import openvqe_amd.evaluator as ev
    import openvqe_amd.qat_compat as qc
    saved = [(m, m.Statevector) for m in (be, ev, fa)]
    caches = (ev._BACKENDS, ev._Evaluator._owner, fa._screens, fa._evaluators)
    for m, _ in saved:
        m.Statevector = cls if cls else m.Statevector
    for c in caches:
        c.clear()
    qc._default_qpu = None
    try:
        yield
    finally:
        for c in (ev._BACKENDS, fa._screens):
            for sv in c.values():
                sv.close()
        for c in caches:
            c.clear()
        if qc._default_qpu:
            for sv in qc._default_qpu._sv.values():
                sv.close()
        qc._default_qpu = None
        for m, s in saved:
            m.Statevector = s


def _check_quccsd_run(problem, run, abs_tol, cnot):
    """the stored numbers of one QUCCSD notebook against EnergyUCC.action_quccsd of the mirror"""
    from openvqe_amd.common_files.circuit import count
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
    ham = problem.jw_hamiltonian()
    size, cluster_ops, _, theta_mp2, hf = problem.uccsd()
    assert size == run["len_op1"] == len(theta_mp2)
    q = EnergyUCC()
    energy = lambda th: q.action_quccsd(th, ham, cluster_ops, hf, [])   # noqa: E731
    assert count("CNOT", q.prepare_state_ansatz(ham, hf, cluster_ops, theta_mp2).ops) == run["CNOT1"] == cnot
    for start, stored in ((np.array(theta_mp2), run["energies_1"]), (np.full(size, 0.01), run["energies_2"])):
        e0 = energy(start)
        assert abs(e0 - stored[0]) < abs_tol, (e0, stored[0])
        for k in range(size):            # the optimiser's first gradient: E(start + h e_k), k = 0..K-1
            t = start.copy()
            t[k] += FD
            assert abs((energy(t) - e0) - (stored[k + 1] - stored[0])) < 1e-12, k
    for key in ("1", "2"):
        e = energy(np.array(run["theta_optimized_result" + key]))
        assert abs(e - run["minimum_energy_result%s_guess" % key]) < abs_tol, key
    return energy, theta_mp2


def test_k5_molecule_level_numbers(runs, h4):
    r = runs["h4_quccsd"]
    assert abs(h4.e_hf - r["info"]["HF"]) < 1e-10
    assert abs(h4.mp2_energy() - r["info"]["MP2"]) < 1e-8                 # the reference's SCF threshold
    assert abs(h4.ci_ground_state()[0] - r["info"]["FCI"]) < 1e-10
    noons, _ = h4.natural_occupations()
    assert np.abs(noons - np.array(r["noons"])).max() < 1e-6              # CISD density; FCI differs at 1e-3
    _, psi = h4.ci_ground_state()
    assert len(psi) == 70                                                 # C(8, 4) determinants


def test_k5_quccsd_full_space_oracle_engine(runs, h4):
    with engine(OracleStatevector):
        _check_quccsd_run(h4.problem(active=False), runs["h4_quccsd"], 5e-9, 292)


def test_k5a_quccsd_active_space_oracle_engine(runs, h4):
    r = runs["h4_quccsd_active"]
    p = h4.problem(active=True)
    assert p.nbqbits == r["active_qubits"] == 6 and p.n_elec == 2 and p.frozen == [0] and p.active == [1, 2, 3]
    assert np.abs(np.array(p.thresholds) - np.array(r["thresholds"])).max() < 1e-6
    with engine(OracleStatevector):
        _check_quccsd_run(p, r, 2e-8, 70)


@pytest.mark.gpu
def test_k5_quccsd_on_gpu(runs, h4, gpu_lib):
    """the same pins through the HIP gate program (fused kernel at 8 / 6 qubits) + the whole stored BFGS run replayed:
    get_energies from theta_MP2 and from 0.01 must land on the stored minima"""
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
    with engine(None):
        _check_quccsd_run(h4.problem(active=False), runs["h4_quccsd"], 5e-9, 292)
        _check_quccsd_run(h4.problem(active=True), runs["h4_quccsd_active"], 2e-8, 70)
        p = h4.problem(active=False)
        ham = p.jw_hamiltonian()
        size, cluster_ops, _, theta_mp2, hf = p.uccsd()
        r = runs["h4_quccsd"]
        with contextlib.redirect_stdout(io.StringIO()):
            it, res = EnergyUCC().get_energies(ham, cluster_ops, hf, theta_mp2, [0.01] * size, r["info"]["FCI"])
        assert res["CNOT1"] == res["CNOT2"] == 292 and res["len_op1"] == 26
        assert abs(it["minimum_energy_result1_guess"][0] - r["minimum_energy_result1_guess"]) < 1e-7   # BFGS tol 1e-5
        assert abs(it["minimum_energy_result2_guess"][0] - r["minimum_energy_result2_guess"]) < 1e-7
        # the optimiser's trajectory itself: the stored evaluations of the first iterations, in call order
        n = min(len(res["energies_1"]), 3 * 27)
        assert np.abs(np.array(res["energies_1"][:n]) - np.array(r["energies_1"][:n])).max() < 5e-7


def test_k5_whole_bfgs_run_oracle_engine(runs, h4):
    """get_energies of the mirror on the oracle engine: same number of function evaluations as the stored run and the
    stored energies along the way (the trajectories separate only at the level the reference's SCF threshold allows)"""
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
    r = runs["h4_quccsd_active"]
    p = h4.problem(active=True)
    ham = p.jw_hamiltonian()
    size, cluster_ops, _, theta_mp2, hf = p.uccsd()
    with engine(OracleStatevector), contextlib.redirect_stdout(io.StringIO()):
        it, res = EnergyUCC().get_energies(ham, cluster_ops, hf, theta_mp2, [0.01] * size, r["info"]["FCI"])
    assert res["CNOT1"] == 70
    for key, mine in (("1", res["energies_1"]), ("2", res["energies_2"])):
        stored = r["energies_" + key]
        n = min(len(mine), len(stored), 4 * (size + 1))
        # offsets grow from 9e-9 (first evaluation) as the two optimisers' line searches drift apart
        assert np.abs(np.array(mine[:n]) - np.array(stored[:n])).max() < 5e-7, key
        assert abs(it["minimum_energy_result%s_guess" % key][0] - r["minimum_energy_result%s_guess" % key]) < 1e-7


# ------------------------------------------------------------------------------------------------ pools
def test_pool_sizes_pinned_by_the_reference_tests():
    """ref:tests/test_main_ucc.py:15 (36), test_main_ucc_active_space.py:15 (18), test_main_quccsd.py:15 (26),
    test_main_quccsd_active_space.py:15 (8), test_main_fermionic_adapt.py:11,15 (175 / 69), test_main_qubit_adapt.py:11,14
    (70; 50 = the 'random' YXXX-family pool of generate_pool_without_cluster)"""
    assert pools.singlet_upccgsd(4, "JW", 2)[0] == 36          # H2/6-31G
    assert pools.singlet_upccgsd(3, "JW", 2)[0] == 18          # H4 active space: 3 orbitals
    assert pools.spin_complement_gsd(4, 4)[0] == 175 and pools.spin_complement_gsd(2, 3)[0] == 69
    assert pools.singlet_gsd(2, 4, "JW")[0] == 70
    assert pools.qubit_pool("random", 8, rng=np.random.default_rng(0))[0] == 50
    mol = chem.molecule("H4")
    mol.rhf()
    assert mol.problem(False).uccsd()[0] == 26 and mol.problem(True).uccsd()[0] == 8


def test_pool_operators_are_antihermitian_and_spin_adapted():
    """every pool operator is anti-Hermitian, conserves particle number and S_z; the singlet pools commute with S^2"""
    from oracle import dense
    n_orb = 3
    n = 2 * n_orb
    number = sum(dense.operator_matrix(_number_op(n, q)) for q in range(n))
    sz = sum((0.5 if q % 2 == 0 else -0.5) * dense.operator_matrix(_number_op(n, q)) for q in range(n))
    for name, fn in (("singlet_sd", pools.singlet_sd), ("singlet_gsd", pools.singlet_gsd), ("uccgsd", pools.uccgsd),
                     ("twin", pools.spin_complement_gsd_twin)):
        size, fermi, spin = fn(2, n_orb, "JW")
        assert size == len(fermi) == len(spin) > 0
        nonzero = 0
        for op in spin:
            m = dense.operator_matrix(op, with_constant=False)
            assert np.abs(m + m.conj().T).max() < 1e-12, name
            assert np.abs(m @ number - number @ m).max() < 1e-12, name
            if name != "uccgsd":
                assert np.abs(m @ sz - sz @ m).max() < 1e-12, name
            nonzero += np.abs(m).max() > 1e-12
        assert nonzero > 0          # the raw enumerations contain identically-zero operators (p == q, ...)
    # normalisation of the singlet doubles (generator_excitations.py:353-357): unit 2-norm of the fermionic coefficients
    _, fermi, _ = pools.singlet_gsd(2, n_orb, "JW")
    for op in fermi[6:]:
        assert abs(sum(abs(t.coeff) ** 2 for t in op.terms) - 1.0) < 1e-12


def _number_op(n, q):
    from openvqe_amd.operators import Hamiltonian, Term
    return Hamiltonian(n, [Term(-0.5, "Z", [q])], 0.5)


def test_normal_ordering_against_matrices_and_the_reference_module():
    """normal_ordered_terms: same operator (JW matrices), canonical form; where the reference tree is present its own
    fermion_util.order_fermionic_term must give the same term lists on the stand-ins"""
    from openvqe_amd.fermionic import FermionHamiltonian, Term, normal_ordered_terms, transform_to_jw_basis
    from oracle import dense
    rng = np.random.default_rng(11)
    n = 5
    cases = [Term(1.0, "CcCc", [3, 1, 2, 0]), Term(-2.0, "CcCc", [1, 1, 2, 2]), Term(1.0, "cC", [2, 2]),
             Term(0.5, "CcCc", [0, 1, 1, 0]), Term(1.0, "CCcc", [4, 2, 3, 3]), Term(1.0, "cCcC", [0, 1, 1, 0])]
    for _ in range(40):
        k = int(rng.integers(1, 3))
        ops = "".join(rng.permutation(list("C" * k + "c" * k)))
        cases.append(Term(float(rng.normal()), ops, rng.integers(0, n, 2 * k).tolist()))
    ref_mod = None
    if os.path.isdir("/root/reference"):
        from tests.test_host_logic import reference_module
        ref_mod = reference_module("openvqe.common_files.fermion_util")
    for t in cases:
        got = normal_ordered_terms(t)
        for u in got:
            k = u.op.count("C")
            assert u.op == "C" * k + "c" * (len(u.op) - k)
            assert u.qbits[:k] == sorted(set(u.qbits[:k])) and u.qbits[k:] == sorted(set(u.qbits[k:]))
        a = dense.operator_matrix(transform_to_jw_basis(FermionHamiltonian(n, [t])))
        b = dense.operator_matrix(transform_to_jw_basis(FermionHamiltonian(n, got))) if got else np.zeros_like(a)
        # the canonical form drops pure numbers (contractions of everything): compare up to a multiple of the identity
        d = a - b
        assert np.abs(d - np.eye(1 << n) * d[0, 0]).max() < 1e-12, (t.op, t.qbits)
        if ref_mod is not None and "c" in t.op and t.op.index("c") < len(t.op):
            try:
                theirs = ref_mod.order_fermionic_term(t)
            except (ValueError, IndexError):
                continue      # the reference's helper needs a 'c' in every intermediate term
            assert [(u.op, u.qbits, u.coeff) for u in theirs] == [(u.op, u.qbits, u.coeff) for u in got], (t.op, t.qbits)


def test_pools_equal_the_reference_generators_on_the_stand_ins():
    """the reference's own generator_excitations / qubit_pool modules, imported unchanged on the qat stand-ins (build
    container only), enumerate the same operators as openvqe_amd.pools"""
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree not present (GPU box)")
    from tests.test_host_logic import reference_module
    gen = reference_module("openvqe.common_files.generator_excitations")
    qp = reference_module("openvqe.common_files.qubit_pool")

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            # (the stand-in transform marks an identically-zero operator with one zero-coefficient string)
            tx = [(t.op, tuple(t.qbits), complex(t.coeff)) for t in x.terms if t.coeff != 0]
            ty = [(t.op, tuple(t.qbits), complex(t.coeff)) for t in y.terms if t.coeff != 0]
            assert len(tx) == len(ty)
            for (o1, q1, c1), (o2, q2, c2) in zip(tx, ty):
                assert o1 == o2 and q1 == q2 and abs(c1 - c2) < 1e-14
    with contextlib.redirect_stdout(io.StringIO()):
        for name, mine, args in (("singlet_sd", pools.singlet_sd, (2, 4)), ("singlet_gsd", pools.singlet_gsd, (2, 3)),
                                 ("uccgsd", pools.uccgsd, (2, 2)), ("spin_complement_gsd_twin", pools.spin_complement_gsd_twin, (2, 3))):
            s1, f1, sp1 = getattr(gen, name)(*args, "JW")
            s2, f2, sp2 = mine(*args, "JW")
            assert s1 == s2, name          # identically-zero operators are kept on both sides (pinned pool sizes)
            same(f1, f2)
            same(sp1, sp2)
        s1, f1, sp1 = gen.singlet_upccgsd(4, "JW", 2)
        s2, f2, sp2 = pools.singlet_upccgsd(4, "JW", 2, with_fermionic=True)
        assert s1 == s2 == 36
        same(sp1, sp2)
        for cond in ("full", "full_without_Z", "reduced_without_Z"):
            n1, p1 = qp.QubitPool().generate_pool_from_cluster(cond, f1, 8)
            n2, p2 = pools.generate_pool_from_cluster(cond, f2, 8)
            assert n1 == n2, cond
            same(p1, p2)
        # the remaining kinds of the qubit-pool dispatcher (ref:openvqe/common_files/qubit_pool.py:1249-1266)
        for n in (4, 6, 8):
            for kind in ("two", "four", "minimal"):
                n1, p1 = qp.QubitPool().generate_pool_without_cluster(kind, nbqbits=n)
                n2, p2 = pools.qubit_pool(kind, n)
                assert n1 == n2, (kind, n)
                same(p1, p2)
        n1, p1 = qp.QubitPool().generate_pool_without_cluster("pure_with_symmetry", nbqbits=8, molecule_symbol="H4")
        n2, p2 = pools.qubit_pool("pure_with_symmetry", 8, molecule_symbol="H4")
        assert n1 == n2 == 11
        same(p1, p2)
        assert pools.qubit_pool("pure_with_symmetry", 8, molecule_symbol="LiH") == (0, [])
        _, _, source = gen.singlet_sd(2, 4, "JW")
        n1, p1 = qp.QubitPool().generate_pool_without_cluster("without_Z_from_generator", nbqbits=8, qubit_pool=source)
        n2, p2 = pools.qubit_pool("without_Z_from_generator", 8, source_pool=source)
        assert n1 == n2
        same(p1, p2)
        n1, p1 = qp.QubitPool().generate_pool_without_cluster("eight", nbqbits=8, qubit_pool=source)
        n2, p2 = pools.qubit_pool("eight", 8, source_pool=source)
        assert n1 == n2
        same(p1, p2)


# ------------------------------------------------------------------------------------------------ K6b: derived qubit pool
def _k6b_problem():
    mol = chem.molecule("H2")
    mol.rhf()
    size, fermi, _ = pools.singlet_upccgsd(mol.nao, "JW", 2, with_fermionic=True)
    with contextlib.redirect_stdout(io.StringIO()):
        n_pool, qubit_pool = pools.generate_pool_from_cluster("reduced_without_Z", fermi, 2 * mol.nao)
    return mol.jw_hamiltonian(), qubit_pool, mol.hf_init(), n_pool


def _check_k6b(runs, energy):
    r = runs["h2_631g_upccgsd_run2"]
    stored = np.array(r["energies_2_first19"])
    t0 = np.full(18, r["theta0"])
    e0 = energy(t0)
    assert abs(e0 - stored[0]) < 3e-8
    for k in range(18):
        t = t0.copy()
        t[k] += FD
        assert abs((energy(t) - e0) - (stored[k + 1] - stored[0])) < 1e-12, k


def test_k6b_derived_reduced_pool_oracle_engine(runs):
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    ham, pool, hf, n_pool = _k6b_problem()
    assert n_pool == runs["h2_631g_upccgsd_run2"]["len_op2"] == 18
    assert [p.terms[0].op for p in pool] == ["YX"] * 12 + ["YXYY"] * 6 and all(p.terms[0].coeff == -1.0 for p in pool)
    with engine(OracleStatevector):
        ucc = EnergyUCC()
        _check_k6b(runs, lambda th: ucc.ucc_action(th, ham, pool, hf, []))


@pytest.mark.gpu
def test_k6b_get_energies_with_the_derived_pool_on_gpu(runs, gpu_lib):
    """ref:openvqe/algorithms/ucc.py:58-81 on the GPU: cluster operators (x 1j) for the first BFGS, the DERIVED
    'reduced_without_Z' pool as ``pool_generator`` for the second; both stored minima of the notebook"""
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    ham, pool, hf, n_pool = _k6b_problem()
    mol = chem.molecule("H2")
    _, spin_ops = pools.singlet_upccgsd(mol.nao, "JW", 2)
    k3 = json.load(open(os.path.join(GOLD, "k3_k5_notebook_traces.json")))
    with engine(None):
        ucc = EnergyUCC()
        _check_k6b(runs, lambda th: ucc.ucc_action(th, ham, pool, hf, []))
        theta = [0.01] * n_pool
        with contextlib.redirect_stdout(io.StringIO()):
            it, res = EnergyUCC().get_energies(ham, [o * 1j for o in spin_ops], pool, hf, theta, theta, k3["h2_631g_info"]["FCI"])
    assert res["len_op1"] == res["len_op2"] == 18 and res["CNOT1"] == res["CNOT2"] == 608
    assert abs(it["minimum_energy_result1_guess"][0] - k3["h2_631g_upccgsd"]["minimum_energy_result1_guess"]) < 1e-6
    assert abs(it["minimum_energy_result2_guess"][0] - runs["h2_631g_upccgsd_run2"]["minimum_energy_result2_guess"]) < 1e-6


# ------------------------------------------------------------------------------------------------ K3a: active-space ADAPT
def _replay_active_adapt(runs, cls):
    import openvqe_amd.adapt.fermionic_adapt_vqe as fa
    mol = chem.molecule("H2")
    mol.rhf()
    p = mol.problem(active=True)
    ham = p.jw_hamiltonian()
    _, pool = pools.spin_complement_gsd(p.n_elec, p.nbqbits // 2)
    with engine(cls), contextlib.redirect_stdout(io.StringIO()):
        it, res = fa.fermionic_adapt_vqe(None, None, None, ham, pool, p.hf_init(), 1, -1.1516885475166085, "COBYLA", 1e-6,
                                         "norm", 1e-2, 35)
    return p, it, res


def _check_k3a(runs, p, it, res):
    r = runs["h2_631g_adapt_active"]
    assert p.nbqbits == r["active_qubits"] == 8
    assert np.abs(np.array(p.thresholds) - np.array(r["thresholds"])).max() < 1e-9
    assert np.abs(np.array(p.noons_full[::2]) - np.array(r["noons"])).max() < 1e-7
    assert res["indices"] == r["result"]["indices"] == [23, 32, 38]
    assert it["CNOTs"] == r["iterations"]["CNOTs"] and it["Hadamard"] == r["iterations"]["Hadamard"]
    assert np.abs(np.array(it["energies"]) - np.array(r["iterations"]["energies"])).max() < 2e-8       # COBYLA 1e-6
    assert np.abs(np.array(it["norms"]) - np.array(r["iterations"]["norms"])).max() < 2e-5
    assert np.abs(np.array(it["fidelity"]) - np.array(r["iterations"]["fidelity"])).max() < 1e-6


def test_k3a_active_space_adapt_trace_oracle_engine(runs):
    from oracle import dense
    p, it, res = _replay_active_adapt(runs, OracleStatevector)
    psi = np.zeros(1 << p.nbqbits)
    psi[p.hf_init()] = 1
    # <HF|H|HF> in the natural-orbital basis, printed by the reference before the loop
    assert abs(dense.expectation(p.jw_hamiltonian(), psi) - runs["h2_631g_adapt_active"]["reference_energy"]) < 1e-12
    _check_k3a(runs, p, it, res)


@pytest.mark.gpu
def test_k3a_active_space_adapt_trace_on_gpu(runs, gpu_lib):
    _check_k3a(runs, *_replay_active_adapt(runs, None))
