"""Bravyi-Kitaev and parity-basis mappings next to Jordan-Wigner (the three `transform` values of the reference,
ref:openvqe/common_files/molecule_factory.py:349-356, ref:…generator_excitations.py:17-22): canonical anticommutation
relations, isospectral Hamiltonians, the recoded Hartree-Fock determinant, and encoding-independence of everything the
hot path computes — UCC energies at fixed parameters and the ADAPT gradient screen."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest

from openvqe_amd import chem, fermion, pools
from oracle import dense

ENCODINGS = ("JW", "parity_basis", "Bravyi-Kitaev")


@pytest.mark.parametrize("n", [1, 2, 5, 6])
def test_ladder_operators_satisfy_the_anticommutation_relations(n):
    for tr in ENCODINGS:
        a = [np.asarray(dense.operator_matrix(fermion.psum_to_hamiltonian(n, fermion.encoded_product(n, [(p, False)], tr),
                                                                          tol=-1.0), sparse=False)) for p in range(n)]
        for i in range(n):
            for j in range(n):
                anti = a[i] @ a[j].conj().T + a[j].conj().T @ a[i]
                assert np.abs(anti - (np.eye(1 << n) if i == j else 0.0)).max() < 1e-12, (tr, i, j)
                assert np.abs(a[i] @ a[j] + a[j] @ a[i]).max() < 1e-12, (tr, i, j)
        # the vacuum of the encoded register is |0...0> and a+_p |vac> is the recoded single occupation
        for p in range(n):
            vec = np.asarray(a[p].conj().T[:, 0]).ravel()
            idx = fermion.recode_occupation(1 << (n - 1 - p), n, tr)
            assert abs(abs(vec[idx]) - 1.0) < 1e-12 and np.abs(np.delete(vec, idx)).max() < 1e-12


def test_hamiltonian_spectrum_hf_energy_and_ucc_energies_do_not_depend_on_the_encoding():
    mol = chem.molecule("H2")
    mol.rhf()
    p = mol.problem(active=False)
    n = p.nbqbits
    rng = np.random.default_rng(8)
    theta = rng.uniform(-0.2, 0.2, 6)
    picks = [38, 32, 29, 23, 2, 57]                       # operators of the stored ADAPT trace + one more
    ref = None
    for tr in ENCODINGS:
        ham = p.spin_hamiltonian(tr)
        hm = dense.operator_matrix(ham)
        spectrum = np.linalg.eigvalsh(hm.toarray())
        hf = p.hf_init(tr)
        psi = np.zeros(1 << n, complex)
        psi[hf] = 1
        assert abs(dense.expectation(ham, psi) - mol.e_hf) < 1e-10
        _, pool = pools.spin_complement_gsd(p.n_elec, n // 2, tr)
        assert len(pool) == 175
        gens = [pool[k] * 1j for k in picks]
        e = dense.ucc_energy(ham, gens, hf, theta)
        state = dense.ucc_state(n, hf, gens, theta)
        grads = dense.fermionic_pool_gradients([dense.operator_matrix(a, with_constant=False) for a in pool], hm, state)
        if ref is None:
            ref = (spectrum, e, np.array(grads))
        else:
            assert np.abs(spectrum - ref[0]).max() < 1e-10
            assert abs(e - ref[1]) < 1e-11
            assert np.abs(np.array(grads) - ref[2]).max() < 1e-10


def test_reference_stack_runs_with_the_bravyi_kitaev_transform():
    """ref:openvqe/main_ucc.py with transform='Bravyi-Kitaev': the reference's own factory + UCC driver on the stand-ins;
    the optimum is the JW one (stored in ref:notebooks/demo_puccgsd.ipynb)"""
    if not os.path.isdir("/root/reference"):
        pytest.skip("reference tree not present (GPU box)")
    import json
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    from openvqe_amd import qat_compat
    qat_compat.install(force=True)
    import openvqe_amd.backend as be
    import openvqe_amd.evaluator as ev
    import openvqe_amd.qat_compat as qc
    from tests.oracle_backend import OracleStatevector
    saved = [(m, m.Statevector) for m in (be, ev)]
    for m, _ in saved:
        m.Statevector = OracleStatevector
    ev._BACKENDS.clear(); ev._Evaluator._owner.clear(); qc._default_qpu = None
    try:
        if "/root/reference" not in sys.path:
            sys.path.append("/root/reference")
        from openvqe.vqe import VQE
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            algo = VQE.algorithm("ucc", "H2", "sUPCCGSD", "Bravyi-Kitaev", False)
            algo.execute()
    finally:
        for m, s in saved:
            m.Statevector = s
        ev._BACKENDS.clear(); ev._Evaluator._owner.clear(); qc._default_qpu = None
    k6 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "k3_k5_notebook_traces.json")))["h2_631g_upccgsd"]
    assert "Pool size:  36" in buf.getvalue()
    assert abs(algo.result["energies_1"][0] - k6["energies_1_first19"][0]) < 3e-8     # same E(0.01): encoding-independent
    assert abs(algo.iterations["minimum_energy_result1_guess"][0] - k6["minimum_energy_result1_guess"]) < 1e-6


@pytest.mark.gpu
def test_bravyi_kitaev_and_parity_programs_on_gpu(gpu_lib):
    """the encoded Hamiltonian + generators through the HIP path: same energies as Jordan-Wigner, same gradient screen"""
    from openvqe_amd.backend import GRAD_FERMIONIC, Statevector
    mol = chem.molecule("H4")
    mol.rhf()
    p = mol.problem(active=False)
    n = p.nbqbits
    rng = np.random.default_rng(3)
    picks = list(rng.choice(175, 10, replace=False))
    theta = rng.uniform(-0.3, 0.3, len(picks))
    out = {}
    for tr in ENCODINGS:
        ham = p.spin_hamiltonian(tr)
        _, pool = pools.spin_complement_gsd(p.n_elec, n // 2, tr)
        gens = [pool[k] * 1j for k in picks]
        with Statevector(n) as sv:
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, p.hf_init(tr))
            e = sv.energy(theta)
            sv.prepare_state(theta)
            g = sv.pool_gradients(pool, GRAD_FERMIONIC)
        out[tr] = (e, np.array(g))
        assert abs(e - dense.ucc_energy(ham, gens, p.hf_init(tr), theta)) < 1e-11
    for tr in ENCODINGS[1:]:
        assert abs(out[tr][0] - out["JW"][0]) < 1e-11
        assert np.abs(out[tr][1] - out["JW"][1]).max() < 1e-10
