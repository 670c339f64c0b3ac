"""Evidence for the explanation of the 1e-8-level offsets between the reference's stored numbers and a tightly
converged replay ("What's weak" 2): the reference's orbitals come from PySCF at its DEFAULT SCF thresholds
(conv_tol 1e-9 on the energy, 3.2e-5 on the gradient), i.e. they are the stationary orbitals rotated by a small angle.

H2/6-31G has one occupied orbital (sigma_g) and, by symmetry, exactly two rotation parameters that a symmetry-conserving
SCF can leave unconverged: occupied sigma_g <-> virtual sigma_g (k02) and the virtual sigma_u pair (k13).  The test FITS
those two numbers on the seven iteration-0 gradients of the stored qubit-ADAPT run (K4, which include the two 2.97e-07
Brillouin residuals that vanish for stationary orbitals) and then PREDICTS, with no further freedom:
  * K6: E(theta = 0.01) of the k-UpCCGSD run — offset 1.1e-8 before, < 5e-10 after;
  * K3: iteration-0 gradient norm and maximum gradient of the fermionic-ADAPT run — 1e-7 before, < 2e-9 after (to
    the digits the notebook prints);
  * K4: the energy after the first optimisation.
The fitted angles are ~1e-7, the size an SCF stopped at PySCF's thresholds leaves (``rhf(tol=1e-9, grad_tol=3.2e-5)`` of
this front-end: 2e-6 with its own DIIS path)."""
import json
import os

import numpy as np

from openvqe_amd import chem, pools
from openvqe_amd.backend import compile_ucc_program
from oracle import masks

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _observables(mol, traces):
    """(K4 seven leading |gradients|, K6 E(0.01), K3 iteration-0 norm and max gradient) for the current orbitals"""
    ham = mol.jw_hamiltonian()
    n, hf = ham.nbqbits, mol.hf_init()
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    hx, hz, hc = ham.packed()
    sig = masks.apply_pauli_sum(psi, hx, hz, hc.real)
    _, qpool = pools.qubit_pool("YXXX", n)
    g4 = []
    for op in qpool:                                  # 2 |<HF| H P |HF>|, ref:openvqe/adapt/qubit_adapt_vqe.py:147-150
        x, z, c = op.packed()
        g4.append(2.0 * abs(np.vdot(sig, masks.apply_pauli_sum(psi, x, z, c))))
    g4 = np.sort(np.array(g4))[::-1][:7]
    _, ops = pools.singlet_upccgsd(mol.nao, "JW", 2)
    rx, rz, rc, pidx, _ = compile_ucc_program(n, [o * 1j for o in ops], 18)
    phi = psi.copy()
    for x, z, c, p in zip(rx, rz, rc, pidx):
        phi = masks.rotate(phi, int(x), int(z), 0.01 * c)
    e6 = masks.expectation(phi, hx, hz, hc.real, ham.constant_coeff)
    _, fpool = pools.spin_complement_gsd(mol.n_elec, mol.nao)
    g3 = []
    for op in fpool:                                  # 2 Re <sigma| A |HF>, ref:openvqe/adapt/fermionic_adapt_vqe.py:67-73
        x, z, c = op.packed()
        g3.append(2.0 * np.vdot(sig, masks.apply_pauli_sum(psi, x, z, c)).real if len(x) else 0.0)
    g3 = np.array(g3)
    return g4, e6, float(np.linalg.norm(g3)), float(np.abs(g3).max())


def test_two_orbital_rotation_angles_explain_every_offset():
    traces = json.load(open(os.path.join(GOLD, "k3_k5_notebook_traces.json")))
    k4 = np.array(traces["h2_631g_qubit_adapt_iter0"]["sorted_gradients"][:7])
    e6_ref = traces["h2_631g_upccgsd"]["energies_1_first19"][0]
    norm_ref = traces["h2_631g_adapt_iterations"]["norms"][0]
    gmax_ref = traces["h2_631g_adapt_iterations"]["Max_gradients"][0]

    def at(k02, k13):
        mol = chem.molecule("H2")
        mol.rhf()
        mol.rotate_orbitals({(0, 2): k02, (1, 3): k13})
        return _observables(mol, traces)
    g0, e0, n0, m0 = at(0.0, 0.0)
    # stationary orbitals: the offsets the round-1 tolerances had to absorb
    assert 1e-8 < np.abs(g0[:5] - k4[:5]).max() < 2e-7 and g0[5] < 1e-9 and abs(k4[5] - 2.969e-7) < 1e-10
    assert 5e-9 < abs(e0 - e6_ref) < 2e-8
    assert 5e-8 < abs(n0 - norm_ref) < 5e-7
    h = 1e-6
    J = np.stack([(at(h, 0.0)[0] - g0) / h, (at(0.0, h)[0] - g0) / h], axis=1)
    kappa = np.linalg.lstsq(J, k4 - g0, rcond=None)[0]
    assert 2e-8 < np.abs(kappa).max() < 1e-6                      # the size PySCF's default thresholds leave
    g1, e1, n1, m1 = at(*kappa)
    assert np.abs(g1 - k4).max() < 5e-10                          # the fit itself: 7 numbers, 2 parameters
    # predictions
    assert abs(e1 - e6_ref) < 5e-10                               # K6 absolute level, 1.1e-8 before
    assert abs(n1 - norm_ref) < 2e-9 and abs(m1 - gmax_ref) < 2e-9  # K3 iteration 0, ~1e-7 before


def test_scf_stopped_at_pyscf_thresholds_leaves_rotated_orbitals():
    tight = chem.molecule("H2")
    tight.rhf()
    loose = chem.molecule("H2")
    loose.rhf(tol=1e-9, grad_tol=3.2e-5)
    S = tight.one_electron()[0]
    rot = tight.mo_coeff.T @ S @ loose.mo_coeff                   # exp(K) between the two orbital sets
    off = np.abs(rot - np.diag(np.diag(rot))).max()
    assert 1e-9 < off < 1e-4 and abs(tight.e_hf - loose.e_hf) < 1e-9
    # symmetry-conserving: only the sigma_g / sigma_g and sigma_u / sigma_u blocks mix
    assert abs(rot[0, 1]) < 1e-12 and abs(rot[0, 3]) < 1e-12 and abs(rot[2, 1]) < 1e-12


def test_h4_offsets_are_not_closed_by_the_two_allowed_rotations():
    """The same question for the QUCCSD pins (K5, H4/STO-3G; VERDICT r2 "What's weak" 1): linear H4 has exactly two
    occupied<->virtual rotations a symmetry-conserving SCF can leave unconverged (sigma_g 0<->2, sigma_u 1<->3).  Stored numbers
    that depend on the orbitals to FIRST order: E(theta_MP2), E(0.01), the two stored optima evaluated at their stored
    parameters, and the MP2 energy (the RHF energy itself agrees to 1e-14: second order).  RESULT — stated, not tolerated away:
    two angles of the size PySCF's thresholds leave (< 1e-6) remove the MP2 offset (3.1e-9 -> < 1e-10) but NOT the others at the
    same time: after the least-squares fit the four energies still differ by 0.5 - 1.5e-9 from the stored ones (before: 2.6e-9,
    3e-10, 1.7e-10, 1.7e-10).  So for H4 the SCF threshold explains the SIZE of the offsets (all < 5e-9, the energy DIFFERENCES
    of the same trace agree to 1e-12) but the two-parameter model does not close at the 5e-10 level; what is left is of the
    order of the notebook's printed precision of theta_optimized (1e-8 in a parameter -> 1e-10 in E) and of myQLM's own MP2
    guess, which this tree cannot see."""
    from openvqe_amd.ucc_family.get_energy_qucc import EnergyUCC
    from tests.oracle_backend import OracleStatevector
    from tests.test_reference_quccsd import engine
    r = json.load(open(os.path.join(GOLD, "k5_k7_notebook_runs.json")))["h4_quccsd"]
    stored = np.array([r["energies_1"][0], r["energies_2"][0], r["minimum_energy_result1_guess"],
                       r["minimum_energy_result2_guess"], r["info"]["MP2"]])

    def observables(k02, k13):
        mol = chem.molecule("H4")
        mol.rhf()
        if k02 or k13:
            mol.rotate_orbitals({(0, 2): k02, (1, 3): k13})
        p = mol.problem(active=False)
        ham = p.jw_hamiltonian()
        size, cluster_ops, _, theta_mp2, hf = p.uccsd()
        q = EnergyUCC()
        e = lambda th: q.action_quccsd(np.array(th), ham, cluster_ops, hf, [])   # noqa: E731
        return np.array([e(theta_mp2), e(np.full(size, 0.01)), e(r["theta_optimized_result1"]),
                         e(r["theta_optimized_result2"]), mol.mp2_energy()]), mol.e_hf

    with engine(OracleStatevector):
        o0, e_hf = observables(0.0, 0.0)
        assert abs(e_hf - r["info"]["HF"]) < 1e-12
        d0 = o0 - stored
        assert np.abs(d0).max() < 5e-9 and abs(d0[4]) > 2e-9          # the offsets at stationary orbitals
        h = 1e-5
        J = np.stack([(observables(h, 0.0)[0] - observables(-h, 0.0)[0]) / (2 * h),
                      (observables(0.0, h)[0] - observables(0.0, -h)[0]) / (2 * h)], axis=1)
        kappa = np.linalg.lstsq(J, -d0, rcond=None)[0]
        assert np.abs(kappa).max() < 1e-6                              # the size an SCF stopped at PySCF's thresholds leaves
        d1 = observables(*kappa)[0] - stored
    assert abs(d1[4]) < 1e-10                                          # the MP2 energy is explained ...
    assert 3e-10 < np.abs(d1[:4]).max() < 2e-9                         # ... the four QUCCSD energies are not, at 5e-10
