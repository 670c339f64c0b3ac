"""The C ABI's exception barrier on the GPU box (include/ovqe_sv.h: "No C++ exception crosses the ABI"):
a host-side C++ exception inside an entry point — a failed allocation, an absurd size — must come back as a negative status with
text in ovqe_last_error, the interpreter stays alive and the handle stays usable."""
import ctypes

import numpy as np
import pytest

from oracle import masks
from tests.util import random_hamiltonian, random_state

pytestmark = pytest.mark.gpu


@pytest.fixture
def testing_lib(gpu_lib, monkeypatch):
    """the OVQE_TESTING build of the same source (fault injection lives only there): handles created inside the test bind to it"""
    from openvqe_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", _lib.TESTING_LIB_PATH)
    monkeypatch.setattr(_lib, "_lib", None)
    return _lib.lib()


def test_product_library_refuses_the_testing_options(gpu_lib):
    from openvqe_amd._lib import BackendError
    from openvqe_amd.backend import Statevector
    with Statevector(6) as sv:
        for name in ("fault_inject", "sector_sweep_dbg", "sector_h_dbg", "sparse_dbg", "sector_debug"):
            with pytest.raises(BackendError, match="unknown option"):
                sv.set_option(name, 1)
        sv.set_option("sector", 1)      # a product option


def test_host_allocation_failure_is_a_status_not_a_dead_process(testing_lib):
    from openvqe_amd._lib import BackendError
    from openvqe_amd.backend import Statevector
    gpu_lib = testing_lib
    n = 10
    rng = np.random.default_rng(5)
    ham = random_hamiltonian(rng, n, 12)
    psi = random_state(rng, n)
    hx, hz, hc = ham.packed()
    want = masks.expectation(psi, hx, hz, hc.real, ham.constant_coeff)
    with Statevector(n) as sv:
        sv.set_state(psi)
        assert abs(sv.expectation(ham) - want) < 1e-11
        sv.set_option("fault_inject", 1)            # the next term-list build throws std::bad_alloc on the host
        other = random_hamiltonian(rng, n, 9)       # (a new operator: the term list of `ham` is cached on the handle)
        with pytest.raises(BackendError) as ei:
            sv.expectation(other)
        assert "-4" in str(ei.value) or "bad_alloc" in str(ei.value)
        assert "bad_alloc" in gpu_lib.ovqe_last_error(sv._h).decode()
        # ... raised once; the handle is intact
        ox, oz, oc = other.packed()
        assert abs(sv.expectation(other) - masks.expectation(psi, ox, oz, oc.real, other.constant_coeff)) < 1e-11
        assert abs(sv.expectation(ham) - want) < 1e-11
        # an absurd size reaches std::vector's own check (std::length_error) before anything is read: status + text
        offsets = np.array([0, 1 << 62], np.int64)
        one = np.zeros(1, np.uint64)
        out = np.zeros(2, np.float64)
        rc = gpu_lib.ovqe_bilinear_batch(sv._h, ctypes.c_void_p(None), ctypes.c_void_p(None), 1, offsets, one, one,
                                         np.zeros(1), np.zeros(1), out)
        assert rc < 0
        text = gpu_lib.ovqe_last_error(sv._h).decode()
        assert "exception" in text or "allocation" in text, text
        assert abs(sv.expectation(ham) - want) < 1e-11


def test_polled_results_equal_synchronised_ones(gpu_lib):
    """"poll_result" (default): the host of a lone evaluation watches the mapped slot its last kernel writes; 0: it synchronises the
    stream — same energies on the support-compacted kernel (14 qubits) and on the sector tables (18 qubits), hundreds of calls in a row"""
    from openvqe_amd import fermion
    from openvqe_amd.backend import Statevector
    for m, o in ((7, 3), (9, 4)):
        ham, gens, hf = fermion.synthetic_molecule(m, o, seed=60 + m)
        thetas = np.random.default_rng(m).uniform(-0.3, 0.3, (150, len(gens)))
        got = {}
        for poll in (1, 0):
            with Statevector(2 * m) as sv:
                sv.set_option("poll_result", poll)
                sv.set_hamiltonian(ham)
                sv.set_ucc_program(gens, hf)
                got[poll] = np.array([sv.energy(t) for t in thetas])
                batch = sv.energy_batch(thetas[:64])
            assert np.abs(batch - got[poll][:64]).max() < 1e-11
        assert np.array_equal(got[0], got[1])


def test_tiled_expectation_of_a_dense_complex_register_census_and_forms(testing_lib):
    """<psi|H|psi> of a DENSE complex 26-qubit state (a shard of the partitioned register, a random state) through the tile cover:
    the first sweep's census finds no sparse tile, the remaining sweeps run in the dense LDS layout (two workgroups per CU) — against
    the same cover with the census off, with per-lane items instead of per-wave entries, and against the kernels that take one pass
    per x-group (tile_bits = 0); testing build: the switches are measurement options"""
    from openvqe_amd.backend import Statevector
    from openvqe_amd.operators import Hamiltonian, Term
    n = 26
    rng = np.random.default_rng(26)
    terms = []
    for _ in range(60):
        qs = sorted(rng.choice(n, int(rng.choice([1, 2, 4])), replace=False).tolist())
        ops = "".join(rng.choice(list("XYZ"), len(qs)))
        if ops.count("Y") & 1:          # keep H real-symmetric: even number of Y
            ops = ops.replace("Y", "X", 1)
        terms.append(Term(float(rng.normal()), ops, qs))
    H = Hamiltonian(n, terms, 0.5, do_clean_up=False)
    l1 = sum(abs(t.coeff) for t in terms)
    got, passes = {}, {}
    for label, opts in (("census_entries", {}), ("no_census", {"expect_dense": 0}), ("items", {"tile_flat": 1}),
                        ("items_no_census", {"tile_flat": 1, "expect_dense": 0}), ("one_pass_per_group", {"tile_bits": 0})):
        with Statevector(n) as sv:
            for k, v in opts.items():
                sv.set_option(k, v)
            sv.randomize(2626, 1.0)
            n2 = sv.norm2()                 # (the synthetic state comes unnormalised: its norm sets the scale of the tolerance)
            got[label] = sv.expectation(H)
            passes[label] = sv.last_passes()[0]
    for label in got:
        if label != "one_pass_per_group":     # a cover of at least four sweeps: the census applies
            assert passes[label] >= 4, passes
    ref = got["one_pass_per_group"]
    for label, e in got.items():
        assert abs(e - ref) < 1e-12 * l1 * max(1.0, n2), (label, e, ref, n2)


def test_one_wave_form_of_the_fused_gradient(testing_lib):
    """ovqe_energy_gradient on the compact support takes the workgroup kernel (k_sparse_grad_wg) whenever the row tables exist; the
    one-wave kernel behind it (k_sparse_grad: programs whose row tables are not built) is reached here by switching the workgroup form
    off (testing option "sparse_wg") — same energy, same derivatives up to the order of the additions"""
    from openvqe_amd import chem, fermion
    from openvqe_amd.backend import Statevector
    mol = chem.molecule("LIH")
    mol.rhf()
    ham = mol.jw_hamiltonian()
    gens = fermion.uccsd_generators(mol.nao, mol.n_elec // 2)
    hf = mol.hf_init()
    theta = np.random.default_rng(8).uniform(-0.2, 0.2, len(gens))
    out = {}
    for wg in (1, 0):
        with Statevector(ham.nbqbits) as sv:
            sv.set_option("sparse_wg", wg)
            sv.set_hamiltonian(ham)
            sv.set_ucc_program(gens, hf)
            out[wg] = sv.energy_gradient(theta)
            assert sv.program_info()["support"] > 0
    scale = float(np.abs(ham.packed()[2]).sum())
    assert abs(out[0][0] - out[1][0]) < 1e-12 * scale and np.abs(out[0][1] - out[1][1]).max() < 1e-12 * scale
    assert np.abs(out[0][1]).max() > 1e-3
