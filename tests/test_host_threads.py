"""host-side housekeeping of the mirrors' driver loops: BLAS thread limits (common_files/host_threads.py) and the evaluator's
one-upload-per-backend rule for the Hamiltonian (evaluator.py)"""
import numpy as np
import pytest

from openvqe_amd.common_files import host_threads
from openvqe_amd.operators import Hamiltonian, Term


def test_usable_cpus_is_positive_and_within_affinity():
    import os
    n = host_threads.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_one_blas_thread_limits_and_restores():
    threadpoolctl = pytest.importorskip("threadpoolctl")
    np.dot(np.ones((8, 8)), np.ones((8, 8)))   # make sure the BLAS is loaded
    before = {m["filepath"]: m["num_threads"] for m in threadpoolctl.threadpool_info() if m["user_api"] == "blas"}
    if not before:
        pytest.skip("no BLAS thread pool visible to threadpoolctl")
    with host_threads.one_blas_thread():
        inside = [m["num_threads"] for m in threadpoolctl.threadpool_info() if m["user_api"] == "blas"]
        assert inside and all(v == 1 for v in inside)
    after = {m["filepath"]: m["num_threads"] for m in threadpoolctl.threadpool_info() if m["user_api"] == "blas"}
    assert after == before


def test_decorator_keeps_signature_and_result():
    @host_threads.on_one_blas_thread
    def f(a, b=2):
        """doc"""
        return a + b
    assert f(1) == 3 and f(1, b=5) == 6 and f.__name__ == "f" and f.__doc__ == "doc"


def test_adapt_entry_points_are_wrapped():
    from openvqe_amd.adapt import fermionic_adapt_vqe as fa, qubit_adapt_vqe as qa
    assert hasattr(fa.fermionic_adapt_vqe, "__wrapped__") and hasattr(qa.qubit_adapt_vqe, "__wrapped__")


def test_evaluator_uploads_a_hamiltonian_once_per_backend(monkeypatch):
    """a new ansatz on the same backend re-loads the program, not the Hamiltonian; an observable edited in place is uploaded again"""
    import openvqe_amd.evaluator as ev
    from tests.oracle_backend import OracleStatevector

    uploads = []

    class Counting(OracleStatevector):
        def set_hamiltonian(self, h):
            uploads.append(len(h.terms))
            return super().set_hamiltonian(h)

    monkeypatch.setattr(ev, "Statevector", Counting)
    ev.release_backends()
    try:
        H = Hamiltonian(2, [Term(0.5, "ZZ", [0, 1]), Term(0.25, "XX", [0, 1])], 0.1)
        g1 = [Hamiltonian(2, [Term(1.0, "XY", [0, 1])], do_clean_up=False)]
        g2 = g1 + [Hamiltonian(2, [Term(1.0, "YX", [0, 1])], do_clean_up=False)]
        e1 = ev.UCCEvaluator(H, g1, 1).energy([0.3])
        e2 = ev.UCCEvaluator(H, g2, 1).energy([0.3, 0.0])
        assert uploads == [2] and abs(e1 - e2) < 1e-12
        H.terms.append(Term(0.125, "Z", [0]))          # edited in place: the fingerprint changes
        e3 = ev.UCCEvaluator(H, g1, 1).energy([0.3])
        assert uploads == [2, 3] and abs(e3 - e1) > 1e-3
        ev.release_backends()                          # a new handle starts without a Hamiltonian
        ev.UCCEvaluator(H, g1, 1).energy([0.3])
        assert uploads == [2, 3, 3]
    finally:
        ev.release_backends()
