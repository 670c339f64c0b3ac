"""TEST INFRASTRUCTURE ONLY — CPU oracle for the OpenVQE statevector hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``openvqe_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status (details: DESIGN.md §6): the arithmetic of the reference's hot path lives in un-vendored third-party
packages (myqlm-fermion 1.1.4, qat-core 1.8.5, myqlm-simulators 1.9.5 — ``/root/reference/requirements.txt:2-17``)
and the reference itself is pure Python, so there is nothing to compile into ``oracle/_ref``; the oracle restates
(a) the reference's *own* in-repo scipy implementation of operator matrices / exponentials / expectation values
(``openvqe/adapt/qubit_adapt_vqe.py:20-55,81-150``, ``openvqe/adapt/fermionic_adapt_vqe.py:12-122``) and (b) the
published myQLM conventions, and it is PINNED against the numeric known answers the reference stores, replayed end
to end (tests/test_oracle.py, tests/test_reference_traces.py, fixtures under tests/golden/ made by
tests/golden/make_fixtures.py): K1 the H2/STO-3G Hamiltonian printed by myQLM + spectrum + VQE optimum, K2 the
``CS_hams.pickle`` Hamiltonians and logged minima, K3 the stored H2/6-31G fermionic-ADAPT trace, K4 the first
qubit-ADAPT iteration, K6 the k-UpCCGSD energy trace and CNOT count, K5 the H4 molecule data (its QUCCSD energies
remain unpinned: they depend on operator ordering inside myQLM).  In this container the reference's own L1 modules
also run unchanged on the ``qat`` stand-ins with this oracle as engine (tests/test_host_logic.py).
"""
