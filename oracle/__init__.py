"""TEST INFRASTRUCTURE ONLY — CPU oracle for the OpenVQE statevector hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path (``openvqe_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status: the arithmetic of the reference's hot
path lives in un-vendored third-party packages (myqlm-fermion 1.1.4,
qat-core 1.8.5, myqlm-simulators 1.9.5 — ``/root/reference/requirements.txt:2-17``),
so the oracle restates (a) the reference's *own* in-repo scipy implementation
of operator matrices / exponentials / expectation values
(``openvqe/adapt/qubit_adapt_vqe.py:20-55,81-150``,
``openvqe/adapt/fermionic_adapt_vqe.py:12-122``) and (b) the published myQLM
conventions, and is pinned against every numeric known-answer the reference
holds for this path: the H2/STO-3G Hamiltonian + spectrum + VQE optimum printed in
``notebooks/demo_WSSVQE.ipynb`` (K1), the ``CS_hams.pickle`` Hamiltonians with the
Rotoselect/ADAPT logged minima (K2), and the stored ADAPT / QUCCSD / k-UpCCGSD
notebook traces (K3-K6) through the in-repo integral front-end.
"""
