"""openvqe_amd — MI355X-native statevector backend for OpenVQE's VQE / ADAPT-VQE inner loop.

Scope: the hot path of SURVEY.md §8 only (Pauli-exponential circuit application, <psi|H|psi>,
ADAPT gradient screens) as hand-written gfx950 HIP kernels behind the C ABI of
``include/ovqe_sv.h``, plus the thin Python mirror of the reference's L1 entry points.
"""
from .operators import Hamiltonian, Observable, SpinHamiltonian, Term  # noqa: F401

__version__ = "0.1.0"
