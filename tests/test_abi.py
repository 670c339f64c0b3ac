"""CPU tests: the C-ABI shared library loads and exports every symbol include/ovqe_sv.h declares
(no compute calls — there is no GPU here), and fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ovqe_sv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ovqe_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ("ovqe_create", "ovqe_destroy", "ovqe_set_hamiltonian", "ovqe_set_program", "ovqe_set_gate_program",
                 "ovqe_energy", "ovqe_energy_batch", "ovqe_pool_gradients", "ovqe_get_state",
                 "ovqe_apply_pauli_rotation", "ovqe_expectation", "ovqe_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(gpu_lib):
    raw = ctypes.CDLL(os.path.join(ROOT, "openvqe_amd", "lib", "libovqe_sv.so"))
    for name in declared_symbols():
        assert hasattr(raw, name), f"{name} declared in ovqe_sv.h but not exported"


def test_python_binding_covers_every_declared_symbol():
    from openvqe_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_no_device_is_a_loud_error(gpu_lib):
    """Without a GPU, creation must fail with OVQE_ERR_NO_DEVICE and a message — never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from openvqe_amd._lib import BackendError
    from openvqe_amd.backend import Statevector
    with pytest.raises(BackendError) as ei:
        Statevector(4)
    assert "no CPU fallback" in str(ei.value) or "-2" in str(ei.value)


def test_product_never_imports_the_oracle():
    """The product package must not import/call anything under oracle/ (checker only)."""
    pkg = os.path.join(ROOT, "openvqe_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dp, f)
                assert "ovqe_oracle" not in text, os.path.join(dp, f)
