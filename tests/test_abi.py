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


def test_cdef_header_for_cffi_is_current_and_preprocessor_free():
    """include/ovqe_sv.cdef.h is what `cffi.FFI().cdef()` takes: no preprocessor line at all,
    every declared entry point with the signature of the main header, the OVQE_* constants as one enum — and it is the
    output of tools/make_cdef.py on the committed header.  Parsed with pycparser / cffi when they are importable."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cdef", os.path.join(ROOT, "tools", "make_cdef.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    header = open(os.path.join(ROOT, "include", "ovqe_sv.h")).read()
    cdef = open(os.path.join(ROOT, "include", "ovqe_sv.cdef.h")).read()
    assert cdef == mk.generate(header), "run tools/make_cdef.py"
    body = re.sub(r"/\*.*?\*/", "", cdef, flags=re.S)
    assert "#" not in body and "extern" not in body
    assert sorted(set(re.findall(r"\b(ovqe_[a-z0-9_]+)\s*\(", body))) == declared_symbols()
    norm = lambda t: re.sub(r"\s+", " ", t).strip()
    plain = norm(re.sub(r"/\*.*?\*/", "", header, flags=re.S))
    for decl in re.findall(r"[^;{}]*\bovqe_[a-z0-9_]+\s*\([^;]*\);", body):
        assert norm(decl) in plain, decl
    for name, value in re.findall(r"#define\s+(OVQE_[A-Z_]+)\s+\(?(-?\d+)\)?", header):
        if name != "OVQE_SV_H":
            assert re.search(rf"\b{name} = {value},", body), name
    try:
        import cffi
    except ImportError:
        cffi = None
    if cffi is not None:
        ffi = cffi.FFI()
        ffi.cdef(cdef)
        assert ffi.typeof("ovqe_handle").kind == "pointer"
