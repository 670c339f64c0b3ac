"""ctypes binding of libovqe_sv.so (include/ovqe_sv.h).  No fallback: if the HIP library is not
built, importing this module's ``lib()`` raises; if no gfx950 device is present every compute
call fails with the library's error text."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libovqe_sv.so")
#: the same source built with -DOVQE_TESTING: it also accepts the measurement / fault-injection options (tests/test_gpu_abi.py, tools/)
TESTING_LIB_PATH = os.path.join(_HERE, "lib", "libovqe_sv_testing.so")
if os.environ.get("OVQE_LIB"):     # measurement scripts: OVQE_LIB=testing (or a path) selects the library to load
    LIB_PATH = TESTING_LIB_PATH if os.environ["OVQE_LIB"] == "testing" else os.environ["OVQE_LIB"]

_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_vp = ctypes.c_void_p
_H = ctypes.c_void_p
_int, _i64, _u64, _dbl = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_double


class _OptF64(object):
    """ndpointer that also accepts None (NULL)."""

    @classmethod
    def from_param(cls, obj):
        if obj is None:
            return None
        return _f64p.from_param(obj)


# every symbol declared in include/ovqe_sv.h: name -> (restype, argtypes)
SIGNATURES = {
    "ovqe_version": (_int, []),
    "ovqe_last_error": (ctypes.c_char_p, [_H]),
    "ovqe_device_count": (_int, [ctypes.POINTER(_int)]),
    "ovqe_create": (_int, [_int, _int, ctypes.POINTER(_H)]),
    "ovqe_create_shard": (_int, [_int, _int, _u64, _int, ctypes.POINTER(_H)]),
    "ovqe_create_view": (_int, [_int, _int, ctypes.c_void_p, ctypes.POINTER(_H)]),
    "ovqe_destroy": (_int, [_H]),
    "ovqe_set_stream": (_int, [_H, _vp]),
    "ovqe_set_option": (_int, [_H, ctypes.c_char_p, _i64]),
    "ovqe_state_ptr": (_int, [_H, ctypes.POINTER(_vp)]),
    "ovqe_adopt_state": (_int, [_H, _vp]),
    "ovqe_init_basis": (_int, [_H, _u64]),
    "ovqe_set_state": (_int, [_H, _f64p]),
    "ovqe_get_state": (_int, [_H, _f64p]),
    "ovqe_get_amplitudes": (_int, [_H, _i64, _u64p, _f64p]),
    "ovqe_randomize": (_int, [_H, _u64, _dbl, ctypes.POINTER(_dbl)]),
    "ovqe_norm2": (_int, [_H, ctypes.POINTER(_dbl)]),
    "ovqe_apply_pauli_rotation": (_int, [_H, _u64, _u64, _dbl]),
    "ovqe_apply_pauli_rotations": (_int, [_H, _i64, _u64p, _u64p, _f64p]),
    "ovqe_apply_gate": (_int, [_H, _int, _int, _int, _dbl]),
    "ovqe_expectation": (_int, [_H, _i64, _u64p, _u64p, _f64p, _dbl, ctypes.POINTER(_dbl)]),
    "ovqe_bilinear": (_int, [_H, _vp, _vp, _i64, _u64p, _u64p, _f64p, _OptF64, _f64p]),
    "ovqe_apply_pauli_sum": (_int, [_H, _vp, _vp, _i64, _u64p, _u64p, _f64p, _OptF64, _int]),
    "ovqe_bilinear_batch": (_int, [_H, _vp, _vp, _i64, _i64p, _u64p, _u64p, _f64p, _OptF64, _f64p]),
    "ovqe_xsum_create": (_int, [_H, _i64, _u64p, _u64p, _f64p, _OptF64, _int, ctypes.POINTER(ctypes.c_int32)]),
    "ovqe_xsum_destroy": (_int, [_H, ctypes.c_int32]),
    "ovqe_xsum_partners": (_int, [_H, ctypes.c_int32, _i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(_i64)]),
    "ovqe_xsum_info": (_int, [_H, ctypes.c_int32, ctypes.POINTER(_i64), _int]),
    "ovqe_xsum_expect_local": (_int, [_H, ctypes.c_int32, ctypes.POINTER(_dbl)]),
    "ovqe_xsum_expect_remote": (_int, [_H, ctypes.c_int32, _u64, _u64, _vp]),
    "ovqe_xsum_expect_finish": (_int, [_H, ctypes.c_int32, _f64p]),
    "ovqe_xsum_apply_local": (_int, [_H, ctypes.c_int32, _vp, _dbl]),
    "ovqe_xsum_apply_remote": (_int, [_H, ctypes.c_int32, _u64, _u64, _vp, _vp]),
    "ovqe_set_hamiltonian": (_int, [_H, _i64, _u64p, _u64p, _f64p, _dbl]),
    "ovqe_set_program": (_int, [_H, _i64, _u64p, _u64p, _f64p, _OptF64, _i32p, ctypes.c_int32, _u64]),
    "ovqe_set_gate_program": (_int, [_H, _i64, _i32p, _i32p, _i32p, _f64p, _f64p, _i32p, ctypes.c_int32, _u64]),
    "ovqe_energy": (_int, [_H, _f64p, ctypes.c_int32, ctypes.POINTER(_dbl)]),
    "ovqe_energy_batch": (_int, [_H, _i64, _f64p, ctypes.c_int32, _f64p]),
    "ovqe_energy_batch_device": (_int, [_H, _i64, _vp, ctypes.c_int32, _vp]),
    "ovqe_prepare_state": (_int, [_H, _f64p, ctypes.c_int32]),
    "ovqe_pool_gradients": (_int, [_H, _i64, _i64p, _u64p, _u64p, _f64p, _OptF64, _int, _f64p]),
    "ovqe_apply_exp_pauli_sum": (_int, [_H, _i64, _u64p, _u64p, _f64p, _OptF64, _dbl]),
    "ovqe_time_pauli_rotation": (_int, [_H, _u64, _u64, _dbl, _int, _int, ctypes.POINTER(_dbl)]),
    "ovqe_last_batch_ms": (_int, [_H, ctypes.POINTER(_dbl)]),
    "ovqe_get_support": (_int, [_H, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "ovqe_last_support": (_int, [_H, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]),
    "ovqe_energy_gradient": (_int, [_H, _f64p, ctypes.c_int32, ctypes.POINTER(_dbl), _f64p]),
    "ovqe_ground_state": (_int, [_H, _dbl, _int, _u64, ctypes.POINTER(_dbl), ctypes.POINTER(_dbl),
                                 ctypes.POINTER(_int)]),
    "ovqe_sector_ground_state": (_int, [_H, _dbl, _int, _u64, ctypes.POINTER(_dbl), ctypes.POINTER(_dbl),
                                        ctypes.POINTER(_int)]),
    "ovqe_program_info": (_int, [_H, ctypes.POINTER(ctypes.c_int64), _int]),
    "ovqe_get_rotation_program": (_int, [_H, _i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_int64)]),
}

_lib = None


class BackendError(RuntimeError):
    pass


def lib():
    """Load libovqe_sv.so (built by ``__graft_entry__.build()``); raise loudly if it is absent."""
    global _lib
    if _lib is None:
        # ONE HIP runtime per process: PyTorch bundles its own libamdhip64.so (soname libamdhip64.so.7, like the
        # system one).  If torch is imported first, libovqe_sv binds to that already-loaded runtime; the other
        # order would load two runtimes (torch looks its copy up by the unversioned file name) and device pointers
        # could not be shared.  So when torch is installed it is imported before the library is opened.
        # Importing torch costs ~1 s; what has to happen first is only that ITS copy of the HIP runtime is the one in the
        # process: it is opened here by path (RTLD_GLOBAL), torch itself only if that fails.
        import sys
        if "torch" not in sys.modules:
            loaded = False
            try:
                import importlib.util
                spec = importlib.util.find_spec("torch")
                if spec and spec.submodule_search_locations:
                    hip_rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
                    if os.path.exists(hip_rt):
                        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
                        loaded = True
            except (OSError, ImportError, ValueError):
                loaded = False
            if not loaded:
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
        if not os.path.exists(LIB_PATH):
            raise BackendError(
                f"{LIB_PATH} is not built — run `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback for the statevector backend.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        # the one-evaluation-per-call entry point once more with raw pointers: scipy's optimisers call it tens of thousands of times,
        # and converting a numpy array through ndpointer + a fresh c_double per call is 2 us of a 24-us call
        raw = ctypes.CFUNCTYPE(_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p)(("ovqe_energy", L))
        L.ovqe_energy_raw = raw
        _lib = L
    return _lib


def check(rc, handle=None):
    if rc != 0:
        msg = lib().ovqe_last_error(handle)
        raise BackendError(f"libovqe_sv error {rc}: {msg.decode() if msg else ''}")
