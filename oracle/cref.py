"""TEST INFRASTRUCTURE ONLY — ctypes wrapper of the plain-C oracle (oracle/c/ovqe_oracle.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "libovqe_oracle.so")
_lib = None

_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_c128p = np.ctypeslib.ndpointer(np.complex128, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "c", "ovqe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "c"), "libovqe_oracle.so"])
    return _SO


def use_native_build():
    """bench.py's CPU-baseline leg: (re)build the library with -march=native ON THIS MACHINE and make lib() load it;
    falls back to the portable build when the compiler is missing.  Call before the first lib().  -> flags used"""
    global _SO
    assert _lib is None, "use_native_build() must precede the first lib()"
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(_HERE, "c"), "native"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _SO = os.path.join(_HERE, "c", "libovqe_oracle_native.so")
        return "-O3 -march=native"
    except (OSError, subprocess.CalledProcessError):
        return "-O3 -march=x86-64-v2"


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_init_basis.argtypes = [_c128p, ctypes.c_int, ctypes.c_uint64]
        L.orc_pauli_rotation.argtypes = [_c128p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_double]
        L.orc_pauli_rotation_gates.argtypes = L.orc_pauli_rotation.argtypes
        for f in (L.orc_expectation_termwise, L.orc_expectation_grouped):
            f.argtypes = [_c128p, ctypes.c_int, ctypes.c_int64, _u64p, _u64p, _f64p]
            f.restype = ctypes.c_double
        L.orc_apply_gate.argtypes = [_c128p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.orc_ucc_energy.argtypes = [_c128p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, _u64p, _u64p, _f64p, _i32p,
                                     _f64p, ctypes.c_int64, _u64p, _u64p, _f64p, ctypes.c_double, ctypes.c_int]
        L.orc_ucc_energy.restype = ctypes.c_double
        L.orc_gate_energy.argtypes = [_c128p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, _i32p, _i32p, _i32p, _f64p,
                                      _f64p, _i32p, _f64p, ctypes.c_int64, _u64p, _u64p, _f64p, ctypes.c_double]
        L.orc_gate_energy.restype = ctypes.c_double
        L.orc_ucc_energy_batch.argtypes = [_c128p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int64, _u64p,
                                           _u64p, _f64p, _i32p, ctypes.c_int64, ctypes.c_int, _f64p, ctypes.c_int64,
                                           _u64p, _u64p, _f64p, ctypes.c_double, ctypes.c_int, _f64p]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_set_threads.argtypes = [ctypes.c_int]
        # never more OpenMP threads than CPUs this process may run on (cgroup / affinity limited boxes)
        L.orc_set_threads(usable_cpus())
        _lib = L
    return _lib


def usable_cpus():
    """CPUs this process can really use: affinity mask capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def sort_by_x(xs, zs, cs):
    order = np.argsort(xs, kind="stable")
    return (np.ascontiguousarray(xs[order]), np.ascontiguousarray(zs[order]), np.ascontiguousarray(cs[order]))


def ucc_energy(n, hf_index, rx, rz, rcoef, pidx, theta, hx, hz, hc, constant, mode=0, psi=None):
    """mode 0: fused (C2); mode 1: gate level (C1).  H coefficients must be real."""
    L = lib()
    if psi is None:
        psi = np.empty(1 << n, dtype=np.complex128)
    hx, hz, hc = sort_by_x(np.asarray(hx, np.uint64), np.asarray(hz, np.uint64), np.asarray(hc, np.float64))
    e = L.orc_ucc_energy(psi, n, int(hf_index), len(rx), np.ascontiguousarray(rx, np.uint64),
                         np.ascontiguousarray(rz, np.uint64), np.ascontiguousarray(rcoef, np.float64),
                         np.ascontiguousarray(pidx, np.int32), np.ascontiguousarray(theta, np.float64),
                         len(hx), hx, hz, hc, float(constant), int(mode))
    return e, psi


def gate_energy(n, hf_index, opcode, b0, b1, ascale, aconst, gpidx, theta, hx, hz, hc, constant, psi=None):
    L = lib()
    if psi is None:
        psi = np.empty(1 << n, dtype=np.complex128)
    e = L.orc_gate_energy(psi, n, int(hf_index), len(opcode), np.ascontiguousarray(opcode, np.int32),
                          np.ascontiguousarray(b0, np.int32), np.ascontiguousarray(b1, np.int32),
                          np.ascontiguousarray(ascale, np.float64), np.ascontiguousarray(aconst, np.float64),
                          np.ascontiguousarray(gpidx, np.int32), np.ascontiguousarray(theta, np.float64),
                          len(hx), np.ascontiguousarray(hx, np.uint64), np.ascontiguousarray(hz, np.uint64),
                          np.ascontiguousarray(hc, np.float64), float(constant))
    return e, psi


def ucc_energy_batch(n, hf_index, rx, rz, rcoef, pidx, thetas, hx, hz, hc, constant, mode=0, nthreads=None):
    """B independent evaluations, one host thread each (nested OpenMP inside a thread stays serial)."""
    L = lib()
    nthreads = nthreads or usable_cpus()
    thetas = np.ascontiguousarray(thetas, np.float64)
    B, K = thetas.shape
    nthreads = max(1, min(nthreads, B))
    scratch = np.empty(nthreads << n, dtype=np.complex128)
    hx, hz, hc = sort_by_x(np.asarray(hx, np.uint64), np.asarray(hz, np.uint64), np.asarray(hc, np.float64))
    out = np.empty(B, np.float64)
    L.orc_ucc_energy_batch(scratch, nthreads, n, int(hf_index), len(rx), np.ascontiguousarray(rx, np.uint64),
                           np.ascontiguousarray(rz, np.uint64), np.ascontiguousarray(rcoef, np.float64),
                           np.ascontiguousarray(pidx, np.int32), B, K, thetas, len(hx), hx, hz, hc, float(constant),
                           int(mode), out)
    return out
