"""Host BLAS threads around loops that alternate small numpy algebra with device calls.

A multi-threaded BLAS (OpenBLAS under numpy / scipy) starts one worker per visible core and leaves the workers spinning for a
while after every call.  Between device calls that costs twice: the HIP runtime's own threads compete with the spinners (N2 QUCCSD
gradient call 31 -> 89 ms, tools/exp_mirror_eval_n2.py), and in a container whose CPU quota is smaller than the host (the MI355X
boxes: 256 cores visible, `cpu.max` = 16 cores per 100-ms period) the spinners use the quota up and the WHOLE process — the thread
that launches kernels included — is throttled until the next period starts: single HIP calls then take 30-80 ms, all ending on the
100-ms grid (tools/trace_slow_calls.sh: 42 such calls = 2.0 of the 5.0 s of a 30-iteration ADAPT run on N2).  The optimiser loops
of the mirrors therefore run their host algebra — vectors of at most a few thousand numbers — on one BLAS thread."""
from __future__ import annotations

import functools
import os
from contextlib import contextmanager


def usable_cpus():
    """cores this process may keep busy: the affinity mask, capped by the cgroup-v2 CPU quota (`cpu.max`)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    return max(1, n)


@contextmanager
def one_blas_thread():
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:   # nothing to limit with: run as is
        yield
        return
    with threadpool_limits(limits=1, user_api="blas"):
        yield


@contextmanager
def usable_blas_threads():
    """inside ``one_blas_thread``: a dense factorisation that is worth the cores (the reference's ``eigh`` of the whole Hamiltonian,
    4096 x 4096 at 12 qubits) — as many BLAS threads as the CPU quota allows, not as many as the host shows"""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits=usable_cpus(), user_api="blas"):
        yield


def on_one_blas_thread(fn):
    """decorator: the whole call under ``one_blas_thread`` (the ADAPT drivers: screens, fidelity and rebuilds between the optimiser runs
    use numpy as well)"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with one_blas_thread():
            return fn(*args, **kwargs)
    return wrapper
