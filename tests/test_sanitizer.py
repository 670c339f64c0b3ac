"""CPU sanitizer run of the plain-C oracle (no GPU sanitizers on this pool, so the CPU restatement of
the in-place pair updates is what runs under AddressSanitizer + UBSan): `make asan` of oracle/c, then a child process
with the sanitizer runtime preloaded drives every exported routine on small registers (edge cases: 1 qubit, x on the
top / bottom bit, empty programs, empty Hamiltonians, the threaded batch)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import cref
cref._SO = %(so)r
L = cref.lib()
rng = np.random.default_rng(5)
for n in (1, 2, 5, 9):
    dim = 1 << n
    psi = np.zeros(dim, np.complex128)
    L.orc_init_basis(psi, n, dim - 1)
    for x, z in ((1, 0), (1 << (n - 1), 1), (dim - 1, dim - 1), (0, dim - 1), (0, 0)):
        L.orc_pauli_rotation(psi, n, x, z, 0.3)
        L.orc_pauli_rotation_gates(psi, n, x, z, -0.3)
    for op in range(6):
        L.orc_apply_gate(psi, n, op, n - 1, 0, 0.7) if (op != 5 or n > 1) else None
    assert abs(np.vdot(psi, psi).real - 1.0) < 1e-12
    T = 7
    xs = np.sort(rng.integers(0, dim, T).astype(np.uint64))
    zs = rng.integers(0, dim, T).astype(np.uint64)
    cs = rng.normal(size=T)
    a = L.orc_expectation_termwise(psi, n, T, xs, zs, cs)
    b = L.orc_expectation_grouped(psi, n, T, xs, zs, cs)
    assert abs(a - b) < 1e-12
    assert L.orc_expectation_grouped(psi, n, 0, xs, zs, cs) == 0.0
    R = 5
    rx = rng.integers(0, dim, R).astype(np.uint64); rz = rng.integers(0, dim, R).astype(np.uint64)
    pidx = rng.integers(0, 3, R).astype(np.int32)
    th = rng.normal(size=(6, 3))
    e0, _ = cref.ucc_energy(n, 0, rx, rz, np.ones(R), pidx, th[0], xs, zs, cs, 0.5, 0)
    e1, _ = cref.ucc_energy(n, 0, rx, rz, np.ones(R), pidx, th[0], xs, zs, cs, 0.5, 1)
    assert abs(e0 - e1) < 1e-12
    eb = cref.ucc_energy_batch(n, 0, rx, rz, np.ones(R), pidx, th, xs, zs, cs, 0.5, 0, nthreads=3)
    assert abs(eb[0] - e0) < 1e-12
    eg, _ = cref.gate_energy(n, 0, [1, 3, 4], [0, n - 1, 0], [0, 0, 0], [0.0, 1.0, -2.0], [0.0, 0.1, 0.0], [-1, 0, 2],
                             th[1], xs, zs, cs, 0.0)
    assert np.isfinite(eg)
print("sanitizer-clean")
"""


def test_c_oracle_under_asan_ubsan(tmp_path):
    cdir = os.path.join(ROOT, "oracle", "c")
    subprocess.check_call(["make", "-s", "-C", cdir, "asan"])
    so = os.path.join(cdir, "libovqe_oracle_asan.so")
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan runtime next to gcc")
    env = dict(os.environ, LD_PRELOAD=os.path.realpath(libasan), OMP_NUM_THREADS="3",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": so}], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitizer-clean" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_regular_support_tables_under_asan_ubsan(tmp_path):
    """the host-side planner of the bit-arithmetic sweeps (openvqe_amd/csrc/sv_regular_host.hpp: Z2 symmetries of a program, per-sweep
    dependent bits, SecRegOp records, group words, table-entry maps, blocks of two ops) compiled with g++ under ASan + UBSan:
    tests/cpu/regular_tables_check.cpp replays random sweeps from the tables the way k_sector_sweep_reg does and compares every
    amplitude with the pair-by-pair definition of the ops"""
    src = os.path.join(ROOT, "tests", "cpu", "regular_tables_check.cpp")
    exe = str(tmp_path / "regular_tables_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe, src])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for seed in ("7", "2024"):
        r = subprocess.run([exe, "400", seed], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "regular tables ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
        assert "stream plans ok" in r.stdout, r.stdout[-2000:]      # (the host side of the per-wave streams' plan: runs, class bits)


def test_clifford_frame_compiler_under_asan_ubsan(tmp_path):
    """the host-side compiler of the Clifford-frame form (openvqe_amd/csrc/sv_frame_host.hpp: frame tracker of ovqe_set_gate_program,
    sparse simulation of the Clifford part for its global phase) compiled with g++ under ASan + UBSan: tests/cpu/clifford_frame_check.cpp
    compares the frame form of random gate lists — emitted Pauli rotations, then the Clifford part — with the literal lists on a dense
    state, amplitude by amplitude, and the sparse <hf|C|hf> with the dense one (ref:openvqe/common_files/circuit.py:13-106 is such a list)"""
    src = os.path.join(ROOT, "tests", "cpu", "clifford_frame_check.cpp")
    exe = str(tmp_path / "clifford_frame_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe, src])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for seed in ("11", "4242"):
        r = subprocess.run([exe, "400", seed], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "clifford frame ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
