"""Full-size (BASELINE.json sizes) parity on the GPU: 30-qubit single-Pauli-string sweeps checked against the
oracle formula on sampled amplitudes (the synthetic state is recomputable on the host bit for bit), plus
size-independent properties: norm conservation, exp(-i phi P) exp(+i phi P) = 1, exp(-i pi P) = -1,
<I> = 1, linearity of the expectation."""
import numpy as np
import pytest

from openvqe_amd import synth
from openvqe_amd.operators import pack_string

pytestmark = pytest.mark.gpu
N = 30


def _strings(n):
    return [("XXXY", [0, 1, 2, 3]), ("XXXY", [n - 4, n - 3, n - 2, n - 1]), ("YXXX", [0, 9, 19, n - 1]),
            ("X" + "Z" * (n - 2) + "Y", list(range(n))), ("Z" * n, list(range(n))), ("Y", [n - 1]), ("X", [n - 6]),
            ("ZXZY", [3, 14, 15, 27])]


def _host_rotate(seed, scale, idx, x, z, phi):
    """oracle formula a'_i = cos(phi) a_i - i sin(phi) i^ny (-1)^{|(i^x)&z|} a_{i^x} on sampled indices"""
    idx = idx.astype(np.uint64)
    a = synth.amplitudes(seed, idx) * scale
    j = idx ^ np.uint64(x)
    b = synth.amplitudes(seed, j) * scale
    par = j & np.uint64(z)
    for s in (32, 16, 8, 4, 2, 1):
        par ^= par >> np.uint64(s)
    sign = 1.0 - 2.0 * (par & np.uint64(1)).astype(float)
    ph = (1j) ** (bin(x & z).count("1") % 4)
    return np.cos(phi) * a - 1j * np.sin(phi) * ph * sign * b


def test_30_qubit_rotation_sampled_parity_and_properties(gpu_lib):
    from openvqe_amd.backend import Statevector
    rng = np.random.default_rng(30)
    seed = 20250227
    with Statevector(N) as sv:
        scale = sv.randomize(seed)
        assert abs(sv.norm2() - 1.0) < 1e-10
        idx = rng.integers(0, 1 << N, 4096).astype(np.uint64)
        idx[:8] = [0, 1, 63, 64, (1 << N) - 1, 1 << (N - 1), (1 << (N - 1)) - 1, 12345]
        base = synth.amplitudes(seed, idx) * scale
        assert np.array_equal(sv.get_amplitudes(idx), base)
        for op, qs in _strings(N):
            x, z = pack_string(N, op, qs)
            phi = float(rng.uniform(0.1, 1.0))
            sv.apply_pauli_rotation(x, z, phi)
            got = sv.get_amplitudes(idx)
            want = _host_rotate(seed, scale, idx, x, z, phi)
            assert np.abs(got - want).max() < 1e-19 + 4 * np.finfo(float).eps * np.abs(want).max(), (op, qs)
            sv.apply_pauli_rotation(x, z, -phi)   # undo
            assert np.abs(sv.get_amplitudes(idx) - base).max() < 8 * np.finfo(float).eps * np.abs(base).max()
        assert abs(sv.norm2() - 1.0) < 1e-10
        # exp(-i pi P) = -1
        x, z = pack_string(N, "XZY", [1, 17, 29])
        sv.apply_pauli_rotation(x, z, np.pi)
        assert np.abs(sv.get_amplitudes(idx) + base).max() < 1e-15 * 64
        sv.apply_pauli_rotation(x, z, np.pi)
        # expectation: <I> = 1, linearity
        xs = np.array([0, *[pack_string(N, o, q)[0] for o, q in _strings(N)[:3]]], np.uint64)
        zs = np.array([0, *[pack_string(N, o, q)[1] for o, q in _strings(N)[:3]]], np.uint64)
        from openvqe_amd.operators import Hamiltonian, Term
        ident = sv.bilinear(xs[:1], zs[:1], [1.0])
        assert abs(ident - 1.0) < 1e-10
        singles = [sv.bilinear(xs[k:k + 1], zs[k:k + 1], [1.0]).real for k in range(1, 4)]
        cs = np.array([0.0, 0.3, -1.7, 2.2])
        combo = sv.bilinear(xs, zs, cs).real
        assert abs(combo - float(np.dot(cs[1:], singles))) < 1e-10
        assert all(abs(v) <= 1.0 + 1e-12 for v in singles)
