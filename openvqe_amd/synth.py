"""Host restatement of the counter-based synthetic-state generator of libovqe_sv (``ovqe_randomize``,
csrc/sv_kernels.hpp ``mix64`` / ``unit_pm1`` / ``k_randomize``): pure integer hashing + exact
int->double conversion, so amplitude i of a 30-qubit synthetic state can be recomputed on the host bit
for bit without holding the state."""
import numpy as np

_M64 = (1 << 64) - 1


def _mix64(v):
    v = (v + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_M64)
    v = (v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    v = (v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return v ^ (v >> np.uint64(31))


def _unit(bits):
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0


def amplitudes(seed, global_indices):
    """unnormalised amplitudes (re, im in [-1,1)) at the given global basis indices"""
    with np.errstate(over="ignore"):
        g = np.asarray(global_indices, dtype=np.uint64)
        h = _mix64(np.uint64(seed) ^ _mix64(g))
        re = _unit(_mix64(h ^ np.uint64(0x1234567)))
        im = _unit(_mix64(h ^ np.uint64(0x89ABCDEF)))
    return re + 1j * im
