"""Python face of the HIP statevector backend (one object == one ``ovqe_handle``).

Everything numerical happens in ``libovqe_sv.so``; this module only packs the reference's
operator objects (``.terms[i].{coeff,op,qbits}``) into the mask arrays of
``include/ovqe_sv.h`` and forwards.  No CPU implementation lives here.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from .operators import pack_string, pack_terms

GATE_OPCODES = {"X": 0, "H": 1, "RX": 2, "RY": 3, "RZ": 4, "CNOT": 5}
GRAD_FERMIONIC, GRAD_QUBIT = 0, 1
_REAL_TOL = 1e-12


def _real_coeff(c, what):
    c = complex(c)
    if abs(c.imag) > _REAL_TOL * max(1.0, abs(c.real)):
        raise ValueError(f"{what}: Pauli coefficient {c} is not real (generator must be Hermitian — "
                         "multiply the anti-Hermitian cluster operator by 1j, ref:openvqe/algorithms/ucc.py:30-31)")
    return c.real


def compile_ucc_program(nbqbits, generators, n_params=None):
    """Flatten ``for op_k, theta_k: build_ucc_ansatz([op_k], ...)([theta_k])``
    (ref:openvqe/ucc_family/get_energy_ucc.py:42-45) into rotation arrays: rotation (k, j) is
    exp(-i theta_k c_kj P_kj), k outer / j in ``terms`` order (one Trotter step)."""
    K = len(generators) if n_params is None else min(len(generators), int(n_params))  # zip truncation
    counts = [len(generators[k].terms) for k in range(K)]
    terms = [term for k in range(K) for term in generators[k].terms]
    xs, zs, cc = pack_terms(nbqbits, terms)
    ps = np.repeat(np.arange(K, dtype=np.int32), counts)
    bad = np.abs(cc.imag) > _REAL_TOL * np.maximum(1.0, np.abs(cc.real))
    if bad.any():
        j = int(np.argmax(bad))
        _real_coeff(cc[j], f"generator {int(ps[j])}")   # raises with the reference's hint
    return xs, zs, np.ascontiguousarray(cc.real), ps, K


class Statevector:
    """n-qubit complex128 state resident on one MI355X (or one shard of a distributed state)."""

    def __init__(self, n_qubits, device=0, n_global=0, shard_index=0, view_of=None):
        """``view_of``: device pointer of a caller-owned buffer of 2^n_qubits amplitudes that IS the state (ovqe_create_view: no
        allocation of that size; the buffer must outlive the object)"""
        self._L = _lib.lib()
        self._h = ctypes.c_void_p()
        self._energy_io = None
        self.nbqbits = int(n_qubits) + int(n_global)
        self.n_local = int(n_qubits)
        self.n_global = int(n_global)
        self.shard_index = int(shard_index)
        if view_of is not None:
            if n_global:
                raise ValueError("a view is a single-device handle")
            rc = self._L.ovqe_create_view(n_qubits, device, ctypes.c_void_p(int(view_of)), ctypes.byref(self._h))
        elif n_global:
            rc = self._L.ovqe_create_shard(n_qubits, n_global, shard_index, device, ctypes.byref(self._h))
        else:
            rc = self._L.ovqe_create(n_qubits, device, ctypes.byref(self._h))
        if rc != 0:
            self._h = ctypes.c_void_p()
            _lib.check(rc, None)
        self._K = 0
        # OVQE_OPTIONS="name=value,name=value": tuning knobs (ovqe_set_option) for handles created inside library code —
        # the L1 mirrors and the qat stand-ins own their Statevector objects
        for item in os.environ.get("OVQE_OPTIONS", "").split(","):
            if "=" in item:
                name, value = item.split("=", 1)
                self.set_option(name.strip(), int(value))

    # -- lifecycle ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.ovqe_destroy(self._h)
            self._h = ctypes.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        _lib.check(rc, self._h)

    def set_option(self, name, value):
        self._ck(self._L.ovqe_set_option(self._h, name.encode(), int(value)))

    def set_stream(self, hip_stream_handle):
        self._ck(self._L.ovqe_set_stream(self._h, ctypes.c_void_p(hip_stream_handle)))

    def adopt_state(self, device_ptr):
        self._ck(self._L.ovqe_adopt_state(self._h, ctypes.c_void_p(device_ptr)))

    def state_ptr(self):
        p = ctypes.c_void_p()
        self._ck(self._L.ovqe_state_ptr(self._h, ctypes.byref(p)))
        return p.value

    # -- state ----------------------------------------------------------------------------------
    def init_basis(self, index):
        self._ck(self._L.ovqe_init_basis(self._h, int(index)))

    def set_state(self, psi):
        psi = np.ascontiguousarray(psi, dtype=np.complex128)
        if psi.shape != (1 << self.n_local,):
            raise ValueError("state has the wrong length")
        self._ck(self._L.ovqe_set_state(self._h, psi.view(np.float64)))

    def get_state(self):
        out = np.empty(1 << self.n_local, dtype=np.complex128)
        self._ck(self._L.ovqe_get_state(self._h, out.view(np.float64)))
        return out

    def get_amplitudes(self, local_indices):
        idx = np.ascontiguousarray(local_indices, dtype=np.uint64)
        out = np.empty(idx.shape[0], dtype=np.complex128)
        self._ck(self._L.ovqe_get_amplitudes(self._h, idx.shape[0], idx, out.view(np.float64)))
        return out

    def randomize(self, seed, norm2_total=0.0):
        scale = ctypes.c_double()
        self._ck(self._L.ovqe_randomize(self._h, int(seed), float(norm2_total), ctypes.byref(scale)))
        return scale.value

    def norm2(self):
        out = ctypes.c_double()
        self._ck(self._L.ovqe_norm2(self._h, ctypes.byref(out)))
        return out.value

    # -- unit operations ------------------------------------------------------------------------
    def apply_pauli_rotation(self, x, z, phi):
        self._ck(self._L.ovqe_apply_pauli_rotation(self._h, int(x), int(z), float(phi)))

    def apply_pauli_rotations(self, xs, zs, phis):
        xs = np.ascontiguousarray(xs, np.uint64)
        zs = np.ascontiguousarray(zs, np.uint64)
        phis = np.ascontiguousarray(phis, np.float64)
        self._ck(self._L.ovqe_apply_pauli_rotations(self._h, xs.shape[0], xs, zs, phis))

    def rotate(self, op, qbits, phi):
        """exp(-i phi P) with P given as (pauli string, reference qubit list)."""
        x, z = pack_string(self.nbqbits, op, qbits)
        self.apply_pauli_rotation(x, z, phi)

    def apply_gate(self, name, qubits, angle=0.0):
        """literal gate on reference qubit indices (ref:openvqe/common_files/circuit.py gate set)."""
        n = self.nbqbits
        b0 = n - 1 - int(qubits[0])
        b1 = n - 1 - int(qubits[1]) if len(qubits) > 1 else 0
        self._ck(self._L.ovqe_apply_gate(self._h, GATE_OPCODES[name], b0, b1, float(angle or 0.0)))

    def expectation(self, hamiltonian):
        """Re <psi|H|psi> (+ constant) — the OBS job value (get_energy_ucc.py:46-48)."""
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        coeff = np.array([_real_coeff(c, "observable") for c in cs], np.float64)
        out = ctypes.c_double()
        const = _real_coeff(getattr(hamiltonian, "constant_coeff", 0.0) or 0.0, "observable constant")
        self._ck(self._L.ovqe_expectation(self._h, xs.shape[0], xs, zs, coeff, const, ctypes.byref(out)))
        return out.value

    def bilinear(self, xs, zs, coeff, bra_ptr=None, ket_ptr=None):
        coeff = np.asarray(coeff, np.complex128)
        out = np.zeros(2, np.float64)
        self._ck(self._L.ovqe_bilinear(self._h, ctypes.c_void_p(bra_ptr), ctypes.c_void_p(ket_ptr), len(xs),
                                       np.ascontiguousarray(xs, np.uint64), np.ascontiguousarray(zs, np.uint64),
                                       np.ascontiguousarray(coeff.real), np.ascontiguousarray(coeff.imag), out))
        return complex(out[0], out[1])

    def apply_pauli_sum(self, xs, zs, coeff, out_ptr, ket_ptr=None, accumulate=False):
        """out (+)= sum_t c_t P_t ket on explicit device buffers (ket None = the resident state); masks may carry one
        global x part when ket is the partner's shard"""
        coeff = np.asarray(coeff, np.complex128)
        self._ck(self._L.ovqe_apply_pauli_sum(self._h, ctypes.c_void_p(ket_ptr), ctypes.c_void_p(out_ptr), len(xs),
                                              np.ascontiguousarray(xs, np.uint64), np.ascontiguousarray(zs, np.uint64),
                                              np.ascontiguousarray(coeff.real), np.ascontiguousarray(coeff.imag),
                                              1 if accumulate else 0))

    def bilinear_batch(self, offsets, xs, zs, coeff, bra_ptr=None, ket_ptr=None):
        """[sum_{j in op k} c_j <bra|P_j|ket>] for every operator k, one launch -> complex array"""
        coeff = np.asarray(coeff, np.complex128)
        n_ops = len(offsets) - 1
        out = np.zeros(2 * max(n_ops, 1), np.float64)
        self._ck(self._L.ovqe_bilinear_batch(self._h, ctypes.c_void_p(bra_ptr), ctypes.c_void_p(ket_ptr), n_ops,
                                             np.ascontiguousarray(offsets, np.int64),
                                             np.ascontiguousarray(xs if len(xs) else [0], np.uint64),
                                             np.ascontiguousarray(zs if len(zs) else [0], np.uint64),
                                             np.ascontiguousarray(coeff.real if len(coeff) else [0.0]),
                                             np.ascontiguousarray(coeff.imag if len(coeff) else [0.0]), out))
        return out[0:2 * n_ops:2] + 1j * out[1:2 * n_ops:2]

    # -- Pauli sums planned once for a shard of the partitioned register (ovqe_xsum_*, csrc/cross_host.inc) -------------------
    def xsum_create(self, xs, zs, coeff, chunk_bits):
        """plan of sum_t c_t P_t (masks in the physical index-bit space of the WHOLE register) on this shard -> id"""
        coeff = np.asarray(coeff, np.complex128)
        sid = ctypes.c_int32()
        n = len(xs)
        one = np.zeros(1)
        self._ck(self._L.ovqe_xsum_create(self._h, n, np.ascontiguousarray(xs if n else [0], np.uint64),
                                          np.ascontiguousarray(zs if n else [0], np.uint64),
                                          np.ascontiguousarray(coeff.real) if n else one,
                                          (np.ascontiguousarray(coeff.imag) if n else one) if np.any(coeff.imag != 0.0) else None,
                                          int(chunk_bits), ctypes.byref(sid)))
        return sid.value

    def xsum_destroy(self, sid):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.ovqe_xsum_destroy(self._h, int(sid))

    def xsum_partners(self, sid):
        """[(rank difference d, passes one chunk of that partner costs)]"""
        n = ctypes.c_int64()
        self._ck(self._L.ovqe_xsum_partners(self._h, int(sid), 0, None, None, ctypes.byref(n)))
        d, p = np.zeros(max(n.value, 1), np.uint64), np.zeros(max(n.value, 1), np.int64)
        self._ck(self._L.ovqe_xsum_partners(self._h, int(sid), n.value, d.ctypes.data, p.ctypes.data, ctypes.byref(n)))
        return [(int(d[k]), int(p[k])) for k in range(n.value)]

    def xsum_info(self, sid):
        out = (ctypes.c_int64 * 10)()
        self._ck(self._L.ovqe_xsum_info(self._h, int(sid), out, 10))
        keys = ("local_groups", "local_terms", "local_tile_sweeps", "local_untiled_groups", "partners", "remote_groups",
                "remote_terms", "remote_passes_per_chunk", "tile_bits", "streaming_fallback")
        return dict(zip(keys, [int(v) for v in out]))

    def xsum_expect_local(self, sid):
        out = ctypes.c_double()
        self._ck(self._L.ovqe_xsum_expect_local(self._h, int(sid), ctypes.byref(out)))
        return out.value

    def xsum_expect_remote(self, sid, d, chunk, ket_ptr):
        self._ck(self._L.ovqe_xsum_expect_remote(self._h, int(sid), int(d), int(chunk), ctypes.c_void_p(ket_ptr)))

    def xsum_expect_finish(self, sid):
        out = np.zeros(2, np.float64)
        self._ck(self._L.ovqe_xsum_expect_finish(self._h, int(sid), out))
        return complex(out[0], out[1])

    def xsum_apply_local(self, sid, out_ptr, ident=0.0):
        self._ck(self._L.ovqe_xsum_apply_local(self._h, int(sid), ctypes.c_void_p(out_ptr), float(ident)))

    def xsum_apply_remote(self, sid, d, chunk, ket_ptr, out_ptr):
        self._ck(self._L.ovqe_xsum_apply_remote(self._h, int(sid), int(d), int(chunk), ctypes.c_void_p(ket_ptr),
                                                ctypes.c_void_p(out_ptr)))

    # -- compiled evaluation --------------------------------------------------------------------
    def set_hamiltonian(self, hamiltonian):
        xs, zs, cs = pack_terms(self.nbqbits, hamiltonian.terms)
        coeff = np.array([_real_coeff(c, "observable") for c in cs], np.float64)
        const = _real_coeff(getattr(hamiltonian, "constant_coeff", 0.0) or 0.0, "observable constant")
        self._ck(self._L.ovqe_set_hamiltonian(self._h, xs.shape[0], xs, zs, coeff, const))

    def set_rotation_program(self, xs, zs, coeffs, pidx, n_params, hf_init, phi0=None):
        xs = np.ascontiguousarray(xs, np.uint64)
        self._ck(self._L.ovqe_set_program(self._h, xs.shape[0], xs, np.ascontiguousarray(zs, np.uint64),
                                          np.ascontiguousarray(coeffs, np.float64),
                                          None if phi0 is None else np.ascontiguousarray(phi0, np.float64),
                                          np.ascontiguousarray(pidx, np.int32), int(n_params), int(hf_init)))
        self._K = int(n_params)

    def set_ucc_program(self, generators, hf_init, n_params=None):
        xs, zs, cs, ps, K = compile_ucc_program(self.nbqbits, generators, n_params)
        self.set_rotation_program(xs, zs, cs, ps, K, hf_init)
        return K

    def set_gate_program(self, gates, n_params, hf_init):
        """gates: list of (name, qubits, angle_scale, angle_const, param_index or -1)."""
        n = self.nbqbits
        G = len(gates)
        opc = np.zeros(G, np.int32)
        b0 = np.zeros(G, np.int32)
        b1 = np.zeros(G, np.int32)
        asc = np.zeros(G, np.float64)
        aco = np.zeros(G, np.float64)
        pid = np.full(G, -1, np.int32)
        for g, (name, qubits, scale, const, p) in enumerate(gates):
            opc[g] = GATE_OPCODES[name]
            b0[g] = n - 1 - int(qubits[0])
            b1[g] = n - 1 - int(qubits[1]) if len(qubits) > 1 else 0
            asc[g], aco[g], pid[g] = scale, const, p
        self._ck(self._L.ovqe_set_gate_program(self._h, G, opc, b0, b1, asc, aco, pid, int(n_params), int(hf_init)))
        self._K = int(n_params)

    def energy(self, theta):
        K = self._K
        io = self._energy_io
        if io is None or io[0].shape[0] != max(K, 1):   # parameter buffer + result slot of this handle, their addresses taken once
            buf = np.zeros(max(K, 1), np.float64)
            out = ctypes.c_double()
            io = self._energy_io = (buf, out, ctypes.c_void_p(buf.ctypes.data), ctypes.c_void_p(ctypes.addressof(out)))
        buf, out, buf_p, out_p = io
        theta = np.asarray(theta, np.float64).reshape(-1)
        if theta.shape[0] < K:
            raise ValueError(f"expected {K} parameters")
        buf[:K] = theta[:K]
        rc = self._L.ovqe_energy_raw(self._h, buf_p, K, out_p)
        if rc:
            self._ck(rc)
        return out.value

    def energy_batch(self, thetas):
        thetas = np.ascontiguousarray(thetas, np.float64)
        if thetas.ndim != 2 or thetas.shape[1] != self._K:
            raise ValueError(f"expected a (B, {self._K}) array")
        out = np.empty(thetas.shape[0], np.float64)
        self._ck(self._L.ovqe_energy_batch(self._h, thetas.shape[0], thetas if thetas.size else np.zeros(1),
                                           self._K, out))
        return out

    def energy_batch_device(self, n_batch, theta_ptr, energies_ptr):
        """B evaluations with theta (B x K float64, row-major) and energies (B float64) resident on the device —
        e.g. ``tensor.data_ptr()`` of torch CUDA tensors; returns when the energies are written."""
        self._ck(self._L.ovqe_energy_batch_device(self._h, int(n_batch), ctypes.c_void_p(theta_ptr), self._K,
                                                  ctypes.c_void_p(energies_ptr)))

    def prepare_state(self, theta):
        theta = np.ascontiguousarray(theta, np.float64).reshape(-1)[: self._K]
        self._ck(self._L.ovqe_prepare_state(self._h, theta if self._K else np.zeros(1), self._K))

    def last_batch_ms(self):
        out = ctypes.c_double()
        self._ck(self._L.ovqe_last_batch_ms(self._h, ctypes.byref(out)))
        return out.value

    def get_support(self, capacity=None):
        """(indices, amplitudes) of the non-zero amplitudes, ascending index, listed on the device — or None when the
        list is not available (register below 12 qubits, shard of a distributed register) or longer than ``capacity``
        (default: 1/8 of the register)"""
        cap = int(capacity if capacity is not None else max(1, (1 << self.n_local) // 8))
        idx = np.empty(cap, np.uint64)
        amp = np.empty(cap, np.complex128)
        count = ctypes.c_int64()
        self._ck(self._L.ovqe_get_support(self._h, cap, idx.ctypes.data, amp.ctypes.data, ctypes.byref(count)))
        if count.value < 0 or count.value > cap:
            return None
        return idx[:count.value].copy(), amp[:count.value].copy()

    def last_screen_support(self):
        """non-zero amplitudes the last ``pool_gradients`` call walked instead of the register (-1: the register)"""
        out = ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 0, ctypes.byref(out)))
        return out.value

    def last_screen_sector(self):
        """determinants of the symmetry sector whose materialised Hamiltonian gave sigma = H psi of the last ``pool_gradients``
        call (0: sigma from the register / the tile cover)"""
        out = ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 2, ctypes.byref(out)))
        return out.value

    def last_fci_rounds(self):
        """matrix-vector rounds the last ``sector_ground_state`` call needed to saturate the block of H connected to the
        reference determinant"""
        out = ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 3, ctypes.byref(out)))
        return out.value

    def last_passes(self):
        """(passes over the state, bytes they move by construction) of the last ``apply_pauli_rotations`` / ``bilinear`` /
        ``expectation`` call"""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 4, ctypes.byref(a)))
        self._ck(self._L.ovqe_last_support(self._h, 5, ctypes.byref(b)))
        return a.value, b.value

    SECTOR_FORMS = ("sweep_pairs", "sweep_wide", "sweep_streams", "sweep_regular", "adjoint_pairs", "adjoint_wide", "adjoint_streams",
                    "adjoint_regular", "pair_builder_first", "pair_builder_staged")

    def sector_forms(self):
        """names of the sector-path kernel forms that served this handle since its program was set (ovqe_last_support, which = 6)"""
        out = ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 6, ctypes.byref(out)))
        return {name for bit, name in enumerate(self.SECTOR_FORMS) if (out.value >> bit) & 1}

    def last_exp_support(self):
        """amplitudes the Taylor steps of the last ``apply_exp_pauli_sum`` call ran over (-1: the register)"""
        out = ctypes.c_int64()
        self._ck(self._L.ovqe_last_support(self._h, 1, ctypes.byref(out)))
        return out.value

    def energy_gradient(self, theta):
        """E(theta) and the exact gradient dE/dtheta by the adjoint method (one forward + one backward pass over the
        program for all K derivatives) -> (energy, grad[K])"""
        theta = np.ascontiguousarray(theta, np.float64).reshape(-1)[: self._K]
        if theta.shape[0] != self._K:
            raise ValueError(f"expected {self._K} parameters")
        e = ctypes.c_double()
        grad = np.zeros(max(self._K, 1), np.float64)
        self._ck(self._L.ovqe_energy_gradient(self._h, theta if self._K else np.zeros(1), self._K, ctypes.byref(e), grad))
        return e.value, grad[: self._K]

    def ground_state(self, tol=1e-10, max_iter=3000, seed=20250227):
        """lowest eigenpair of the stored Hamiltonian by device-side Lanczos; the eigenvector is left in the
        state buffer (``get_state``).  -> (energy, residual |H y - E y|, iterations)"""
        e, r, it = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self._L.ovqe_ground_state(self._h, float(tol), int(max_iter), int(seed), ctypes.byref(e),
                                           ctypes.byref(r), ctypes.byref(it)))
        return e.value, r.value, it.value

    def sector_ground_state(self, tol=1e-10, max_iter=1000, seed=20250227):
        """lowest eigenpair of the stored Hamiltonian restricted to the support of the stored program's states (sector tables):
        the FCI energy of the Hartree-Fock determinant's symmetry sector for a number- and spin-conserving ansatz.
        -> (energy, residual |H y - E y|, iterations)"""
        e, r, it = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self._L.ovqe_sector_ground_state(self._h, float(tol), int(max_iter), int(seed), ctypes.byref(e),
                                                  ctypes.byref(r), ctypes.byref(it)))
        return e.value, r.value, it.value

    def program_info(self):
        """shape of the compiled program: ops, rotations, literal gates, sweeps per evaluation, tiled sweeps,
        fused-kernel ops, support size (-1 = not analysed yet)"""
        out = (ctypes.c_int64 * 30)()
        self._ck(self._L.ovqe_program_info(self._h, out, 30))
        keys = ("ops", "rotations", "literal_gates", "sweeps", "tiled_sweeps", "fused_ops", "support",
                "h_tile_sweeps", "h_untiled_groups", "h_entries", "h_merged_terms", "h_pair_terms_per_tile",
                "real_stream", "sp_ops", "sp_pairs", "sp_h_entries",
                "sector_support", "sector_sweeps", "sector_pairs", "sector_h_sweeps", "sector_h_elements", "sector_bytes",
                "sector_circuit_us", "sector_expect_us", "sector_h_stream_bytes", "sector_fci_block",
                "sp_conflicts_discovery_order", "sp_conflicts", "sector_regular_slot_bits", "sector_free_bits")
        return dict(zip(keys, [int(v) for v in out]))

    def rotation_program(self):
        """the stored program as its Pauli-rotation sequence (ovqe_get_rotation_program): a gate program in Clifford-frame form
        comes back with its conjugated strings.  -> (xs, zs, coeffs, phi0s, pidx) numpy arrays in execution order"""
        n = ctypes.c_int64()
        self._ck(self._L.ovqe_get_rotation_program(self._h, 0, None, None, None, None, None, ctypes.byref(n)))
        R = n.value
        xs, zs = np.empty(R, np.uint64), np.empty(R, np.uint64)
        cs, p0, pi = np.empty(R, np.float64), np.empty(R, np.float64), np.empty(R, np.int32)
        self._ck(self._L.ovqe_get_rotation_program(self._h, R, xs.ctypes.data, zs.ctypes.data, cs.ctypes.data, p0.ctypes.data,
                                                   pi.ctypes.data, ctypes.byref(n)))
        return xs, zs, cs, p0, pi

    # -- ADAPT ----------------------------------------------------------------------------------
    def pool_gradients(self, pool_ops, mode):
        """Gradient screen over ``pool_ops`` on the resident state, with the stored Hamiltonian.
        The packed form of the pool is cached (the pool is the same object in every ADAPT iteration)."""
        cache = getattr(self, "_pool_cache", None)
        if cache is not None and cache[0] is pool_ops and cache[1] == len(pool_ops):
            offsets, xs, zs, cs = cache[2]
        else:
            offsets = np.zeros(len(pool_ops) + 1, np.int64)
            xs, zs, cs = [], [], []
            for k, op in enumerate(pool_ops):
                px, pz, pc = pack_terms(self.nbqbits, op.terms)
                xs.append(px)
                zs.append(pz)
                cs.append(pc)
                offsets[k + 1] = offsets[k] + px.shape[0]
            xs = np.concatenate(xs) if xs else np.zeros(0, np.uint64)
            zs = np.concatenate(zs) if zs else np.zeros(0, np.uint64)
            cs = np.concatenate(cs) if cs else np.zeros(0, np.complex128)
            if xs.shape[0] == 0:
                xs, zs, cs = np.zeros(1, np.uint64), np.zeros(1, np.uint64), np.zeros(1, np.complex128)
            self._pool_cache = (pool_ops, len(pool_ops), (offsets, xs, zs, cs))
        out = np.zeros(len(pool_ops), np.float64)
        self._ck(self._L.ovqe_pool_gradients(self._h, len(pool_ops), offsets, xs, zs,
                                             np.ascontiguousarray(cs.real), np.ascontiguousarray(cs.imag),
                                             int(mode), out))
        return out

    def apply_exp_pauli_sum(self, operator, theta, prefactor=1.0):
        """psi <- exp(theta * prefactor * operator) psi, exact (no Trotter splitting)."""
        xs, zs, cs = pack_terms(self.nbqbits, operator.terms)
        cs = cs * prefactor
        if getattr(operator, "constant_coeff", 0.0):
            raise ValueError("apply_exp_pauli_sum: operator with a constant term")
        self._ck(self._L.ovqe_apply_exp_pauli_sum(self._h, xs.shape[0], xs, zs, np.ascontiguousarray(cs.real),
                                                  np.ascontiguousarray(cs.imag), float(theta)))

    def time_pauli_rotation(self, x, z, phi, warmup=3, reps=20):
        out = ctypes.c_double()
        self._ck(self._L.ovqe_time_pauli_rotation(self._h, int(x), int(z), float(phi), int(warmup), int(reps),
                                                  ctypes.byref(out)))
        return out.value


def device_count():
    c = ctypes.c_int()
    _lib.lib().ovqe_device_count(ctypes.byref(c))
    return c.value
