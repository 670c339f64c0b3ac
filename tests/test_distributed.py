"""CPU tests of the multi-GPU path: world_size 2 and 4 over gloo, shard-local arithmetic by an oracle-backed
engine (tests only), so that the partition logic — qubit permutation tracking, half-shard exchange, Belady
victim choice, global-x expectation groups, scalar all-reduce — is exercised without GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import masks


class OracleShardEngine:
    def __init__(self, n_local, n_global, rank):
        self.n_local, self.base = n_local, rank << n_local
        self.tensor = torch.zeros(1 << n_local, dtype=torch.complex128)

    def new_buffer(self, count):
        return torch.empty(count, dtype=torch.complex128)

    def sync(self):
        pass

    def _np(self):
        return self.tensor.numpy()

    def init_basis(self, gidx):
        self.tensor.zero_()
        if (gidx >> self.n_local) == (self.base >> self.n_local):
            self.tensor[gidx & ((1 << self.n_local) - 1)] = 1

    def norm2(self):
        return float((self.tensor.abs() ** 2).sum())

    def rotations(self, xs, zs, phis):
        psi = self._np()
        for x, z, p in zip(xs, zs, phis):
            x, z = int(x), int(z)
            assert x >> self.n_local == 0
            psi[:] = np.cos(p) * psi - 1j * np.sin(p) * masks.pauli_apply(psi, x, z, index_offset=self.base)

    def bilinear(self, xs, zs, coeffs, ket=None):
        bra = self._np()
        total = 0j
        lm = (1 << self.n_local) - 1
        for x, z, c in zip(xs, zs, coeffs):
            x, z = int(x), int(z)
            src = bra if ket is None else ket.numpy()
            # partner's global index = (base ^ x_global) | (i ^ x_local)
            off = (self.base ^ x) & ~lm
            i = np.arange(1 << self.n_local, dtype=np.uint64)
            j = (i ^ np.uint64(x & lm))
            gj = j | np.uint64(off)
            par = gj & np.uint64(z)
            for s in (32, 16, 8, 4, 2, 1):
                par ^= par >> np.uint64(s)
            sign = 1.0 - 2.0 * (par & np.uint64(1)).astype(float)
            ph = (1j) ** (bin(x & z).count("1") % 4)
            total += complex(c) * ph * np.vdot(bra, sign * src[j.astype(np.int64)])
        return total

    # -- planned sums (the engine protocol of openvqe_amd.distributed: masks in the physical bit space of the whole register) -------
    def plan_sum(self, xs, zs, coeffs, chunk_bits):
        sums = self.__dict__.setdefault("_sums", {})
        sid = max(sums, default=-1) + 1
        sums[sid] = {"terms": [(int(x), int(z), complex(c)) for x, z, c in zip(xs, zs, coeffs)], "m": int(chunk_bits), "acc": 0j}
        return sid

    def free_sum(self, sid):
        self._sums.pop(sid)

    def _chunk_form(self, m, d, chunk, x, z):
        """ket index j of chunk `chunk` of the shard of (rank ^ d) -> (own local index i = j ^ x_local, sign, i^ny)"""
        lm = (1 << self.n_local) - 1
        jl = (np.arange(1 << m, dtype=np.uint64) | np.uint64(chunk << m))
        gj = jl | np.uint64((self.base ^ (d << self.n_local)) & ~lm)
        par = gj & np.uint64(z)
        for s in (32, 16, 8, 4, 2, 1):
            par ^= par >> np.uint64(s)
        sign = 1.0 - 2.0 * (par & np.uint64(1)).astype(float)
        return (jl ^ np.uint64(x & lm)).astype(np.int64), sign, (1j) ** (bin(x & z).count("1") % 4)

    def sum_expect_local(self, sid):
        t = [v for v in self._sums[sid]["terms"] if v[0] >> self.n_local == 0]
        return self.bilinear([v[0] for v in t], [v[1] for v in t], [v[2] for v in t]).real if t else 0.0

    def sum_expect_remote(self, sid, d, chunk, ket):
        S = self._sums[sid]
        bra, k = self._np(), ket.numpy()
        for x, z, c in S["terms"]:
            if x >> self.n_local != d:
                continue
            i, sign, ph = self._chunk_form(S["m"], d, chunk, x, z)
            S["acc"] += c * ph * np.vdot(bra[i], sign * k)

    def sum_expect_finish(self, sid):
        out, self._sums[sid]["acc"] = self._sums[sid]["acc"], 0j
        return out

    def sum_apply_local(self, sid, out, ident=0.0):
        psi = self._np()
        acc = ident * psi
        for x, z, c in self._sums[sid]["terms"]:
            if x >> self.n_local == 0:
                acc = acc + c * masks.pauli_apply(psi, x, z, index_offset=self.base)
        out.numpy()[:] = acc

    def sum_apply_remote(self, sid, d, chunk, ket, out):
        S = self._sums[sid]
        o, k = out.numpy(), ket.numpy()
        for x, z, c in S["terms"]:
            if x >> self.n_local != d:
                continue
            i, sign, ph = self._chunk_form(S["m"], d, chunk, x, z)
            o[i] += c * ph * sign * k

    def bilinear_batch(self, offsets, xs, zs, coeffs, bra, ket=None):
        saved = self.tensor
        self.tensor = bra                       # bilinear() contracts self.tensor as the bra
        try:
            # ket None means "this rank's psi shard", i.e. the saved tensor
            return np.array([self.bilinear(xs[a:b], zs[a:b], coeffs[a:b], saved if ket is None else ket)
                             for a, b in zip(offsets[:-1], offsets[1:])])
        finally:
            self.tensor = saved


    # -- chunk-level contractions (partner shards arrive in chunks of 2^m amplitudes; masks on the m low bits only)
    @staticmethod
    def _low_form(m, x, z):
        j = np.arange(1 << m, dtype=np.uint64)
        par = j & np.uint64(z)
        for s in (32, 16, 8, 4, 2, 1):
            par ^= par >> np.uint64(s)
        sign = 1.0 - 2.0 * (par & np.uint64(1)).astype(float)
        return (j ^ np.uint64(x)).astype(np.int64), sign, (1j) ** (bin(x & z).count("1") % 4)

    def sub_bilinear(self, m, bra, bra_off, ket, xs, zs, coeffs):
        b, k = bra.numpy()[bra_off:bra_off + (1 << m)], ket.numpy()
        total = 0j
        for x, z, c in zip(xs, zs, coeffs):
            assert int(x) >> m == 0 and int(z) >> m == 0
            i, sign, ph = self._low_form(m, int(x), int(z))
            total += complex(c) * ph * np.vdot(b[i], sign * k)        # <bra|P|ket> = sum_j conj(bra_{j^x}) i^ny (-1)^{|j&z|} ket_j
        return total

    def sub_bilinear_batch(self, m, offsets, xs, zs, coeffs, bra, bra_off, ket):
        return np.array([self.sub_bilinear(m, bra, bra_off, ket, xs[a:b], zs[a:b], coeffs[a:b])
                         for a, b in zip(offsets[:-1], offsets[1:])])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, out, engine="oracle", chunk_bits=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if chunk_bits is not None:
        os.environ["OVQE_SHARD_CHUNK_BITS"] = str(chunk_bits)   # partner shards in chunks of 2^chunk_bits amplitudes
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvqe_amd.distributed import ShardedStatevector
        rng = np.random.default_rng(seed)
        R, T = 40, (25 if world < 8 else 90)
        g = world.bit_length() - 1

        def xmask():  # X/Y on at most n_local - 1 qubits (chemistry strings carry <= 4), anywhere in the register
            w = int(rng.integers(1, max(2, n - g)))
            return sum(1 << int(b) for b in rng.choice(n, w, replace=False))

        xs = [xmask() for _ in range(R)]
        zs = [int(v) for v in rng.integers(0, 1 << n, R)]
        xs[3] = 0                                   # a diagonal string
        xs[7] = xs[6]                                # a fusable pair
        xs[10] = 1 << (n - 1); zs[10] = 0            # X on the top (global) qubit alone
        if g >= 2:
            xs[14] = 0b11 << (n - 2)                 # x on TWO rank bits at once (two half-shard exchanges for one rotation)
        if g >= 3:
            xs[20] = 0b111 << (n - 3)                # ... and on all THREE rank bits of eight shards
            xs[27] = (0b101 << (n - 3)) | 1
        phis = rng.uniform(-1, 1, R)
        hx = [xmask() if rng.random() < 0.8 else 0 for _ in range(T)]
        hz = [int(v) for v in rng.integers(0, 1 << n, T)]
        hc = rng.normal(size=T)
        hf = int(rng.integers(0, 1 << n))
        # engine = "hip": the product engine, every rank's shard handle on device 0 (tests/test_gpu_distributed.py)
        sv = (ShardedStatevector(n, device=0) if engine == "hip" else
              ShardedStatevector(n, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r)))
        e = sv.energy(hx, hz, hc, 0.25, xs, zs, phis, hf)
        full = sv.gather_state()
        n2 = sv.norm2()
        if rank == 0:
            out.put((e, full, n2, dict(sv.stats), (xs, zs, phis, hx, hz, hc, hf)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 6, 3), (4, 7, None), (2, 3, 1), (8, 8, 3), (8, 10, 4)])
def test_sharded_state_matches_single_process_oracle(world, n, chunk_bits):
    """world size 8 = three rank bits: rotations whose x sits on two and on three of them, <H> with all seven partner groups,
    every partner's shard read in chunks (chunk_bits below the shard size: several chunks per read, bra chunk != ket chunk)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 1234 + n, out, "oracle", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    e, full, n2, stats, (xs, zs, phis, hx, hz, hc, hf) = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if world == 8:
        # Hermitian halving (ShardedStatevector.share_of): of the seven partner groups rank 0 contracts the cross terms with ranks
        # 1, 2, 3 and its alternate share of the diametric rank 4 — four shards read in one chunked pass, the other three partners
        # read rank 0's shard instead
        assert stats["partners_per_read"] == 4
        assert stats["chunk_reads"] == 4 * (1 << (n - 3 - chunk_bits))
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    assert np.abs(full - psi).max() < 1e-12
    assert abs(n2 - 1.0) < 1e-12
    assert abs(e - masks.expectation(psi, hx, hz, hc, 0.25)) < 1e-11
    assert stats["swaps"] >= 1 and stats["full_shard_reads"] >= 1
    # lazy un-swapping + Belady victims: far fewer exchanges than one per global-x rotation
    g = world.bit_length() - 1
    glob = sum(1 for x in xs if x >> (n - g))
    # (one rotation can need an exchange per rank bit its x touches: the multi-bit masks of the world-8 cases)
    assert stats["swaps"] <= (glob if world < 8 else g * glob)


@pytest.mark.parametrize("world", [2, 4, 8, 16])
def test_hermitian_halving_orients_every_pair_once_and_balances_the_ranks(world):
    """ShardedStatevector.share_of: of the two ranks of a pair exactly one contracts each cross x-group (the other neither computes
    nor receives for it), and no rank takes more than W/2 of its W - 1 partners"""
    from openvqe_amd.distributed import ShardedStatevector as S
    groups = list(range(100, 111))          # 11 x-groups per partner
    for d in range(1, world):
        for r in range(world):
            mine, theirs = S.share_of(r, d, world, groups), S.share_of(r ^ d, d, world, groups)
            assert sorted(mine + theirs) == groups          # every group exactly once over the pair
    for r in range(world):
        full = sum(1 for d in range(1, world) if len(S.share_of(r, d, world, groups)) == len(groups))
        part = sum(1 for d in range(1, world) if 0 < len(S.share_of(r, d, world, groups)) < len(groups))
        assert full == world // 2 - 1 and part == 1          # (W - 2) / 2 whole partners + its share of the diametric one


def test_permute_mask():
    from openvqe_amd.distributed import permute_mask
    assert permute_mask(0b1011, [2, 0, 1, 3]) == 0b1101


def _screen_worker(rank, world, port, n, seed, out, engine="oracle", chunk_bits=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if chunk_bits is not None:
        os.environ["OVQE_SHARD_CHUNK_BITS"] = str(chunk_bits)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvqe_amd.distributed import ShardedStatevector
        rng = np.random.default_rng(seed)
        g = world.bit_length() - 1

        def xmask(maxw):
            w = int(rng.integers(1, maxw + 1))
            return sum(1 << int(b) for b in rng.choice(n, w, replace=False))

        R, T, NOPS = 12, (20 if world < 8 else 60), (9 if world < 8 else 24)
        xs = [xmask(max(1, n - g - 1)) for _ in range(R)]
        zs = [int(v) for v in rng.integers(0, 1 << n, R)]
        phis = rng.uniform(-1, 1, R)
        hx = [xmask(n) if rng.random() < 0.85 else 0 for _ in range(T)]      # x anywhere, incl. every global bit
        hz = [int(v) for v in rng.integers(0, 1 << n, T)]
        hc = rng.normal(size=T)
        pool = []
        for k in range(NOPS):
            nt = int(rng.integers(1, 4))
            pool.append(([xmask(n) for _ in range(nt)], [int(v) for v in rng.integers(0, 1 << n, nt)],
                         list(rng.normal(size=nt) + 1j * rng.normal(size=nt))))
        pool[2] = ([0], [int(rng.integers(1, 1 << n))], [1.0 + 0j])          # a diagonal operator
        hf = int(rng.integers(0, 1 << n))
        # engine = "hip": the product engine, every rank's shard handle on device 0 (tests/test_gpu_distributed.py)
        sv = (ShardedStatevector(n, device=0) if engine == "hip" else
              ShardedStatevector(n, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r)))
        sv.init_basis(hf)
        sv.apply_pauli_rotations(xs, zs, phis)
        gf = sv.pool_gradients((hx, hz, hc, 0.3), pool, "fermionic")
        gq = sv.pool_gradients((hx, hz, hc, 0.3), pool, "qubit")
        if rank == 0:
            out.put((gf, gq, dict(sv.stats), (xs, zs, phis, hx, hz, hc, pool, hf)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 6, 2), (4, 7, None), (8, 9, 4)])
def test_sharded_adapt_screen_matches_single_process_oracle(world, n, chunk_bits):
    """sigma = H psi assembled per partner shard + pool gradients per partner shard against the
    dense single-process formulas 2 Re <psi|H A|psi> / 2 |<psi|H P|psi>|; world size 8: all seven partner groups, chunked"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_screen_worker, args=(r, world, port, n, 99 + n, out, "oracle", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    gf, gq, stats, (xs, zs, phis, hx, hz, hc, pool, hf) = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if world == 8:
        assert stats["partners_per_read"] == 7
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    sigma = 0.3 * psi
    for x, z, c in zip(hx, hz, hc):
        sigma = sigma + c * masks.pauli_apply(psi, int(x), int(z))
    want = np.array([sum(c * np.vdot(sigma, masks.pauli_apply(psi, int(x), int(z))) for x, z, c in zip(*op)) for op in pool])
    assert np.abs(gf - 2.0 * want.real).max() < 1e-11
    assert np.abs(gq - 2.0 * np.abs(want)).max() < 1e-11
    assert stats["full_shard_reads"] >= 2      # sigma and the pool contraction both needed partner shards


def _real_worker(rank, world, port, n, seed, out, engine="oracle", chunk_bits=None):
    """a UCC-like program: every rotation string has an odd number of Y, so the state stays real and the exchanges / partner reads
    move real parts only"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if chunk_bits is not None:
        os.environ["OVQE_SHARD_CHUNK_BITS"] = str(chunk_bits)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvqe_amd.distributed import ShardedStatevector
        rng = np.random.default_rng(seed)
        g = world.bit_length() - 1
        R, T = 30, 40

        def odd_y_string():
            w = int(rng.integers(2, max(3, min(5, n - g))))
            bits = [int(b) for b in rng.choice(n, w, replace=False)]
            ny = 1 if w < 3 else int(rng.choice([1, 3]))
            x = sum(1 << b for b in bits)
            z = sum(1 << b for b in bits[:ny])                      # Y on `ny` of the X positions ...
            for b in rng.choice(n, 2, replace=False):                # ... and a few Z elsewhere
                if not (x >> int(b)) & 1:
                    z |= 1 << int(b)
            return x, z

        xs, zs = zip(*[odd_y_string() for _ in range(R)])
        phis = rng.uniform(-1, 1, R)
        hx = [sum(1 << int(b) for b in rng.choice(n, int(rng.integers(1, 4)), replace=False)) if rng.random() < 0.85 else 0 for _ in range(T)]
        hz = [int(v) for v in rng.integers(0, 1 << n, T)]
        hc = rng.normal(size=T)
        hf = int(rng.integers(0, 1 << n))
        res = {}
        # "stored": real STORAGE where the engine offers it (HIP shards: float64 amplitudes, real-amplitude kernels); True / False:
        # complex shards with / without real parts only on the wire
        for variant in ("stored", True, False):
            sv = (ShardedStatevector(n, device=0) if engine == "hip" else
                  ShardedStatevector(n, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r)))
            sv.real_storage = variant == "stored"
            sv.real_transfers = variant is not False
            e = sv.energy(hx, hz, hc, 0.5, list(xs), list(zs), phis, hf)
            stored = sv._storage_real()                               # (before the read-back widens the shard)
            counters = dict(getattr(sv.engine, "counters", {}))
            res[variant] = (e, sv.gather_state(), dict(sv.stats), sv.real, stored, counters)
        if rank == 0:
            out.put((res, (xs, zs, phis, hx, hz, hc, hf)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 7, 3), (4, 8, None), (8, 9, 3)])
def test_real_amplitude_transfers_halve_the_exchanged_bytes(world, n, chunk_bits):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, n, 4242 + n, out, "oracle", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    res, (xs, zs, phis, hx, hz, hc, hf) = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, int(x), int(z), p)
    assert np.abs(psi.imag).max() < 1e-15                     # the premise: odd-Y strings keep a basis state real
    want = masks.expectation(psi, hx, hz, hc, 0.5)
    (e1, full1, st1, real1, _, _), (e0, full0, st0, _, _, _) = res[True], res[False]
    es, fulls, sts, _, stored, _ = res["stored"]
    assert not stored and abs(es - want) < 1e-11 and sts["bytes_sent"] == st1["bytes_sent"]    # (the CPU engine has complex shards only)
    assert real1 and np.abs(full1 - psi).max() < 1e-12 and np.abs(full0 - psi).max() < 1e-12
    assert abs(e1 - want) < 1e-11 and abs(e0 - want) < 1e-11
    assert st1["swaps"] == st0["swaps"] >= 1 and st1["real_exchanges"] == st1["swaps"] and st0["real_exchanges"] == 0
    assert st1["real_chunk_reads"] == st1["chunk_reads"] > 0
    assert st1["bytes_sent"] * 2 == st0["bytes_sent"]


def _program_worker(rank, world, port, n, seed, out, engine="oracle", chunk_bits=None):
    """a compiled program on the sharded register: planned once, evaluated at several parameter vectors"""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if chunk_bits is not None:
        os.environ["OVQE_SHARD_CHUNK_BITS"] = str(chunk_bits)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openvqe_amd.distributed import ShardedStatevector
        rng = np.random.default_rng(seed)
        g = world.bit_length() - 1
        K, R, T = 7, 36, 30
        xs, zs = [], []
        for _ in range(R):
            w = int(rng.integers(2, max(3, min(5, n - g))))
            bits = [int(b) for b in rng.choice(n, w, replace=False)]
            x = sum(1 << b for b in bits)
            z = 1 << bits[0]
            for b in rng.choice(n, 2, replace=False):
                if not (x >> int(b)) & 1:
                    z |= 1 << int(b)
            xs.append(x); zs.append(z)
        coeff = rng.uniform(0.5, 1.5, R)
        pidx = rng.integers(0, K, R)
        hx = [sum(1 << int(b) for b in rng.choice(n, int(rng.integers(1, 4)), replace=False)) if rng.random() < 0.85 else 0 for _ in range(T)]
        hz = [int(v) for v in rng.integers(0, 1 << n, T)]
        hc = rng.normal(size=T)
        hf = int(rng.integers(0, 1 << n))
        thetas = rng.uniform(-0.8, 0.8, (3, K))
        sv = (ShardedStatevector(n, device=0) if engine == "hip" else
              ShardedStatevector(n, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r)))
        prog = sv.compile_program(xs, zs, coeff, pidx, hf, hamiltonian=(hx, hz, hc, 0.75))
        es = [sv.program_energy(prog, th) for th in thetas]
        full = sv.gather_state()                                    # the state of the last evaluation, logical order
        swaps_per_run = sv.stats["swaps"] // len(thetas)
        # the same list through the uncompiled path: same exchanges, same energy
        before = sv.stats["swaps"]
        e_plain = sv.energy(hx, hz, hc, 0.75, xs, zs, coeff * thetas[2][pidx], hf)
        plain_swaps = sv.stats["swaps"] - before
        # planning a LONG list must stay cheap (the scan it replaces was quadratic in the list length)
        long_x = [xs[int(k)] for k in rng.integers(0, R, 6000)]
        t0 = time.perf_counter()
        long_prog = sv.compile_program(long_x, [1] * 6000, np.ones(6000), np.zeros(6000, int), hf)
        t_plan = time.perf_counter() - t0
        if rank == 0:
            out.put((es, full, e_plain, swaps_per_run, plain_swaps, prog["swaps"], t_plan, len(long_prog["steps"]),
                     (xs, zs, coeff, pidx, hx, hz, hc, hf, thetas)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 7, 3), (4, 8, 2), (8, 9, 3)])
def test_compiled_program_on_the_sharded_register(world, n, chunk_bits):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_program_worker, args=(r, world, port, n, 31 + n, out, "oracle", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    es, full, e_plain, swaps_per_run, plain_swaps, planned, t_plan, nsteps, (xs, zs, coeff, pidx, hx, hz, hc, hf, thetas) = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for e, th in zip(es, thetas):
        psi = np.zeros(1 << n, complex)
        psi[hf] = 1
        for x, z, c, k in zip(xs, zs, coeff, pidx):
            psi = masks.rotate(psi, int(x), int(z), c * th[k])
        assert abs(e - masks.expectation(psi, hx, hz, hc, 0.75)) < 1e-11
    assert np.abs(full - psi).max() < 1e-12
    assert abs(e_plain - es[2]) < 1e-12
    assert swaps_per_run == plain_swaps == planned >= 1          # the plan IS what the uncompiled path does, made once
    assert t_plan < 5.0 and nsteps > 0, t_plan


# ---- the deadline over the collective waits (task 6a) -------------------------------------------------------
def test_watchdog_fires_only_without_progress():
    import time

    from openvqe_amd.distributed import DistWatchdog
    fired = []
    wd = DistWatchdog(on_expire=lambda phase, s: fired.append((phase, s)), timeout_s=0.4, exit_code=None)
    for _ in range(6):            # progress every 0.15 s: 0.9 s without firing
        time.sleep(0.15)
        wd.kick("exchange")
    assert not fired and not wd.expired
    time.sleep(1.0)               # ... then silence
    assert fired and fired[0][0] == "exchange" and fired[0][1] >= 0.4 and wd.expired
    wd.stop()


def _stalled_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OVQE_DIST_TIMEOUT_S"] = "3"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    from openvqe_amd import distributed as dd

    def expired(phase, seconds):
        out.put((rank, phase, seconds))
        time.sleep(0.2)           # (let the queue's feeder thread write before os._exit)

    dd.watchdog = dd.DistWatchdog(on_expire=expired)       # exit code 3 through os._exit
    sv = dd.ShardedStatevector(6, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r))
    sv.init_basis(5)
    if rank == 1:
        time.sleep(10 ** 6)       # never posts its half of the exchange
    sv.apply_pauli_rotations([1 << 5], [0], [0.3])          # X on the rank bit: a half-shard exchange with the stalled partner
    out.put((rank, "finished", 0.0))


def test_stalled_partner_ends_the_wait_with_exit_code_3():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stalled_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 3
    assert [g[0] for g in got] == [0, 1]
    assert got[0][1] == "half-shard exchange" and got[0][2] >= 3.0      # rank 0 sat in the exchange's wait
    assert got[1][1] == "start"                                           # rank 1 never made a step


def test_compiled_program_refuses_a_short_parameter_vector():
    """run_program clipped the parameter index instead of raising (the single-GPU path raises)"""
    from openvqe_amd.distributed import ShardedStatevector
    sv = ShardedStatevector(5, engine_factory=lambda nl, ng, r: OracleShardEngine(nl, ng, r))
    prog = sv.compile_program([0b00011, 0b01100, 0b10001], [0b00001, 0b00100, 0b10000], [1.0, 0.5, -1.0], [0, 1, 2], 0b00101)
    assert prog["n_params"] == 3
    sv.run_program(prog, [0.1, 0.2, 0.3])
    with pytest.raises(ValueError, match="expected 3 parameters"):
        sv.run_program(prog, [0.1, 0.2])
    with pytest.raises(ValueError, match="expected 3 parameters"):
        sv.run_program(prog, [])
