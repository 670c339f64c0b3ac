"""Index-bit-partitioned statevector across the GPUs of one node (one process per GPU, RCCL over xGMI
through ``torch.distributed`` — backend "nccl" IS RCCL on ROCm; "gloo" in the CPU tests).

Layout: W = 2^g ranks, rank r owns the 2^(n-g) amplitudes whose PHYSICAL index has r in its top g bits.
A logical->physical bit permutation is tracked, so "which logical qubits are global" is a planner
decision, not a fixed property of the register.

Per Pauli rotation exp(-i phi P), P = (x, z):
  * z on a global bit  -> a rank-dependent sign inside the local kernel, zero traffic;
  * x only on local bits -> the local HIP sweep of libovqe_sv (shard handle, ``ovqe_create_shard``);
  * x on a global bit  -> that logical qubit is first made local by a HALF-SHARD EXCHANGE with the
    partner rank (swap of one global with one local physical bit: each rank sends the half of its shard
    it no longer owns and receives the half it now owns — S/2 bytes each way over one xGMI link), after
    which the rotation is local.  The swap is NOT undone: the permutation is updated instead, and the
    victim local bit is chosen Belady-style (the qubit whose next X/Y use is farthest away), so a run of
    rotations touching the same qubits pays for one exchange only.

Expectation values: Hamiltonian terms are grouped by the global part of their x mask; the x_g = 0 group
is a local partial sum; the <= 2^g - 1 other groups read their partner's shard in CHUNKS (2^26 amplitudes =
1 GiB by default): chunk c of ALL partners is received as one batch — every xGMI link of the group list
busy at once — double-buffered against the contraction of chunk c - 1 (2 x 7 x 1 GiB instead of two
whole shards), and ``ovqe_bilinear`` on an m-bit sub-register contracts the matching chunk of the own
shard (bra) with the received chunk (ket), the index bits above the chunk folded into the coefficients
on the host; one scalar all-reduce at the end.
The ADAPT gradient screen shards the same way (``apply_hamiltonian`` builds the sigma shard group by
group, ``pool_gradients`` contracts the pool per partner shard in one batched launch each, one
all-reduce of the pool-sized result).  Half-shard exchanges travel in pipelined pieces.  No other
collective exists on the data path.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.distributed as dist


class DistWatchdog:
    """Deadline over the collective waits of a multi-rank run.  RCCL's `Work.wait()` only orders streams and the stream
    synchronisation that follows cannot time out, so a rank whose partner never posts its half of an exchange would sit there
    until somebody kills the job.  A daemon thread checks a deadline that every progress point of the sharded register
    (`kick`: each half-shard exchange, each chunk of a partner read, each all-reduce) pushes `timeout_s` into the future; when
    it passes, `on_expire(phase, seconds)` runs on that thread (the blocked main thread has released the GIL inside the wait)
    and the process leaves with `exit_code` through `os._exit` — no exec, no clean-up that could block again.
    `OVQE_DIST_TIMEOUT_S` (default 120) sets the time."""

    def __init__(self, on_expire=None, timeout_s=None, exit_code=3):
        import os
        import threading
        self.timeout_s = float(os.environ.get("OVQE_DIST_TIMEOUT_S", "120")) if timeout_s is None else float(timeout_s)
        self.on_expire = on_expire
        self.exit_code = exit_code
        self.phase = "start"
        self.expired = False
        self._deadline = time.monotonic() + self.timeout_s
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="ovqe-dist-watchdog", daemon=True)
        self._thread.start()

    def kick(self, phase=None):
        if phase is not None:
            self.phase = phase
        self._deadline = time.monotonic() + self.timeout_s

    def stop(self):
        self._stop.set()

    def _run(self):
        import os
        while not self._stop.wait(min(0.25, self.timeout_s / 4)):
            late = time.monotonic() - self._deadline
            if late > 0:
                self.expired = True
                try:
                    if self.on_expire is not None:
                        self.on_expire(self.phase, self.timeout_s + late)
                finally:
                    if self.exit_code is not None:
                        os._exit(self.exit_code)
                return


#: the watchdog of this process, if the launcher installed one (bench.py does for every multi-rank run)
watchdog = None


def _progress(phase=None):
    if watchdog is not None:
        watchdog.kick(phase)


def permute_mask(mask, perm):
    out = 0
    m = int(mask)
    b = 0
    while m:
        if m & 1:
            out |= 1 << perm[b]
        m >>= 1
        b += 1
    return out


class HipShardEngine:
    """Shard-local compute on one MI355X through the C ABI; the state buffer is a torch tensor so that
    torch.distributed can send/receive slices of it."""

    def __init__(self, n_local, n_global, rank, device):
        from .backend import Statevector
        self.device = torch.device("cuda", device)
        self.tensor = torch.zeros(1 << n_local, dtype=torch.complex128, device=self.device)
        self.sv = Statevector(n_local, device=device, n_global=n_global, shard_index=rank)
        self.sv.adopt_state(self.tensor.data_ptr())
        # RCCL's Work.wait() orders torch's CURRENT stream behind a transfer, nothing else: the handle's kernels must run
        # on that very stream, or a contraction could read a partner shard before it has arrived
        self.stream = torch.cuda.current_stream(self.device)
        self.sv.set_stream(self.stream.cuda_stream)
        self._subs = {}
        # what the local kernels really did: passes over the shard (launches that stream it) and the bytes they move by
        # construction — fused same-x runs and LDS-tiled multi-run sweeps make this smaller than one sweep per rotation
        self.counters = {"rotations": 0, "rotation_passes": 0, "rotation_bytes": 0, "contraction_calls": 0, "contraction_passes": 0,
                         "contraction_bytes": 0}

    def check_stream(self):
        if torch.cuda.current_stream(self.device).cuda_stream != self.stream.cuda_stream:
            raise RuntimeError("ShardedStatevector was created under another torch stream: its shard kernels would not be "
                               "ordered behind RCCL transfers waited for on this one")

    def new_buffer(self, count, dtype=None):
        """scratch of ``count`` amplitudes in the shard's current storage (complex128, or float64 under real storage)"""
        return torch.empty(count, dtype=self.tensor.dtype if dtype is None else dtype, device=self.device)

    @property
    def is_real(self):
        return self.tensor.dtype == torch.float64

    def set_real(self, flag, discard=False):
        """Storage of the shard: 2^n_local complex128 (16 B) or — while every applied rotation keeps a real state real — float64
        (8 B: half the HBM bytes of every sweep, of <H> and of every transfer; option "real_state" of the handle).  The amplitudes
        are converted unless ``discard`` (an initialisation follows)."""
        flag = bool(flag)
        if flag == self.is_real:
            return
        size = self.tensor.numel()
        if discard:
            self.tensor = None
            new = torch.zeros(size, dtype=torch.float64 if flag else torch.complex128, device=self.device)
        elif flag:
            new = torch.view_as_real(self.tensor)[:, 0].contiguous()
        else:
            new = torch.complex(self.tensor, torch.zeros_like(self.tensor))
        self.sync()
        self.tensor = new
        self.sv.adopt_state(new.data_ptr())
        self.sv.set_option("real_state", 1 if flag else 0)
        for sub in self._subs.values():
            sub.close()
        self._subs = {}

    def sync(self):
        torch.cuda.synchronize(self.device)

    def init_basis(self, global_index):
        self.sv.init_basis(global_index)

    def randomize(self, seed, norm2_total=0.0):
        return self.sv.randomize(seed, norm2_total)

    def norm2(self):
        return self.sv.norm2()

    def _count(self, what, sv, calls=1):
        passes, nbytes = sv.last_passes()
        self.counters[what + "_passes"] += passes
        self.counters[what + "_bytes"] += nbytes
        self.counters["rotations" if what == "rotation" else "contraction_calls"] += calls

    def rotations(self, xs, zs, phis):
        self.sv.apply_pauli_rotations(xs, zs, phis)
        self._count("rotation", self.sv, len(xs))

    def bilinear_batch(self, offsets, xs, zs, coeffs, bra, ket=None):
        return self.sv.bilinear_batch(offsets, xs, zs, coeffs, bra_ptr=bra.data_ptr(),
                                      ket_ptr=None if ket is None else ket.data_ptr())

    # -- Pauli sums planned once per (Hamiltonian, permutation): ovqe_xsum_* (csrc/cross_host.inc, csrc/sv_cross.hpp).  Masks in the
    # physical bit space of the whole register; the remote calls only enqueue kernels on the engine's stream
    def plan_sum(self, xs, zs, coeffs, chunk_bits):
        return self.sv.xsum_create(xs, zs, coeffs, chunk_bits)

    def free_sum(self, sid):
        self.sv.xsum_destroy(sid)

    def sum_info(self, sid):
        return self.sv.xsum_info(sid)

    def sum_partners(self, sid):
        return self.sv.xsum_partners(sid)

    def sum_expect_local(self, sid):
        out = self.sv.xsum_expect_local(sid)
        self._count("contraction", self.sv)
        return out

    def sum_expect_remote(self, sid, d, chunk, ket):
        self.sv.xsum_expect_remote(sid, d, chunk, ket.data_ptr())
        self._count("contraction", self.sv)

    def sum_expect_finish(self, sid):
        return self.sv.xsum_expect_finish(sid)

    def sum_apply_local(self, sid, out, ident=0.0):
        self.sv.xsum_apply_local(sid, out.data_ptr(), ident)

    def sum_apply_remote(self, sid, d, chunk, ket, out):
        self.sv.xsum_apply_remote(sid, d, chunk, ket.data_ptr(), out.data_ptr())

    # -- the pool contraction on CHUNKS: 2^m consecutive amplitudes of a shard-sized buffer against a received chunk of the
    # partner's shard; masks live on the m low bits (the host layer folds everything above them into the coefficients)
    def _sub(self, m):
        if m not in self._subs:
            from .backend import Statevector
            # (never used as a state — bra and ket pointers come with every call: a view of the shard, no 2^m-amplitude allocation)
            sub = Statevector(m, device=self.device.index, view_of=self.tensor.data_ptr())
            sub.set_stream(self.stream.cuda_stream)
            self._subs[m] = sub
        return self._subs[m]

    def sub_bilinear_batch(self, m, offsets, xs, zs, coeffs, bra, bra_off, ket):
        return self._sub(m).bilinear_batch(offsets, xs, zs, coeffs, bra_ptr=bra.data_ptr() + 16 * bra_off, ket_ptr=ket.data_ptr())


class ShardedStatevector:
    """n-qubit state over ``dist.get_world_size()`` ranks (a power of two)."""

    def __init__(self, n_qubits, engine_factory=None, group=None, device=None, dry_rank=None):
        """``dry_rank`` = (world, rank): ONE rank of a ``world``-rank register without the others — no process group; a half-shard
        exchange packs, "receives" its own packed half and unpacks, a partner-shard read hands out this rank's own chunks.  The
        amplitudes are then meaningless, but every kernel launch, every pack / unpack copy and every byte that would cross a link
        is what that rank of the real job executes and counts: the per-rank compute time of configs[4] at full shard size, measured
        on one GPU (bench.py ``scale_proxy``)."""
        self.group = group
        self.dry = dry_rank is not None
        if self.dry:
            self.world, self.rank = int(dry_rank[0]), int(dry_rank[1])
        else:
            self.world = dist.get_world_size(group) if dist.is_initialized() else 1
            self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        g = self.world.bit_length() - 1
        if (1 << g) != self.world:
            raise ValueError("world size must be a power of two")
        if n_qubits - g < 1:
            raise ValueError("too few qubits for this many ranks")
        self.n, self.g, self.n_local = int(n_qubits), g, int(n_qubits) - g
        self.perm = list(range(self.n))  # logical index bit -> physical index bit
        if engine_factory is None:
            dev = device if device is not None else torch.cuda.current_device()
            engine_factory = lambda nl, ng, r: HipShardEngine(nl, ng, r, dev)  # noqa: E731
        self.engine = engine_factory(self.n_local, self.g, self.rank)
        self._dist = dist.is_initialized() and not self.dry
        self.stats = {"swaps": 0, "bytes_sent": 0, "full_shard_reads": 0, "chunk_reads": 0, "partners_per_read": 0, "pieces": 0,
                      "swap_s": 0.0, "shard_read_s": 0.0, "real_exchanges": 0, "real_chunk_reads": 0,
                      # seconds this rank's kernels ran, by phase (each section ends with a device synchronisation)
                      "local_sweeps_s": 0.0, "expectation_local_s": 0.0, "expectation_remote_s": 0.0, "apply_s": 0.0}
        # Several ranks on ONE device (the gloo runs of the tests and of bench.py's single-device mode) would time each other's
        # kernels: with a lock file every compute section takes the device alone (flock), so its seconds are this rank's own
        self.compute_lock = None
        self._tmp = None
        self._chunk_bufs = None
        self._chunk_real = None
        self._chunk_send = None
        self._sigma = None
        # Real-amplitude transfers: a basis state evolved by rotations whose strings all carry an odd number of Y — every UCC / ADAPT
        # generator (-i phi P is then a real antisymmetric matrix) — has real amplitudes; while that holds, half-shard exchanges and
        # partner-shard reads move the real parts only: 8 instead of 16 bytes per amplitude over xGMI.  (The shard kernels keep their
        # complex layout; the imaginary parts are exact zeros and are rebuilt as such on arrival.)
        self.real = False
        self.real_transfers = True
        # Real STORAGE (engines that offer it: the HIP engine): a compiled program whose strings all have an odd number of Y runs on
        # float64 shards — local sweeps, <H> and transfers on 8-byte amplitudes; anything that needs complex amplitudes (a rotation
        # with an even number of Y, sigma = H psi, the gradient screens) widens the shard back first
        self.real_storage = True

    # -- helpers ----------------------------------------------------------------------------
    def _compute(self, phase):
        """context of one compute section: its wall time (device synchronised on both sides) goes to stats[phase + "_s"]"""
        import contextlib

        @contextlib.contextmanager
        def section():
            lock = None
            if self.compute_lock:
                import fcntl
                lock = open(self.compute_lock, "a+")
                fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                self.engine.sync()
                t0 = time.perf_counter()
                yield
                self.engine.sync()
                self.stats[phase + "_s"] += time.perf_counter() - t0
            finally:
                if lock is not None:
                    import fcntl
                    fcntl.flock(lock, fcntl.LOCK_UN)
                    lock.close()
        return section()

    def _phys(self, mask):
        return permute_mask(mask, self.perm)

    def _storage_real(self):
        return self.engine.tensor.dtype == torch.float64

    def _choose_storage(self, real_program):
        """before an initialisation: 8-byte amplitudes for a rotation list that keeps them real (engines that offer it)"""
        if hasattr(self.engine, "set_real"):
            want = bool(real_program and self.real_storage and self.n_local >= 2)
            if want != self._storage_real():
                self.engine.set_real(want, discard=True)
                self._tmp = None
                self._chunk_bufs = None

    def _complex_storage(self):
        if self._storage_real():
            self.engine.set_real(False)
            self._tmp = None
            self._chunk_bufs = None

    def _local_mask(self):
        return (1 << self.n_local) - 1

    def _tmp_buffer(self, count):
        if self._tmp is None or self._tmp.numel() < count or self._tmp.dtype != self.engine.tensor.dtype:
            self._tmp = None
            self._tmp = self.engine.new_buffer(count)
        return self._tmp[:count]

    #: pieces of one exchange: transfer k + 1 runs while piece k is unpacked into the shard / packed for sending
    EXCHANGE_PIECES = 8

    def _post_pair(self, snd, rcv, partner, tag=0):
        """post send + receive with ``partner`` -> object with ``wait()``.  RCCL takes the device tensors as they are.
        gloo (tests; CPU engines) does not synchronise with the GPU and talks to raw pointers, so device tensors travel
        through host copies there: sent from a host copy taken now, received into a host buffer that ``wait()`` copies
        to the device."""
        staged = snd.is_cuda and dist.get_backend(self.group) == "gloo"
        if snd.is_cuda and not staged and hasattr(self.engine, "check_stream"):
            self.engine.check_stream()
        s_buf = snd.cpu() if staged else snd
        r_buf = torch.empty(rcv.shape, dtype=rcv.dtype, device="cpu") if staged else rcv
        ops = [dist.P2POp(dist.isend, s_buf, partner, self.group, tag=tag), dist.P2POp(dist.irecv, r_buf, partner, self.group, tag=tag)]
        if self.rank > partner:   # lower rank sends first in a pair (gloo needs an order)
            ops.reverse()
        works = dist.batch_isend_irecv(ops)

        class _Pending:
            def wait(_self):
                for w in works:
                    w.wait()
                if staged:
                    rcv.copy_(r_buf)
                _self.keep = None

        pending = _Pending()
        pending.keep = (s_buf, r_buf)
        return pending

    def _exchange(self, partner, make_send, recv_pieces, on_arrival):
        """piece p: ``make_send(p)`` (pack, on the compute stream) then its send / receive with ``partner`` is posted —
        RCCL orders each transfer behind the pack it depends on and runs the transfers in sequence on its own stream, so
        pack p + 1 and transfer p overlap; then every received piece is handed to ``on_arrival`` as soon as it has
        landed, while the later pieces are still on the link.  Lower rank sends first in a pair (gloo needs an order)."""
        works, keep = [], []
        for p, rcv in enumerate(recv_pieces):
            snd = make_send(p)
            keep.append(snd)
            if self.dry:     # (one rank alone: the piece it would send stands in for the piece it would receive)
                rcv.copy_(snd.view(rcv.shape))

                class _Done:
                    def wait(_self):
                        pass
                works.append(_Done())
                continue
            works.append(self._post_pair(snd, rcv, partner, tag=p))   # one tag per piece: gloo matches by tag
        for p, w in enumerate(works):
            w.wait()
            on_arrival(p)
            _progress()

    def _swap(self, gbit, lbit):
        """exchange physical global bit ``gbit`` with physical local bit ``lbit``: each rank sends the half of its shard
        it no longer owns and receives the half it now owns (S/2 bytes each way on one xGMI link), in EXCHANGE_PIECES
        pipelined pieces; a strided half (lbit below the top local bit) is packed piece by piece, never as a whole"""
        k = gbit - self.n_local
        alpha = (self.rank >> k) & 1
        partner = self.rank ^ (1 << k)
        t = self.engine.tensor.view(1 << (self.n_local - 1 - lbit), 2, 1 << lbit)
        mine_out = t[:, 1 - alpha, :]          # (A, B): the half this rank gives away / receives into
        A, B = mine_out.shape
        half = A * B
        P = self.EXCHANGE_PIECES
        recv = self._tmp_buffer(half)
        if A >= P or A >= B:                    # pieces = row blocks (strided rows of length B)
            P = max(1, min(P, A))               # never a zero-length piece: an empty send/recv pair is still a group launch
            views = [mine_out[(A * p) // P:(A * (p + 1)) // P, :] for p in range(P)]
        else:                                   # few long rows: cut the columns
            P = max(1, min(P, B))
            views = [mine_out[:, (B * p) // P:(B * (p + 1)) // P] for p in range(P)]
        sizes = [v.numel() for v in views]
        starts = [sum(sizes[:p]) for p in range(P)]
        recv_pieces = [recv[starts[p]:starts[p] + sizes[p]] for p in range(P)]
        self.engine.sync()
        _progress("half-shard exchange")
        t_swap = time.perf_counter()

        stored_real = self._storage_real()             # float64 shards travel as they are
        real = self.real and self.real_transfers and not stored_real     # (the same on every rank: the flag follows the rotation list)
        if real:
            rrecv = torch.view_as_real(recv).reshape(-1)[:half]        # the receive buffer's first half, as doubles
            recv_pieces = [rrecv[starts[p]:starts[p] + sizes[p]] for p in range(P)]

        def pack(p):
            v = views[p]
            if real:
                return torch.view_as_real(v)[..., 0].contiguous().view(-1)
            return v.reshape(-1) if v.is_contiguous() else v.contiguous().view(-1)

        def unpack(p):
            if real:
                dst = torch.view_as_real(views[p])
                dst[..., 0].copy_(recv_pieces[p].view(views[p].shape))
                dst[..., 1].zero_()
            else:
                views[p].copy_(recv_pieces[p].view(views[p].shape))

        self._exchange(partner, pack, recv_pieces, unpack)
        self.engine.sync()
        self.stats["swap_s"] += time.perf_counter() - t_swap
        _progress("local sweeps")
        # the logical qubits living on these two physical bits trade places
        la, lb = self.perm.index(gbit), self.perm.index(lbit)
        self.perm[la], self.perm[lb] = lbit, gbit
        self.stats["swaps"] += 1
        self.stats["bytes_sent"] += half * (8 if (real or stored_real) else 16)
        self.stats["real_exchanges"] += 1 if (real or stored_real) else 0
        self.stats["pieces"] += P

    @staticmethod
    def _use_lists(x_logical_seq, n):
        """per logical qubit: the (ascending) rotation numbers whose x mask touches it — next-use queries by bisection"""
        uses = [[] for _ in range(n)]
        for r, x in enumerate(x_logical_seq):
            m, b = int(x), 0
            while m:
                if m & 1:
                    uses[b].append(r)
                m >>= 1
                b += 1
        return uses

    def _localise(self, x_logical_seq, r, uses=None, swap=None):
        """make every X/Y qubit of rotation r local; victims by farthest next X/Y use (Belady).  ``uses``: _use_lists of the
        sequence (else scanned); ``swap``: what performs an exchange (default: the real one; the planner records instead)"""
        import bisect
        lmask = self._local_mask()
        swap = swap or self._swap
        while True:
            xp = self._phys(x_logical_seq[r])
            xg = xp & ~lmask
            if not xg:
                return xp
            gbit = xg.bit_length() - 1
            # candidate local physical bits not touched by this rotation's x
            best, best_next = None, -1
            for lbit in range(self.n_local - 1, -1, -1):
                if (xp >> lbit) & 1:
                    continue
                logical = self.perm.index(lbit)
                nxt = len(x_logical_seq) + 1
                if uses is not None:
                    u = uses[logical]
                    k = bisect.bisect_right(u, r)
                    if k < len(u):
                        nxt = u[k]
                else:
                    for s in range(r + 1, len(x_logical_seq)):
                        if (x_logical_seq[s] >> logical) & 1:
                            nxt = s
                            break
                if nxt > best_next:
                    best, best_next = lbit, nxt
            if best is None:
                raise ValueError("rotation touches more qubits than fit in one shard")
            swap(gbit, best)

    # -- state ------------------------------------------------------------------------------
    def init_basis(self, logical_index):
        self.engine.init_basis(self._phys(logical_index))
        self.real = True

    def randomize(self, seed):
        """synthetic state defined on PHYSICAL indices (bench/scaling use; permutation reset)"""
        self.perm = list(range(self.n))
        self.real = False
        self._complex_storage()
        self.engine.randomize(seed, 1.0)
        n2 = torch.tensor([self.engine.norm2()], dtype=torch.float64, device=self.engine.tensor.device)
        if self._dist:
            dist.all_reduce(n2, group=self.group)
        self.engine.tensor.mul_(1.0 / float(n2.item()) ** 0.5)
        return float(n2.item())

    def norm2(self):
        n2 = torch.tensor([self.engine.norm2()], dtype=torch.float64, device=self.engine.tensor.device)
        if self._dist:
            dist.all_reduce(n2, group=self.group)
        return float(n2.item())

    # -- operations ---------------------------------------------------------------------------
    def apply_pauli_rotations(self, xs, zs, phis):
        """rotations in order, masks in LOGICAL index-bit space of the full register"""
        xs = [int(v) for v in xs]
        zs = [int(v) for v in zs]
        if self.real and any(not (bin(x & z).count("1") & 1) for x, z in zip(xs, zs)):
            self.real = False     # a string with an even number of Y (or a diagonal one) makes the amplitudes complex
        if not self.real:
            self._complex_storage()
        batch_x, batch_z, batch_p = [], [], []

        def flush():
            if batch_x:
                with self._compute("local_sweeps"):
                    self.engine.rotations(np.array(batch_x, np.uint64), np.array(batch_z, np.uint64),
                                          np.array(batch_p, np.float64))
                batch_x.clear(); batch_z.clear(); batch_p.clear()

        uses = self._use_lists(xs, self.n) if len(xs) > 8 else None
        for r in range(len(xs)):
            if self._phys(xs[r]) & ~self._local_mask():
                flush()  # the permutation is about to change: masks already queued used the old one
                xp = self._localise(xs, r, uses)
            else:
                xp = self._phys(xs[r])
            batch_x.append(xp)
            batch_z.append(self._phys(zs[r]))
            batch_p.append(float(phis[r]))
        flush()

    def apply_pauli_rotation(self, x, z, phi):
        self.apply_pauli_rotations([x], [z], [phi])

    def _group_by_partner(self, xs, zs, coeffs):
        """terms grouped by the GLOBAL part of their physical x mask (= the rank difference of the partner shard),
        in a rank-independent order"""
        groups = {}
        for t, (x, z, c) in enumerate(zip(xs, zs, coeffs)):
            xp, zp = self._phys(int(x)), self._phys(int(z))
            groups.setdefault(xp >> self.n_local, []).append((xp, zp, complex(c), t))
        return [(xg, groups[xg]) for xg in sorted(groups)]

    #: log2 of the amplitudes per chunk of a partner-shard read: at most 29 (8 GiB: the x bits of a term ABOVE the chunk cost extra
    #: passes of the cross-shard kernels — 34 qubits on 8 ranks: 44 / 41 / 37 / 32 passes per chunk set at 26 / 27 / 28 / 29 bits — so
    #: chunks are as large as the double buffers allow: 2 x 4 partners x 8 GiB beside a 32-GiB shard) and at least four chunks per
    #: shard (the transfer of chunk c + 1 hides behind the contraction of chunk c); OVQE_SHARD_CHUNK_BITS overrides it (tests)
    CHUNK_BITS = 29

    def _chunk_bits(self):
        import os
        env = os.environ.get("OVQE_SHARD_CHUNK_BITS")
        m = int(env) if env is not None else min(self.CHUNK_BITS, self.n_local - 2)
        return max(1, min(self.n_local, m))

    def _post_multi(self, items):
        """post every (send tensor or None, receive tensor or None, partner rank, tag) of ``items`` as ONE batch -> object with
        ``wait()``.  RCCL runs the transfers of a batch concurrently — one xGMI link per partner — and takes the device tensors as
        they are; gloo (tests; CPU engines) talks to raw host pointers, so device tensors travel through host copies there."""
        some = next(t for it in items for t in it[:2] if t is not None)
        staged = some.is_cuda and dist.get_backend(self.group) == "gloo"
        if some.is_cuda and not staged and hasattr(self.engine, "check_stream"):
            self.engine.check_stream()
        ops, keep = [], []
        sent = {}
        for snd, rcv, partner, tag in items:
            pair = []
            if snd is not None:
                if staged and id(snd) not in sent:
                    sent[id(snd)] = snd.cpu()          # (one host copy of the own chunk serves every destination)
                s_buf = sent[id(snd)] if staged else snd
                pair.append(dist.P2POp(dist.isend, s_buf, partner, self.group, tag=tag))
            if rcv is not None:
                r_buf = torch.empty(rcv.shape, dtype=rcv.dtype, device="cpu") if staged else rcv
                keep.append((r_buf, rcv))
                pair.append(dist.P2POp(dist.irecv, r_buf, partner, self.group, tag=tag))
            if len(pair) == 2 and self.rank > partner:   # lower rank sends first in a pair (gloo needs an order)
                pair.reverse()
            ops += pair
        works = dist.batch_isend_irecv(ops) if ops else []
        hold = list(sent.values())

        class _Pending:
            def wait(_self):
                for w in works:
                    w.wait()
                if staged:
                    for r_buf, rcv in keep:
                        rcv.copy_(r_buf)
                keep.clear()
                hold.clear()

        return _Pending()

    def _partner_chunks(self, partners, send_to=None, after_first_post=None):
        """generator over the chunks of the partners' psi shards: yields (c, [chunk c of the shard of rank ^ d for d in
        ``partners``]); this rank's own chunk c goes to rank ^ d for d in ``send_to`` (default: the same list — a symmetric read;
        the Hermitian halving of <H> reads from some partners and sends to others).  Chunk c + 1 of ALL partners is posted as one
        batch before chunk c is handed out (double buffering: 2 x len(partners) x 2^m amplitudes instead of two whole shards), so
        every link of the list carries traffic at once while the previous chunk is contracted ("several
        partners concurrently in chunks")."""
        send_to = list(partners) if send_to is None else list(send_to)
        partners = list(partners)
        if not partners and not send_to:
            return
        m = self._chunk_bits()
        csize = 1 << m
        nchunks = 1 << (self.n_local - m)
        np_ = len(partners)
        stored_real = self._storage_real()
        real = self.real and self.real_transfers and not stored_real
        if self.dry:     # one rank alone: its own chunks stand in for the partners' (same kernels, same bytes counted)
            if after_first_post is not None:
                after_first_post()
            for c in range(nchunks):
                own = self.engine.tensor[c * csize:(c + 1) * csize]
                self.stats["chunk_reads"] += np_
                self.stats["real_chunk_reads"] += np_ if (real or stored_real) else 0
                self.stats["bytes_sent"] += len(send_to) * csize * (8 if (real or stored_real) else 16)
                yield c, [own] * np_
                self.engine.sync()
            self.stats["full_shard_reads"] += np_
            self.stats["partners_per_read"] = max(self.stats["partners_per_read"], np_)
            return
        if np_ and (self._chunk_bufs is None or self._chunk_bufs[0].numel() < np_ * csize or self._chunk_bufs[0].dtype != self.engine.tensor.dtype):
            self._chunk_bufs = None
            self._chunk_bufs = [self.engine.new_buffer(np_ * csize), self.engine.new_buffer(np_ * csize)]
        self.engine.sync()

        if real and np_ and (self._chunk_real is None or self._chunk_real[0].numel() < np_ * csize):
            self._chunk_real = [torch.empty(np_ * csize, dtype=torch.float64, device=self.engine.tensor.device) for _ in range(2)]
        if real and send_to and (self._chunk_send is None or self._chunk_send[0].numel() != csize):   # (sized by the chunk alone: its own test)
            self._chunk_send = [torch.empty(csize, dtype=torch.float64, device=self.engine.tensor.device) for _ in range(2)]

        def post(c):
            # one tag per (chunk parity, partner difference): gloo matches by tag, and both ends of a pair agree on it
            own = self.engine.tensor[c * csize:(c + 1) * csize]
            if real and send_to:   # real parts only over the links; the complex chunks are rebuilt when the transfer has landed
                send = self._chunk_send[c & 1]
                send.copy_(torch.view_as_real(own)[:, 0])
                own = send
            into = self._chunk_real if real else self._chunk_bufs
            items = {}
            for d in send_to:
                items[d] = [own, None]
            for k, d in enumerate(partners):
                items.setdefault(d, [None, None])[1] = into[c & 1][k * csize:(k + 1) * csize]
            return self._post_multi([(sr[0], sr[1], self.rank ^ d, 2 * d + (c & 1)) for d, sr in sorted(items.items())])

        _progress("partner-shard read")
        pending = post(0)
        if after_first_post is not None:   # work that needs no partner data runs while the first chunks are on the links
            after_first_post()
        for c in range(nchunks):
            _progress()
            t0 = time.perf_counter()
            pending.wait()
            if hasattr(self.engine, "stream"):
                self.engine.stream.synchronize()      # the transfers themselves, for the link rates of the bench line
            self.stats["shard_read_s"] += time.perf_counter() - t0
            if real and np_:
                dst = torch.view_as_real(self._chunk_bufs[c & 1][:np_ * csize])
                dst[:, 0].copy_(self._chunk_real[c & 1][:np_ * csize])
                dst[:, 1].zero_()
            pending = post(c + 1) if c + 1 < nchunks else None
            self.stats["chunk_reads"] += np_
            self.stats["real_chunk_reads"] += np_ if (real or stored_real) else 0
            self.stats["bytes_sent"] += len(send_to) * csize * (8 if (real or stored_real) else 16)
            yield c, [self._chunk_bufs[c & 1][k * csize:(k + 1) * csize] for k in range(np_)]
            self.engine.sync()   # the buffers of this parity are posted again two chunks later
        self.stats["full_shard_reads"] += np_
        self.stats["partners_per_read"] = max(self.stats["partners_per_read"], np_)

    def _split_by_chunk(self, terms, xg):
        """terms of one partner group (physical masks) -> {h: (x_low, z_low, coefficient, z_high, x_high&z_high parity, t)} with
        h = the part of the local x mask ABOVE the chunk bits: the term pairs ket chunk c with bra chunk c ^ h"""
        m = self._chunk_bits()
        low = (1 << m) - 1
        lmask = self._local_mask()
        out = {}
        for xp, zp, c, t in terms:
            h = (xp & lmask) >> m
            ny_high = bin((xp & zp) >> m).count("1") & 3
            out.setdefault(h, []).append((xp & low, zp & low, c * (1j) ** ny_high, zp >> m, t))
        return out

    @staticmethod
    def _ket_sign(z_high, ket_high):
        return -1.0 if bin(z_high & ket_high).count("1") & 1 else 1.0

    def _remote_plan(self, groups):
        """[(xg, {h: terms})] for the partner groups + the list of rank differences, in a rank-independent order"""
        remote = [(xg, self._split_by_chunk(terms, xg)) for xg, terms in groups if xg]
        return remote, [xg for xg, _ in remote]

    # -- Hermitian sums: planned once per (term list, permutation) ----------------------------------------------------------------
    @staticmethod
    def share_of(rank, d, world, x_groups):
        """Which x-groups of the cross terms between ``rank`` and ``rank ^ d`` this RANK contracts.  A Hermitian term gives the pair
        (i on rank, j on rank ^ d) and its mirror image conjugate contributions, so ONE of the two ranks evaluates
        2 Re <shard|P|partner shard> and the other neither computes nor receives anything for it: half the contractions AND half the
        xGMI traffic of <H>.  The pairs are oriented like a round-robin tournament — rank r takes the partners s with
        1 <= (s - r) mod W < W/2: (W - 2)/2 each — and the diametric pair (d = W/2) shares its groups alternately.
        -> the sub-list of ``x_groups`` (sorted physical x masks)"""
        s = rank ^ d
        k = (s - rank) % world
        if 2 * k == world:
            return x_groups[0::2] if rank < s else x_groups[1::2]
        return list(x_groups) if 1 <= k < world / 2 else []

    def plan_hamiltonian(self, xs, zs, coeffs, constant=0.0):
        """plan of H = constant + sum_t c_t P_t (real c_t; logical masks) under the CURRENT permutation: the terms this rank
        evaluates for <H> — its d = 0 terms and, doubled, its share of the cross terms — as one engine sum; the partners it reads
        from / sends its shard to; the full list for sigma = H psi (planned at its first use)"""
        m = self._chunk_bits()
        phys = [(self._phys(int(x)), self._phys(int(z)), complex(c)) for x, z, c in zip(xs, zs, coeffs)]
        by_d = {}
        for xp, zp, c in phys:
            by_d.setdefault(xp >> self.n_local, {}).setdefault(xp, []).append((zp, c))
        ex, read_from, send_to = [], [], []
        for d in sorted(by_d):
            groups = sorted(by_d[d])
            if d == 0:
                ex += [(xp, zp, c) for xp in groups for zp, c in by_d[d][xp]]
                continue
            mine = self.share_of(self.rank, d, self.world, groups)
            ex += [(xp, zp, 2.0 * c) for xp in mine for zp, c in by_d[d][xp]]
            if mine:
                read_from.append(d)
            if self.share_of(self.rank ^ d, d, self.world, groups):
                send_to.append(d)
        # (the engine sums are made at their first use: an ADAPT screen never needs "expect", an energy never "apply")
        return {"m": m, "perm": tuple(self.perm), "const": float(np.real(constant)), "terms": phys, "expect_terms": ex,
                "read_from": read_from, "send_to": send_to, "partners": sorted(d for d in by_d if d), "expect": None, "apply": None}

    def _plan_sum(self, plan, which):
        if plan[which] is None:
            t = plan["expect_terms" if which == "expect" else "terms"]
            plan[which] = self.engine.plan_sum(np.array([v[0] for v in t], np.uint64), np.array([v[1] for v in t], np.uint64),
                                               np.array([v[2] for v in t], np.complex128), plan["m"])
        return plan[which]

    def free_plan(self, plan):
        for key in ("expect", "apply"):
            if plan.get(key) is not None:
                self.engine.free_sum(plan[key])
                plan[key] = None

    def _plan_for(self, xs, zs, coeffs, constant):
        """ad-hoc calls: the plans of the last few (term list, permutation) pairs are kept"""
        key = (tuple(self.perm), self._chunk_bits(), np.asarray(xs, np.uint64).tobytes(), np.asarray(zs, np.uint64).tobytes(),
               np.asarray(coeffs, np.complex128).tobytes())
        cache = self.__dict__.setdefault("_plans", {})
        plan = cache.pop(key, None)
        if plan is None:
            plan = self.plan_hamiltonian(xs, zs, coeffs, 0.0)
            while len(cache) >= 4:
                self.free_plan(cache.pop(next(iter(cache))))
        cache[key] = plan        # (most recently used last)
        plan["const"] = float(np.real(constant))
        return plan

    def expectation(self, xs, zs, coeffs, constant=0.0):
        """Re sum_t c_t <psi|P_t|psi> + constant over the whole register (same value on every rank)"""
        return self._expectation_planned(self._plan_for(xs, zs, coeffs, constant))

    def _expectation_planned(self, plan):
        if plan["perm"] != tuple(self.perm):
            raise ValueError("the Hamiltonian was planned under another qubit permutation")
        sid = self._plan_sum(plan, "expect")
        local = []

        def local_part():     # (posted first: the first chunk of every partner travels while the shard's own terms are evaluated)
            with self._compute("expectation_local"):
                local.append(self.engine.sum_expect_local(sid))

        for c, chunks in self._partner_chunks(plan["read_from"], plan["send_to"], after_first_post=local_part):
            with self._compute("expectation_remote"):
                for d, ket in zip(plan["read_from"], chunks):
                    self.engine.sum_expect_remote(sid, d, c, ket)
        if not local:
            local_part()      # (no partner at all: nothing was posted)
        total = local[0]
        if plan["read_from"]:
            total += self.engine.sum_expect_finish(sid).real
        val = torch.tensor([total], dtype=torch.float64, device=self.engine.tensor.device)
        _progress("all-reduce of the energy")
        if self._dist:
            dist.all_reduce(val, group=self.group)
        out = float(val.item()) + plan["const"]
        _progress("local sweeps")
        return out

    # -- ADAPT gradient screen on the sharded register ("ADAPT screen identical with sigma also sharded")
    def apply_hamiltonian(self, xs, zs, coeffs, constant=0.0, plan=None):
        """sigma = (H + constant) psi, sharded like psi: the x_g = 0 terms act inside the shard; the other rank differences
        are accumulated chunk by chunk from the partners' psi shards (all partners of a chunk in flight at once).
        -> this rank's sigma shard (device buffer owned by the caller until the next call)"""
        size = 1 << self.n_local
        self._complex_storage()       # (sigma = H psi and the screens built on it work on complex amplitudes)
        if getattr(self, "_sigma", None) is None or self._sigma.numel() < size:
            self._sigma = self.engine.new_buffer(size)
        sigma = self._sigma[:size]
        plan = plan if plan is not None else self._plan_for(xs, zs, coeffs, constant)
        if plan["perm"] != tuple(self.perm):
            raise ValueError("the Hamiltonian was planned under another qubit permutation")
        sid = self._plan_sum(plan, "apply")
        with self._compute("apply"):
            self.engine.sum_apply_local(sid, sigma, float(np.real(constant)))
        for c, chunks in self._partner_chunks(plan["partners"]):
            with self._compute("apply"):
                for d, ket in zip(plan["partners"], chunks):
                    self.engine.sum_apply_remote(sid, d, c, ket, sigma)
        self.engine.sync()
        return sigma

    def pool_gradients(self, ham, pool, mode="fermionic"):
        """ADAPT screen over ``pool`` = [(xs, zs, coeffs) per operator] with H = (ham_xs, ham_zs, ham_coeffs, constant):
        sigma = H psi once (sharded), then v_k = sum_j c_j <sigma|P_j|psi> with the pool terms grouped by partner shard
        — one batched launch per (rank difference, chunk pair) —, one all-reduce of the n_ops complex values;
        g_k = 2 Re v_k (fermionic, ref:openvqe/adapt/fermionic_adapt_vqe.py:67-73) or 2 |v_k| (qubit,
        ref:openvqe/adapt/qubit_adapt_vqe.py:147-150).  Same values on every rank."""
        hx, hz, hc, const = ham
        sigma = self.apply_hamiltonian(hx, hz, hc, const)
        flat = [(int(x), int(z), complex(c), k) for k, (xs, zs, cs) in enumerate(pool) for x, z, c in zip(xs, zs, cs)]
        groups = self._group_by_partner([f[0] for f in flat], [f[1] for f in flat], [f[2] for f in flat])
        owner = [f[3] for f in flat]
        vals = np.zeros(len(pool), np.complex128)

        def csr(terms, coeff_of):
            # CSR over the operators that have terms in this group (operator order kept)
            by_op = {}
            for t in terms:
                by_op.setdefault(owner[t[-1]], []).append(t)
            ops = sorted(by_op)
            offsets = np.zeros(len(ops) + 1, np.int64)
            xs, zs, cs = [], [], []
            for i, k in enumerate(ops):
                offsets[i + 1] = offsets[i] + len(by_op[k])
                for t in by_op[k]:
                    xs.append(t[0]); zs.append(t[1]); cs.append(coeff_of(t))
            return ops, offsets, np.array(xs, np.uint64), np.array(zs, np.uint64), np.array(cs, np.complex128)

        for xg, terms in groups:
            if xg == 0:
                ops, offsets, txs, tzs, tcs = csr(terms, lambda t: t[2])
                vals[ops] += self.engine.bilinear_batch(offsets, txs, tzs, tcs, sigma, None)
        remote, partners = self._remote_plan(groups)
        m = self._chunk_bits()
        for c, chunks in self._partner_chunks(partners):
            for (xg, by_h), ket in zip(remote, chunks):
                ket_high = ((self.rank ^ xg) << (self.n_local - m)) | c
                for h, terms in sorted(by_h.items()):
                    ops, offsets, txs, tzs, tcs = csr(terms, lambda t: t[2] * self._ket_sign(t[3], ket_high))
                    vals[ops] += self.engine.sub_bilinear_batch(m, offsets, txs, tzs, tcs, sigma, (c ^ h) << m, ket)
        buf = torch.from_numpy(np.stack([vals.real, vals.imag])).to(self.engine.tensor.device)
        if self._dist:
            dist.all_reduce(buf, group=self.group)
        v = buf[0].cpu().numpy() + 1j * buf[1].cpu().numpy()
        return 2.0 * v.real if mode == "fermionic" else 2.0 * np.abs(v)

    # -- compiled programs: the exchange plan of a rotation list is made ONCE (ref:openvqe/ucc_family/get_energy_ucc.py:42-45 runs the
    # same list with new angles at every optimiser step) -----------------------------------------------------------------------
    def compile_program(self, rot_xs, rot_zs, rot_coeffs, rot_pidx, hf_index, hamiltonian=None, rot_phi0=None):
        """plan of `|hf> -> prod_r exp(-i coeff_r theta[pidx_r] P_r)` on this partition, from the identity permutation: the
        half-shard exchanges (Belady victims over the whole list, next uses by bisection: O(R n log R) instead of the O(R^2 n) scan)
        and, between them, the local rotations with their masks already in PHYSICAL bit space; with ``hamiltonian`` =
        (xs, zs, coeffs, constant) also its terms grouped by partner shard under the program's FINAL permutation.
        -> an object for ``run_program`` / ``program_energy``; identical on every rank"""
        xs = [int(v) for v in rot_xs]
        zs = [int(v) for v in rot_zs]
        saved = list(self.perm)
        self.perm = list(range(self.n))
        uses = self._use_lists(xs, self.n)
        steps, batch = [], []

        def record_swap(gbit, lbit):
            la, lb = self.perm.index(gbit), self.perm.index(lbit)
            self.perm[la], self.perm[lb] = lbit, gbit
            steps.append(("swap", gbit, lbit))

        def flush():
            if batch:
                idx = np.array([b[0] for b in batch], np.int64)
                steps.append(("rot", np.array([b[1] for b in batch], np.uint64), np.array([b[2] for b in batch], np.uint64), idx))
                batch.clear()

        try:
            for r in range(len(xs)):
                if self._phys(xs[r]) & ~self._local_mask():
                    flush()
                    xp = self._localise(xs, r, uses, swap=record_swap)
                else:
                    xp = self._phys(xs[r])
                batch.append((r, xp, self._phys(zs[r])))
            flush()
            final_perm = list(self.perm)
            prog = {"steps": steps, "coeff": np.asarray(rot_coeffs, np.float64), "pidx": np.asarray(rot_pidx, np.int64),
                    "phi0": None if rot_phi0 is None else np.asarray(rot_phi0, np.float64),
                    "hf": int(hf_index), "perm": final_perm, "real": all(bin(x & z).count("1") & 1 for x, z in zip(xs, zs)),
                    "swaps": sum(1 for st in steps if st[0] == "swap"), "ham": None,
                    "n_params": int(np.asarray(rot_pidx, np.int64).max(initial=-1)) + 1}
            if hamiltonian is not None:
                hx, hz, hc, const = hamiltonian
                prog["ham"] = self.plan_hamiltonian(hx, hz, hc, const)   # under the final permutation (self.perm right now)
        finally:
            self.perm = saved
        return prog

    def run_program(self, prog, theta):
        """|hf> -> the program's state at ``theta`` (the plan's exchanges and local sweeps; no planning)"""
        theta = np.asarray(theta, np.float64).reshape(-1)
        if theta.size < prog["n_params"]:
            raise ValueError(f"expected {prog['n_params']} parameters, got {theta.size}")
        self.perm = list(range(self.n))
        self._choose_storage(prog["real"])
        self.init_basis(prog["hf"])
        self.real = prog["real"]      # (a list with one even-Y string anywhere travels complex from the start: the flag is per program)
        pidx = prog["pidx"]
        # rotation r: exp(-i (coeff_r theta[pidx_r] + phi0_r) P_r); pidx_r < 0: a constant angle — phi0_r when the program carries
        # constants (ovqe_set_program's convention), else coeff_r
        fixed = 1.0 if prog.get("phi0") is None else 0.0
        phis = prog["coeff"] * (np.where(pidx >= 0, theta[np.maximum(pidx, 0)], fixed) if theta.size else fixed)
        if prog.get("phi0") is not None:
            phis = phis + prog["phi0"]
        for st in prog["steps"]:
            if st[0] == "swap":
                self._swap(st[1], st[2])
            else:
                with self._compute("local_sweeps"):
                    self.engine.rotations(st[1], st[2], phis[st[3]])
        assert self.perm == prog["perm"]

    def program_energy(self, prog, theta):
        """E(theta) = <psi(theta)|H|psi(theta)> + constant with the Hamiltonian the program was compiled with"""
        if prog["ham"] is None:
            raise ValueError("compile_program was called without a Hamiltonian")
        self.run_program(prog, theta)
        return self._expectation_planned(prog["ham"])

    def energy(self, ham_xs, ham_zs, ham_coeffs, constant, rot_xs, rot_zs, rot_phis, hf_index):
        """one whole evaluation: |hf> -> rotations -> <H>"""
        self.perm = list(range(self.n))
        self._choose_storage(all(bin(int(x) & int(z)).count("1") & 1 for x, z in zip(rot_xs, rot_zs)) and len(rot_xs) > 0)
        self.init_basis(hf_index)
        self.apply_pauli_rotations(rot_xs, rot_zs, rot_phis)
        return self.expectation(ham_xs, ham_zs, ham_coeffs, constant)

    # -- read-back (tests / small registers) ------------------------------------------------------
    def gather_state(self):
        """full state in LOGICAL index order on every rank (small n only)"""
        if self._storage_real():
            self._complex_storage()
        shard = self.engine.tensor.detach().to("cpu")
        if self.world > 1:
            parts = [torch.empty_like(shard) for _ in range(self.world)]
            if self.engine.tensor.is_cuda:
                dev_parts = [torch.empty_like(self.engine.tensor) for _ in range(self.world)]
                dist.all_gather(dev_parts, self.engine.tensor, group=self.group)
                parts = [p.cpu() for p in dev_parts]
            else:
                dist.all_gather(parts, shard, group=self.group)
            phys = torch.cat(parts).numpy()
        else:
            phys = shard.numpy()
        idx = np.arange(1 << self.n, dtype=np.int64)
        pidx = np.zeros_like(idx)
        for b in range(self.n):
            pidx |= ((idx >> b) & 1) << self.perm[b]
        return phys[pidx]
