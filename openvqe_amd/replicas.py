"""The GPUs of one node behind the reference's entry points, chosen per register:

  * **partitioned register** — ``partitioned.PartitionedStatevector`` (index-bit partition, ``distributed.ShardedStatevector``) when
    the register has at least ``partition_min_qubits()`` qubits: no single device holds it (34 qubits = 256 GiB), every rank works on
    the same evaluation;
  * **replicas over the batch** — below that size every rank holds the whole register (all molecule configs: n <= 24) and the ranks
    share the BATCH of a call: the K + 1 parameter vectors of a forward-difference gradient (``batched_gradient`` of the UCC
    mirrors, ref:openvqe/ucc_family/get_energy_ucc.py:158-175) or the operators of an ADAPT pool
    (ref:openvqe/adapt/fermionic_adapt_vqe.py:77-122); one all-gather of the scalars, no other collective.

One process per GPU (``python -m torch.distributed.run --nproc-per-node N script.py``): every rank runs the SAME script with the same
inputs — the optimiser loops of the reference stay as they are and see identical numbers on every rank.  Nothing here imports torch
unless a process group exists: a single-GPU caller never pays for it.
"""
from __future__ import annotations

import os
import sys

import numpy as np


def _dist():
    if "torch" not in sys.modules:       # a process group cannot exist without torch having been imported
        return None
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    d = _dist()
    return d.get_world_size() if d is not None else 1


def rank():
    d = _dist()
    return d.get_rank() if d is not None else 0


def device():
    """the GPU of this rank: LOCAL_RANK under a launcher; 0 for a single process and when OVQE_SINGLE_DEVICE is set (several ranks on
    one GPU: the gloo runs of the tests)"""
    if world() == 1 or os.environ.get("OVQE_SINGLE_DEVICE"):
        return 0
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend="nccl"):
    """what a launcher script calls first: one rank per GPU over RCCL ("nccl" IS RCCL on ROCm; "gloo" for the one-GPU tests)"""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if backend == "nccl":
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)


def partition_min_qubits():
    """registers from this size on are partitioned across the ranks (OVQE_PARTITION_MIN_QUBITS; default 31: a 30-qubit state with its
    scratch buffers still fits one 288-GB device comfortably and needs no exchange at all)"""
    return int(os.environ.get("OVQE_PARTITION_MIN_QUBITS", "31"))


def partitioned(nbqbits):
    """OVQE_PARTITION_FORCE=1: also under a one-rank process group (one shard = the register: the collectives of the partitioned
    path over RCCL on a single GPU, tests/test_gpu_nccl.py)"""
    w = world()
    if w == 1 and not (os.environ.get("OVQE_PARTITION_FORCE") and _dist() is not None):
        return False
    return (w & (w - 1)) == 0 and int(nbqbits) >= partition_min_qubits()


def active(nbqbits):
    """replicas over the batch: several ranks, register not partitioned"""
    return world() > 1 and not partitioned(nbqbits)


def my_slice(count):
    """the contiguous share of ``count`` batch items this rank evaluates -> (start, stop)"""
    w, r = world(), rank()
    base, extra = divmod(int(count), w)
    start = r * base + min(r, extra)
    return start, start + base + (1 if r < extra else 0)


def gather(values, count):
    """every rank's share of a length-``count`` float64 result -> the whole array on every rank (one all-gather of padded rows)"""
    import torch
    d = _dist()
    w = world()
    width = (int(count) + w - 1) // w
    dev = "cuda:%d" % device() if d.get_backend() == "nccl" else "cpu"
    mine = torch.zeros(max(width, 1), dtype=torch.float64, device=dev)
    vals = np.ascontiguousarray(values, np.float64)
    if vals.size:
        mine[: vals.size] = torch.from_numpy(vals).to(dev)
    rows = [torch.empty_like(mine) for _ in range(w)]
    d.all_gather(rows, mine)
    out = np.empty(int(count), np.float64)
    base, extra = divmod(int(count), w)
    at = 0
    for r in range(w):
        n = base + (1 if r < extra else 0)
        out[at:at + n] = rows[r][:n].cpu().numpy()
        at += n
    return out


def energy_batch(sv, thetas):
    """``sv.energy_batch`` with the rows of ``thetas`` shared between the ranks"""
    thetas = np.ascontiguousarray(thetas, np.float64)
    a, b = my_slice(thetas.shape[0])
    part = sv.energy_batch(thetas[a:b]) if b > a else np.zeros(0)
    return gather(part, thetas.shape[0])


_pool_slices = {}


def pool_gradients(sv, pool_ops, mode):
    """``sv.pool_gradients`` with the operators of the pool shared between the ranks (every rank holds the same screen state)"""
    a, b = my_slice(len(pool_ops))
    key = (id(pool_ops), len(pool_ops), a, b)
    mine = _pool_slices.get(key)
    if mine is None or mine[0] is not pool_ops:
        _pool_slices.clear()                      # (the slice object is kept: the backend caches the packed pool by identity)
        mine = _pool_slices[key] = (pool_ops, list(pool_ops[a:b]))
    part = sv.pool_gradients(mine[1], mode) if b > a else np.zeros(0)
    return gather(part, len(pool_ops))
