#!/usr/bin/env python3
"""configs[4]: synthetic random JW Hamiltonian + Pauli rotations on a statevector sharded over the
GPUs of one node (index-bit partition, half-shard exchange over RCCL).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py \
      --qubits-per-gpu 31          # weak scaling: n = 31 + log2(N)  (32 GiB shard per GPU)
  ... --qubits 31                 # strong scaling: fixed n

Prints one JSON line on rank 0: sweep time, exchanged bytes, aggregate GB/s of the local sweeps, energy."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def two_body_string(rng, n):
    """(x, z) of a JW double-excitation-like string: X/Y on 4 random qubits, Z chains between the pairs"""
    q = sorted(rng.choice(n, 4, replace=False).tolist())
    x = sum(1 << (n - 1 - k) for k in q)
    z = 0
    for lo, hi in ((q[0], q[1]), (q[2], q[3])):
        for k in range(lo + 1, hi):
            z |= 1 << (n - 1 - k)
    ys = rng.choice(4, int(rng.choice([1, 3])), replace=False)
    for k in ys:
        z |= 1 << (n - 1 - q[k])
    return x, z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--qubits-per-gpu", type=int, default=None)
    ap.add_argument("--qubits", type=int, default=None)
    ap.add_argument("--rotations", type=int, default=64)
    ap.add_argument("--terms", type=int, default=1000)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    g = world.bit_length() - 1
    n = args.qubits if args.qubits else (args.qubits_per_gpu or 31) + g
    from openvqe_amd.distributed import ShardedStatevector
    rng = np.random.default_rng(34)
    rots = [two_body_string(rng, n) for _ in range(args.rotations)]
    xs, zs = [r[0] for r in rots], [r[1] for r in rots]
    phis = rng.uniform(-0.2, 0.2, args.rotations)
    hx, hz = [], []
    for _ in range(args.terms):
        x, z = two_body_string(rng, n)
        if rng.random() < 0.3:
            x = 0  # diagonal term
        else:
            z ^= x & z if rng.random() < 0.5 else 0
            if bin(x & z).count("1") & 1:   # keep H real-symmetric: even number of Y
                z ^= x & -x
        hx.append(x); hz.append(z)
    hc = rng.normal(size=args.terms)
    sv = ShardedStatevector(n, device=local_rank)
    sv.randomize(20250227)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    sv.apply_pauli_rotations(xs, zs, phis)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_rot = time.perf_counter() - t0
    swaps_rot, bytes_rot = sv.stats["swaps"], sv.stats["bytes_sent"]
    t0 = time.perf_counter()
    e = sv.expectation(hx, hz, hc, 0.0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_exp = time.perf_counter() - t0
    n2 = sv.norm2()
    if rank == 0:
        groups = len(set(hx))
        print(json.dumps({
            "workload": f"{args.rotations} JW two-body rotations + {args.terms}-term random JW Hamiltonian ({groups} x-groups)",
            "n_qubits": n, "n_gpus": world, "shard_GiB": 16 * 2 ** (n - g) / 2 ** 30,
            "rotations_s": t_rot, "swaps": swaps_rot, "exchanged_GiB_per_rank": bytes_rot / 2 ** 30,
            "local_sweep_GBs_aggregate": 32.0 * 2 ** n * args.rotations / t_rot / 1e9,
            "expectation_s": t_exp, "full_shard_reads": sv.stats["full_shard_reads"],
            "expectation_GBs_aggregate": 16.0 * 2 ** n * groups / t_exp / 1e9,
            "energy": e, "norm2": n2}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
