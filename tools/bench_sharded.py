#!/usr/bin/env python3
"""configs[4] alone: the `sharded` block of bench.py without the replica benchmark around it.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_sharded.py \
      --qubits-per-gpu 31          # weak scaling: n = 31 + log2(N)  (32 GiB shard per GPU)
  ... --qubits 31                 # strong scaling: fixed n

Prints one JSON line on rank 0 (`bench.sharded_leg`: sweep time, exchanged bytes, per-link GB/s, energy)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
    import bench
    out = bench.sharded_leg(n, local_rank, world, rank, args.rotations, args.terms, dist.barrier if world > 1 else None)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
