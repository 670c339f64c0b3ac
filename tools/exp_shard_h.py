"""configs[4]'s one-shard leg (bench.sharded_leg at world size 1) at n qubits with options from OVQE_OPTIONS (testing build:
OVQE_LIB=testing OVQE_OPTIONS=tile_flat=0 python tools/exp_shard_h.py 29)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29
leg = bench.sharded_leg(n, 0, 1, 0)
print({k: leg[k] for k in ("n_qubits", "rotations_s", "expectation_s", "expectation_passes_executed", "expectation_GBs_per_gpu", "energy", "norm2")}, flush=True)
