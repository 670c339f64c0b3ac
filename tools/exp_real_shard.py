"""31-qubit REAL-state shard (float64 amplitudes): rotations and <H> of the bench's sharded workload under variants of the tiled <H>
(testing library: OVQE_LIB=testing).  Usage: OVQE_LIB=testing python tools/exp_real_shard.py [qubits]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
for label, opts in (("default", ""), ("entries_not_items", "tile_flat=0"), ("no_dense_census", "expect_dense=0"),
                    ("entries_no_census", "tile_flat=0,expect_dense=0"), ("one_stream", "expect_streams=1")):
    os.environ["OVQE_OPTIONS"] = opts
    leg = bench.sharded_leg(n, 0, 1, 0, real_state=True)
    print(json.dumps({"variant": label, "options": opts, "rotations_s": leg["rotations_s"], "expectation_s": leg["expectation_s"],
                      "passes": leg["expectation_passes_executed"], "sweeps": leg["local_sweeps_executed"], "energy": leg["energy"],
                      "real_storage": leg.get("real_storage_through_the_leg")}), flush=True)
os.environ["OVQE_OPTIONS"] = "expect_streams=1"
leg = bench.sharded_leg(n, 0, 1, 0, real_state=False)
print(json.dumps({"variant": "complex_state_one_stream", "rotations_s": leg["rotations_s"], "expectation_s": leg["expectation_s"],
                  "passes": leg["expectation_passes_executed"]}), flush=True)
os.environ["OVQE_OPTIONS"] = ""
leg = bench.sharded_leg(n, 0, 1, 0, real_state=False)
print(json.dumps({"variant": "complex_state", "rotations_s": leg["rotations_s"], "expectation_s": leg["expectation_s"],
                  "passes": leg["expectation_passes_executed"], "sweeps": leg["local_sweeps_executed"], "energy": leg["energy"]}), flush=True)
