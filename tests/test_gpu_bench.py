"""bench.py's N > 1 path on one GPU: two processes (torch.distributed over gloo, both ranks on device 0 — RCCL refuses two
ranks on one device) run the replica benchmark; rank 0 must print ONE well-formed JSON line with the contract's keys and
the whole-job aggregate over both ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_process_replica_bench_line(gpu_lib):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OVQE_BENCH_BACKEND="gloo", OVQE_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "2048", "--no-roofline", "--no-cpu", "--no-extra"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "evals/s" and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert "workload" in out["config"] and out["config"]["parallelism"].endswith("x2")
    # whole-job aggregate: both ranks' batches over the max-over-ranks time
    assert abs(out["value"] - 2 * 2048 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_two_process_sharded_block_against_oracle(gpu_lib):
    """configs[4] through the plain command `python bench.py --gpus 2` at a size the oracle holds (gloo, both ranks on
    device 0): the parent starts the two ranks itself, and the line must carry the `sharded` block — weak run at n = 16 (one global qubit), strong run at n = 15 — and both energies
    must equal the bit-mask oracle's on the host-recomputed synthetic state."""
    import numpy as np

    import bench
    from openvqe_amd import synth
    from oracle import masks
    # the PLAIN command, no launcher and no WORLD_SIZE: bench.py starts its own two ranks (a fresh child, never exec)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OVQE_BENCH_BACKEND="gloo", OVQE_BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "1024", "--no-roofline", "--no-cpu", "--no-extra", "--sharded-qubits", "15", "--sharded-rotations", "24",
           "--sharded-terms", "80"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])     # the JSON line is the LAST line
    assert out["n_gpus"] == 2 and out["config"]["parallelism"].endswith("x2")
    sh = out["sharded"]
    assert sh["weak"]["n_qubits"] == 16 and sh["strong"]["n_qubits"] == 15
    for leg in (sh["weak"], sh["strong"]):
        n = leg["n_qubits"]
        xs, zs, phis, hx, hz, hc = bench.sharded_workload(n, 24, 80)
        psi = synth.amplitudes(bench.SHARDED_SEED, np.arange(1 << n, dtype=np.uint64))
        psi = psi / np.linalg.norm(psi)
        for x, z, p in zip(xs, zs, phis):
            psi = masks.rotate(psi, x, z, p)
        want = masks.expectation(psi, hx, hz, hc, 0.0)
        assert abs(leg["energy"] - want) < 1e-11 * np.abs(hc).sum(), (n, leg["energy"], want)
        assert abs(leg["norm2"] - 1.0) < 1e-12
        assert leg["n_gpus"] == 2 and leg["swaps"] >= 1 and leg["exchanged_GiB_per_rank"] > 0
        assert leg["xgmi_link_GBs_exchange"] > 0 and leg["full_shard_reads"] >= 1
