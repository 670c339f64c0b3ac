"""The RCCL transport (torch.distributed backend "nccl") on the one GPU of the pool, i.e. at world size 1: process-group
init on the device, barrier, device-tensor all_reduce, device-tensor batch_isend_irecv (loop-back, pieces of an exchange),
the sharded register under that group at 28 qubits against the oracle formula on sampled amplitudes, and bench.py's
distributed code path (OVQE_BENCH_FORCE_DIST) with the JSON line last.  Every child is spawned before it touches the
GPU and runs under a timeout.  The world-size-2/4 logic is covered over gloo (tests/test_distributed.py,
tests/test_gpu_distributed.py); RCCL refuses two ranks on one device."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from openvqe_amd import synth
from tests.test_distributed import _free_port
from tests.test_gpu_fullsize import _host_rotate

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20250227


def _nccl_worker(port, n, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from openvqe_amd.distributed import ShardedStatevector
        from openvqe_amd.operators import pack_string
        res = {}
        dist.barrier()
        t = torch.arange(8, dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        res["all_reduce"] = t.cpu().numpy()
        sv = ShardedStatevector(n, device=0)
        assert dist.get_backend() == "nccl" and sv.world == 1 and sv._dist
        # device-tensor point-to-point through the product's own posting code: a loop-back pair, then the piece
        # pattern of an exchange (unequal pieces, one per tag)
        a = torch.randn(1 << 16, dtype=torch.float64, device="cuda").to(torch.complex128)
        b = torch.zeros_like(a)
        sv._post_pair(a, b, 0).wait()
        torch.cuda.synchronize()
        res["loopback"] = bool(torch.equal(a, b))
        cuts = [0, 1000, 1001, 30000, 1 << 16]
        c = torch.zeros_like(a)
        got = []
        sv._exchange(0, lambda p: a[cuts[p]:cuts[p + 1]], [c[cuts[p]:cuts[p + 1]] for p in range(4)], got.append)
        torch.cuda.synchronize()
        res["pieces"] = bool(torch.equal(a, c)) and got == [0, 1, 2, 3]
        # the sharded register under the RCCL group: norm (device all_reduce), rotations, <H>
        n2_raw = sv.randomize(SEED)
        scale = 1.0 / np.sqrt(n2_raw)
        res["norm2"] = sv.norm2()
        rng = np.random.default_rng(n)
        idx = rng.integers(0, 1 << n, 2048).astype(np.uint64)
        idx[:4] = [0, 1, (1 << n) - 1, 1 << (n - 1)]
        res["base"] = sv.engine.sv.get_amplitudes(idx)
        checks = []
        for op, qs in (("XXXY", [0, 1, 2, 3]), ("YXXX", [0, 9, 19, n - 1]), ("ZXZY", [3, 14, 15, 27]), ("ZZ", [0, n - 1])):
            x, z = pack_string(n, op, qs)
            phi = float(rng.uniform(0.1, 1.0))
            sv.apply_pauli_rotations([x], [z], [phi])
            checks.append((x, z, phi, sv.engine.sv.get_amplitudes(idx)))
            sv.apply_pauli_rotations([x], [z], [-phi])
        res["checks"] = checks
        xs = [0] + [pack_string(n, o, q)[0] for o, q in (("XXXY", [0, 1, 2, 3]), ("ZZ", [0, n - 1]))]
        zs = [0] + [pack_string(n, o, q)[1] for o, q in (("XXXY", [0, 1, 2, 3]), ("ZZ", [0, n - 1]))]
        res["ident"] = sv.expectation(xs[:1], zs[:1], [1.0])
        res["singles"] = [sv.expectation(xs[k:k + 1], zs[k:k + 1], [1.0]) for k in (1, 2)]
        res["combo"] = sv.expectation(xs, zs, [0.5, 0.3, -1.7], 0.25)
        res["idx"], res["scale"] = idx, scale
        out.put(res)
    finally:
        dist.destroy_process_group()


def test_rccl_world_size_one_sharded_register(gpu_lib):
    n = 28
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), n, out))
    p.start()
    try:
        res = out.get(timeout=600)
    finally:
        p.join(timeout=120)
        if p.is_alive():
            p.terminate()
    assert p.exitcode == 0
    assert np.array_equal(res["all_reduce"], np.arange(8.0))
    assert res["loopback"] and res["pieces"]
    assert abs(res["norm2"] - 1.0) < 1e-10
    idx, scale = res["idx"], res["scale"]
    base = synth.amplitudes(SEED, idx) * scale
    assert np.abs(res["base"] - base).max() < 4 * np.finfo(float).eps * np.abs(base).max()
    for x, z, phi, got in res["checks"]:
        want = _host_rotate(SEED, scale, idx, x, z, phi)
        assert np.abs(got - want).max() < 8 * np.finfo(float).eps * np.abs(want).max(), (x, z)
    assert abs(res["ident"] - 1.0) < 1e-10
    assert abs(res["combo"] - (0.25 + 0.5 + 0.3 * res["singles"][0] - 1.7 * res["singles"][1])) < 1e-10


def test_bench_distributed_path_over_rccl_at_world_size_one(gpu_lib):
    """bench.py --gpus 1 with the process group forced on: RCCL init on the device, barriers, the CUDA-tensor MAX
    all_reduce of the wall time, the `sharded` block through the same group — and the JSON line is the LAST line."""
    env = dict(os.environ, OVQE_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4096",
           "--no-roofline", "--no-cpu", "--no-extra", "--sharded-qubits", "24", "--sharded-rotations", "16",
           "--sharded-terms", "60"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    out = json.loads(last)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert out["sharded"]["weak_qubits"] == 24
    sh = json.load(open(os.path.join(ROOT, out["extra"])))["sharded"]     # the legs' detail is in bench_extra.json
    assert sh["weak"]["n_qubits"] == 24 and sh["weak"]["swaps"] == 0 and abs(sh["weak"]["norm2"] - 1.0) < 1e-10
    assert sh["strong"]["energy"] == sh["weak"]["energy"]


def _nccl_api_worker(port, out):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OVQE_PARTITION_MIN_QUBITS="6",
                      OVQE_PARTITION_FORCE="1", OVQE_SHARD_CHUNK_BITS="5")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import openvqe_amd.evaluator as ev
        from openvqe_amd.partitioned import PartitionedStatevector
        from tests.test_partitioned_api import FLOWS, run_flows
        res = run_flows(FLOWS)
        assert ev._BACKENDS and all(isinstance(sv, PartitionedStatevector) for sv in ev._BACKENDS.values())
        out.put(res)
    finally:
        dist.destroy_process_group()


def test_reference_entry_points_on_the_partitioned_register_over_rccl(gpu_lib):
    """the partitioned register behind `ucc_action` / both ADAPT mirrors under an RCCL process group (world size 1 — RCCL refuses two
    ranks on one device —: the device-tensor all-reduces of energies, norms and pool gradients go through RCCL): same results as the
    single-process oracle engine"""
    from tests.test_partitioned_api import FLOWS, _single_process_oracle, check_partitioned
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_nccl_api_worker, args=(_free_port(), out))
    p.start()
    got = out.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    check_partitioned(got, _single_process_oracle(FLOWS))
