"""The sharded state on its PRODUCT engine at world size 2 and 4: one process per rank, every rank's shard handle on
device 0 (the pool gives one GPU; RCCL refuses two ranks on one device, so the exchange runs over gloo, which moves the
device tensors point to point).  Same workloads and checks as tests/test_distributed.py, where the shard arithmetic is the
bit-mask oracle: here the HIP kernels do it — rotations with local x masks, pipelined half-shard exchanges for global
ones, <H> with global-x groups against partner shards, sigma = H psi and the ADAPT pool contraction per partner shard."""
import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import masks
from tests.test_distributed import _free_port, _screen_worker, _worker

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 14, None), (4, 15, 10), (2, 17, 12), (8, 16, 10), (2, 12, 8), (4, 16, 11)])
def test_sharded_state_on_hip_shards(gpu_lib, world, n, chunk_bits):
    """(8, 16, 10): eight HIP shards of 13 local qubits on the one GPU — x on two and three rank bits, the partner groups of <H> read
    in 8 chunks of 2^10 amplitudes each and contracted by the LDS-tiled cross-shard passes (k_tile_cross: tiles of 2^10, 2^11 and
    2^12 amplitudes over the cases; (2, 12, 8): chunks below the tile sizes, k_cross_small)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 4321 + n, out, "hip", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    e, full, n2, stats, (xs, zs, phis, hx, hz, hc, hf) = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    assert np.abs(np.asarray(full) - psi).max() < 1e-12
    assert abs(n2 - 1.0) < 1e-12
    assert abs(e - masks.expectation(psi, hx, hz, hc, 0.25)) < 1e-11
    g = world.bit_length() - 1
    assert 1 <= stats["swaps"] <= (1 if world < 8 else g) * sum(1 for x in xs if x >> (n - g)) and stats["full_shard_reads"] >= 1
    if world == 8:   # Hermitian halving: rank 0 contracts four of its seven partner shards (ShardedStatevector.share_of)
        assert stats["partners_per_read"] == 4 and stats["chunk_reads"] == 4 * (1 << (n - 3 - chunk_bits))


@pytest.mark.parametrize("world,n,chunk_bits", [(2, 13, 9), (4, 14, None), (8, 15, 9)])
def test_sharded_adapt_screen_on_hip_shards(gpu_lib, world, n, chunk_bits):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_screen_worker, args=(r, world, port, n, 77 + n, out, "hip", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    gf, gq, stats, (xs, zs, phis, hx, hz, hc, pool, hf) = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, x, z, p)
    sigma = 0.3 * psi
    for x, z, c in zip(hx, hz, hc):
        sigma = sigma + c * masks.pauli_apply(psi, int(x), int(z))
    want = np.array([sum(c * np.vdot(sigma, masks.pauli_apply(psi, int(x), int(z))) for x, z, c in zip(*op)) for op in pool])
    assert np.abs(np.asarray(gf) - 2.0 * want.real).max() < 1e-11
    assert np.abs(np.asarray(gq) - 2.0 * np.abs(want)).max() < 1e-11
    assert stats["full_shard_reads"] >= 2


@pytest.mark.parametrize("world,n,chunk_bits", [(4, 15, 10), (8, 16, 10), (2, 17, 13), (2, 15, 12)])
def test_real_amplitude_transfers_on_hip_shards(gpu_lib, world, n, chunk_bits):
    """odd-Y rotations from a basis state, three ways: float64 shards (real-amplitude kernels: tiles of 2^13 / 2^12 / 2^11 doubles and
    the streaming fall-back over the cases), complex shards with real parts only on the wire, complex shards and complex wire —
    state and <H> against the oracle"""
    from tests.test_distributed import _real_worker
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, world, port, n, 909 + n, out, "hip", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    res, (xs, zs, phis, hx, hz, hc, hf) = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    psi = np.zeros(1 << n, complex)
    psi[hf] = 1
    for x, z, p in zip(xs, zs, phis):
        psi = masks.rotate(psi, int(x), int(z), p)
    want = masks.expectation(psi, hx, hz, hc, 0.5)
    (e1, full1, st1, real1, stored1, cnt1), (e0, full0, st0, _, stored0, cnt0) = res[True], res[False]
    assert real1 and np.abs(np.asarray(full1) - psi).max() < 1e-12 and np.abs(np.asarray(full0) - psi).max() < 1e-12
    assert abs(e1 - want) < 1e-11 and abs(e0 - want) < 1e-11 and not stored1 and not stored0
    assert st1["real_exchanges"] == st1["swaps"] >= 1 and st1["real_chunk_reads"] == st1["chunk_reads"] > 0
    assert st1["bytes_sent"] * 2 == st0["bytes_sent"]
    # real STORAGE (the default on HIP shards): the shard held 2^n_local doubles through the sweeps and <H> — the real-amplitude
    # kernels ran (k_tile_sweep<REAL> / k_rot_pairs_real, k_tile_expect<REAL>, k_tile_cross_real), every sweep and every pass of
    # <H> moved half the bytes of the complex run, the wire carried 8 bytes per amplitude; state and energy against the oracle
    es, fulls, sts, reals, stored, cnts = res["stored"]
    assert stored and reals and np.abs(np.asarray(fulls) - psi).max() < 1e-12
    assert abs(es - want) < 1e-10 * np.abs(hc).sum()
    assert sts["bytes_sent"] == st1["bytes_sent"] and sts["real_exchanges"] == sts["swaps"] == st1["swaps"]
    assert cnts["rotation_bytes"] > 0 and cnts["contraction_bytes"] > 0
    if n - (world.bit_length() - 1) >= 15:    # shards large enough for the real tile sweeps (2^12 doubles from 14 local qubits on):
        assert cnts["rotation_bytes"] * 2 <= cnt0["rotation_bytes"] * 1.01 + 1     # 16 instead of 32 B per amplitude and sweep
        assert cnts["contraction_bytes"] <= 0.6 * cnt0["contraction_bytes"]


def test_compiled_program_on_hip_shards(gpu_lib):
    """the exchange plan of a rotation list made once, evaluated at three parameter vectors on eight HIP shards (one GPU, gloo)"""
    from tests.test_distributed import _program_worker
    world, n, chunk_bits = 8, 16, 10
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_program_worker, args=(r, world, port, n, 515, out, "hip", chunk_bits)) for r in range(world)]
    for p in procs:
        p.start()
    es, full, e_plain, swaps_per_run, plain_swaps, planned, t_plan, nsteps, (xs, zs, coeff, pidx, hx, hz, hc, hf, thetas) = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for e, th in zip(es, thetas):
        psi = np.zeros(1 << n, complex)
        psi[hf] = 1
        for x, z, c, k in zip(xs, zs, coeff, pidx):
            psi = masks.rotate(psi, int(x), int(z), c * th[k])
        assert abs(e - masks.expectation(psi, hx, hz, hc, 0.75)) < 1e-11
    assert np.abs(np.asarray(full) - psi).max() < 1e-12 and abs(e_plain - es[2]) < 1e-12
    assert swaps_per_run == plain_swaps == planned >= 1


@pytest.mark.parametrize("world", [2, 4])
def test_reference_entry_points_on_hip_shards(gpu_lib, world):
    """EnergyUCC.ucc_action, fermionic_adapt_vqe and qubit_adapt_vqe (two macro-iterations each) and exact exponentials with the
    register partitioned over HIP shards (threshold forced to 6 qubits; every rank's shard on this GPU, gloo): same picks, energies
    and norms as the single-process oracle engine"""
    from tests.test_partitioned_api import FLOWS, _launch, _single_process_oracle, check_partitioned
    check_partitioned(_launch(world, "hip", FLOWS, 6), _single_process_oracle(FLOWS))


def test_replicas_share_batches_and_pools_on_hip(gpu_lib):
    """below the threshold every rank holds the register on its GPU (here: two one-device handles on this GPU) and the ranks share
    the rows of a batch and the operators of a pool"""
    from tests.test_partitioned_api import _launch, _single_process_oracle, check_replicas
    check_replicas(_launch(2, "hip", ("replicas",), 31), _single_process_oracle(("replicas",)))
