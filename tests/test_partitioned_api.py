"""The entry points north_star names — ``EnergyUCC.ucc_action``, ``fermionic_adapt_vqe``, ``qubit_adapt_vqe`` (ref:openvqe/ucc_family/
get_energy_ucc.py:8-50, ref:openvqe/adapt/fermionic_adapt_vqe.py:371-593, ref:openvqe/adapt/qubit_adapt_vqe.py:310-605) — run
UNCHANGED on several ranks:

  * with the register PARTITIONED across the ranks (openvqe_amd.partitioned.PartitionedStatevector behind evaluator / HipQPU / the ADAPT
    screens; threshold forced down to 6 qubits so that H2/6-31G's 8 qubits are sharded over 2 and 4 ranks), shard arithmetic by the
    oracle engine of tests/test_distributed.py over gloo — energies, picked pool indices and gradient norms equal those of the
    single-process oracle engine;
  * below the threshold as REPLICAS: the rows of a forward-difference batch and the operators of an ADAPT pool shared between the
    ranks, one all-gather.
The same workers run on HIP shards in tests/test_gpu_distributed.py."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from openvqe_amd import chem, pools
from tests.oracle_backend import OracleStatevector
from tests.test_distributed import OracleShardEngine, _free_port


def _h2_problem():
    mol = chem.molecule("H2")
    mol.rhf()
    ham = mol.jw_hamiltonian()
    _, pool = pools.spin_complement_gsd(mol.n_elec, mol.nao)
    _, qpool = pools.qubit_pool("random", 8, rng=np.random.default_rng(4))
    return mol, ham, pool, qpool


def _reset():
    import openvqe_amd.adapt.fermionic_adapt_vqe as fa
    import openvqe_amd.adapt.qubit_adapt_vqe as qa
    import openvqe_amd.evaluator as ev
    import openvqe_amd.qat_compat as qc
    for cache in (ev._BACKENDS, ev._Evaluator._owner, fa._screens, fa._evaluators, qa._screens, qa._evaluators):
        cache.clear()
    qc._default_qpu = None


def run_flows(flows):
    """the reference-named entry points on whatever backend the process is set up for -> dict of results"""
    from openvqe_amd.adapt.fermionic_adapt_vqe import fermionic_adapt_vqe
    from openvqe_amd.adapt.qubit_adapt_vqe import qubit_adapt_vqe
    from openvqe_amd.ucc_family.get_energy_ucc import EnergyUCC
    mol, ham, pool, qpool = _h2_problem()
    hf = mol.hf_init()
    out = {}
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        if "ucc" in flows:
            gens = [complex(0.0, 1.0) * pool[k] for k in (38, 32, 29)]     # Hermitian generators 1j * (T - T^+)
            ucc = EnergyUCC()
            energies = []
            for th in ([0.0, 0.0, 0.0], [0.05, -0.02, 0.11], [-0.3, 0.2, 0.4]):
                ucc.ucc_action(th, ham, gens, hf, energies)
            out["ucc"] = energies
        if "fermionic" in flows:
            it, res = fermionic_adapt_vqe(None, None, None, ham, pool, hf, 1, -1.1516885475166094, "COBYLA", 1e-6, "norm", 1e-2, 2)
            out["fermionic"] = (it["energies"], it["norms"], it["Max_gradients"], it["CNOTs"])
            import openvqe_amd.adapt.fermionic_adapt_vqe as fa
            screen = next(iter(fa._screens.values()))
            out["fermionic_screen_norm2"] = screen.norm2()
        if "qubit" in flows:
            it, _, res, _ = qubit_adapt_vqe(ham, None, None, 8, qpool, hf, -1.1516885475166103, n_max_grads=1, adapt_conver="norm",
                                            adapt_thresh=1e-7, adapt_maxiter=2, tolerance_sim=1e-9, method_sim="BFGS")
            out["qubit"] = (it["energies"], it["norms"], it["Max_gradient"])
        if "taylor" in flows:
            # exp(theta A) of an operator whose strings do NOT commute (X and Z on qubit 0): the Taylor series of sigma = A psi on the
            # partitioned register; and one whose strings do (a pool operator): rotations
            from openvqe_amd.operators import Hamiltonian, Term
            from openvqe_amd.partitioned import make_backend
            op = Hamiltonian(8, [Term(0.7j, "XY", [0, 1]), Term(0.4j, "ZY", [0, 5]), Term(-0.3j, "YZX", [7, 3, 2])])
            sv = make_backend(8)
            sv.init_basis(hf)
            sv.apply_exp_pauli_sum(pool[38], 0.21)
            sv.apply_exp_pauli_sum(op, 0.35)
            sv.apply_exp_pauli_sum(qpool[20], -0.4, prefactor=-1j)
            out["taylor"] = np.asarray(sv.get_state())
        if "replicas" in flows:
            from openvqe_amd.evaluator import UCCEvaluator
            gens = [complex(0.0, 1.0) * pool[k] for k in (38, 32, 29, 23, 2)]
            ev = UCCEvaluator(ham, gens, hf)
            pts = np.random.default_rng(5).uniform(-0.2, 0.2, (7, 5))      # 7 rows over 2 / 4 ranks: ragged shares
            out["batch"] = ev.energy_batch(pts).tolist()
            import openvqe_amd.adapt.fermionic_adapt_vqe as fa
            screen = fa.prepare_adapt_state(hf, [pool[38]], [0.07], ham)
            out["screen"] = fa.return_signed_gradients(pool, ham, screen)
    out["printed_new_ansatz"] = sink.getvalue().count("New ansatz created")
    out["picks"] = _selected(sink.getvalue())
    return out


def _selected(printed):
    """pool indices the fermionic loop printed as its picks ("sorted_index1:  [38]")"""
    import re
    return [int(v) for v in re.findall(r"sorted_index1:  \[(\d+)\]", printed)]


def _worker(rank, world, port, out, engine, flows, threshold):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OVQE_PARTITION_MIN_QUBITS"] = str(threshold)
    os.environ["OVQE_SHARD_CHUNK_BITS"] = "3"
    os.environ["OVQE_SINGLE_DEVICE"] = "1"          # (HIP engine: every rank's shard on device 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import openvqe_amd.backend as be
        import openvqe_amd.evaluator as ev
        import openvqe_amd.partitioned as part
        if engine == "oracle":
            import openvqe_amd.adapt.fermionic_adapt_vqe as fa
            import openvqe_amd.adapt.qubit_adapt_vqe as qa
            part.ENGINE_FACTORY = lambda nl, ng, r: OracleShardEngine(nl, ng, r)
            be.Statevector = ev.Statevector = fa.Statevector = qa.Statevector = OracleStatevector   # (registers below the threshold: the one-device stand-in)
        _reset()
        res = run_flows(flows)
        if threshold <= 8:
            from openvqe_amd.partitioned import PartitionedStatevector
            assert all(isinstance(sv, PartitionedStatevector) for sv in ev._BACKENDS.values()) and ev._BACKENDS
        if rank == 0:
            out.put(res)
    finally:
        dist.destroy_process_group()


def _single_process_oracle(flows):
    import openvqe_amd.adapt.fermionic_adapt_vqe as fa
    import openvqe_amd.adapt.qubit_adapt_vqe as qa
    import openvqe_amd.backend as be
    import openvqe_amd.evaluator as ev
    mods = (be, ev, fa, qa)
    saved = [m.Statevector for m in mods]
    for m in mods:
        m.Statevector = OracleStatevector
    _reset()
    try:
        return run_flows(flows)
    finally:
        for m, cls in zip(mods, saved):
            m.Statevector = cls
        _reset()


def _launch(world, engine, flows, threshold):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, engine, flows, threshold)) for r in range(world)]
    for p in procs:
        p.start()
    res = out.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def check_partitioned(got, want):
    assert np.abs(np.array(got["ucc"]) - np.array(want["ucc"])).max() < 1e-11
    ge, gn, gm, gc = got["fermionic"]
    we, wn, wm, wc = want["fermionic"]
    assert got["picks"] == want["picks"] == [38, 32]              # ADAPT ranking identical: K3's first two picks
    assert gc == wc == [48, 96]
    assert np.abs(np.array(ge) - np.array(we)).max() < 1e-9       # (optimiser end points: COBYLA on identical energies)
    assert np.abs(np.array(gn) - np.array(wn)).max() < 1e-7 and np.abs(np.array(gm) - np.array(wm)).max() < 1e-7
    assert abs(got["fermionic_screen_norm2"] - 1.0) < 1e-12
    qe, qn, qm = got["qubit"]
    assert np.abs(np.array(qe) - np.array(want["qubit"][0])).max() < 1e-9
    assert np.abs(np.array(qn) - np.array(want["qubit"][1])).max() < 1e-6 and np.abs(np.array(qm) - np.array(want["qubit"][2])).max() < 1e-6
    assert got["printed_new_ansatz"] == want["printed_new_ansatz"] == 2


    assert np.abs(got["taylor"] - want["taylor"]).max() < 1e-12 and abs(np.vdot(got["taylor"], got["taylor"]).real - 1.0) < 1e-12


FLOWS = ("ucc", "fermionic", "qubit", "taylor")


@pytest.mark.parametrize("world", [2, 4])
def test_reference_entry_points_on_the_partitioned_register(world):
    check_partitioned(_launch(world, "oracle", FLOWS, 6), _single_process_oracle(FLOWS))


def check_replicas(got, want):
    assert np.abs(np.array(got["batch"]) - np.array(want["batch"])).max() < 1e-12
    assert np.abs(np.array(got["screen"]) - np.array(want["screen"])).max() < 1e-12 and len(got["screen"]) == len(want["screen"])


@pytest.mark.parametrize("world", [2, 4])
def test_batches_and_pools_are_shared_between_replicas_below_the_threshold(world):
    check_replicas(_launch(world, "oracle", ("replicas",), 31), _single_process_oracle(("replicas",)))
