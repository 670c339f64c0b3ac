"""world-size-W sharded state on HIP shards (all on device 0, gloo): apply rotations one by one, compare with the oracle"""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp
from oracle import masks

def worker(rank, world, port, n, seed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openvqe_amd.distributed import ShardedStatevector
    rng = np.random.default_rng(seed)
    g = world.bit_length() - 1
    R = 40
    def xmask():
        w = int(rng.integers(1, max(2, n - g)))
        return sum(1 << int(b) for b in rng.choice(n, w, replace=False))
    xs = [xmask() for _ in range(R)]; zs = [int(v) for v in rng.integers(0, 1 << n, R)]
    xs[3] = 0; xs[7] = xs[6]; xs[10] = 1 << (n - 1); zs[10] = 0
    phis = rng.uniform(-1, 1, R)
    hf = int(rng.integers(0, 1 << n))
    sv = ShardedStatevector(n, device=0)
    if os.environ.get('PIECES'): sv.EXCHANGE_PIECES = int(os.environ['PIECES'])
    sv.init_basis(hf)
    psi = np.zeros(1 << n, complex); psi[hf] = 1
    for r in range(R):
        sv.apply_pauli_rotations(xs[r:r + 1], zs[r:r + 1], phis[r:r + 1])
        full = sv.gather_state()
        psi = masks.rotate(psi, xs[r], zs[r], phis[r])
        if rank == 0:
            err = np.abs(np.asarray(full) - psi).max()
            print(r, hex(xs[r]), "global x" if xs[r] >> (n - g) else "", "err %.2e" % err, dict(sv.stats), "perm", sv.perm, flush=True)
            if err > 1e-10: break
    dist.destroy_process_group()

if __name__ == "__main__":
    world, n = int(sys.argv[1]), int(sys.argv[2])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, n, 4321 + n, out)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(timeout=300)
