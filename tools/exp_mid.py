"""molecule-shaped UCCSD energy evaluation at 2*m qubits on the streaming kernels (timing / profiling helper)"""
import sys, os, time, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import fermion
from openvqe_amd.backend import Statevector
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--cache" in sys.argv: args = [a for a in args if a != sys.argv[sys.argv.index("--cache") + 1]]
m, o = (int(args[0]), int(args[1])) if len(args) > 1 else (12, 5)
cache = sys.argv[sys.argv.index("--cache") + 1] if "--cache" in sys.argv else None
reps = 1 if "--once" in sys.argv else 4
t = time.time()
if cache and os.path.exists(cache):
    ham, gens, hf = pickle.load(open(cache, "rb"))
else:
    ham, gens, hf = fermion.synthetic_molecule(m, o, seed=24)
    if cache:
        pickle.dump((ham, gens, hf), open(cache, "wb"))
n = 2 * m
R = sum(len(g.terms) for g in gens); G = len(set(ham.packed()[0].tolist()))
print(f"n={n} build {time.time()-t:.1f}s terms={len(ham.terms)} groups={G} gens={len(gens)} rots={R}", flush=True)
theta = np.random.default_rng(1).uniform(-0.1, 0.1, len(gens))
opts = [a[6:].split("=") for a in sys.argv if a.startswith("--opt=")]
with Statevector(n) as sv:
    for k, v in opts: sv.set_option(k, int(v))
    t = time.time(); sv.set_hamiltonian(ham); sv.set_ucc_program(gens, hf); print(f"upload {time.time()-t:.2f}s", flush=True)
    for rep in range(reps):
        t = time.time(); sv.prepare_state(theta); t1 = time.time() - t
        t = time.time(); e2 = sv.energy(theta); t3 = time.time() - t
        t2 = t3 - t1
        print(f"prepare {t1*1e3:.1f} ms ({32*2**n*len(gens)/t1/1e9:.0f} GB/s per fused sweep; {32*2**n*R/t1/1e9:.0f} GB/s per rotation)  "
              f"expectation ~{t2*1e3:.1f} ms ({16*2**n*G/t2/1e9:.0f} GB/s algorithmic)  energy {t3*1e3:.1f} ms  "
              f"B_eval={(32*2**n*R+16*2**n*G)/1e9:.0f} GB -> {(32*2**n*R+16*2**n*G)/t3/1e9:.0f} GB/s  E={e2:.10f}", flush=True)
    print(sv.program_info(), flush=True)
