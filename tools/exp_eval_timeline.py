"""a few single evaluations of an ADAPT-sized ansatz at 24 qubits (K spin-adapted generators of the N2 pool), nothing else: the
workload of tools/trace_timeline.sh (kernel start / end times of the last evaluations).  python tools/exp_eval_timeline.py [K=28] [evals=20] [name=value]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools
from openvqe_amd.backend import Statevector
pos = [a for a in sys.argv[1:] if "=" not in a]
K = int(pos[0]) if pos else 28
nev = int(pos[1]) if len(pos) > 1 else 20
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, singlets = pools.singlet_sd(10, 12)
rng = np.random.default_rng(3)
# the operators the fermionic ADAPT run on this molecule selects, in its order (tools/exp_adapt_n2.py 30)
order = [589, 629, 567, 637, 358, 266, 396, 259, 286, 176, 214, 638, 561, 313, 461, 547, 496, 601, 588, 452, 545, 498, 150, 633, 628, 583,
         660, 590, 53, 95]
if K > len(order):
    order = order + [int(k) for k in rng.permutation(len(singlets)) if k not in order]
gens = [1j * singlets[k] for k in order[:K]]
with Statevector(24) as sv:
    for a in sys.argv[1:]:
        if "=" in a:
            k, v = a.split("="); sv.set_option(k, int(v))
    sv.set_hamiltonian(ham)
    sv.set_ucc_program(gens, hf)
    theta = rng.uniform(-0.1, 0.1, K)
    ts = []
    for rep in range(nev):
        t = time.perf_counter(); e = sv.energy(theta + 0.001 * rep); ts.append(1e6 * (time.perf_counter() - t))
    info = sv.program_info()
    print(f"K={K}: us per evaluation (last 8) {[round(t) for t in ts[-8:]]}; support {info.get('sector_support')}, {info.get('sector_sweeps')} circuit sweeps, {info.get('sector_h_sweeps')} <H> sweeps")
