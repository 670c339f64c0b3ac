"""fermionic ADAPT-VQE (mirror of ref:openvqe/adapt/fermionic_adapt_vqe.py) on N2 / cc-pVDZ (10e,12o) = 24 qubits with the UCCSD
singlet singles-and-doubles pool (ref:openvqe/common_files/generator_excitations.py:274-359): a few macro-iterations, wall time per phase"""
import os, sys, time, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem
from openvqe_amd import pools
from openvqe_amd.adapt import fermionic_adapt_vqe as fav
mol = chem.molecule("N2-CCPVDZ"); e_rhf = mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
size, cluster_ops, spin_ops, theta_mp2, hf = prob.uccsd()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if "--sector-ground-space" in sys.argv:   # the fun_fidelity reference vector from ovqe_sector_ground_state (opt-in of the mirror)
    fav.SECTOR_GROUND_SPACE = True
t = time.perf_counter(); pool_size, _, pool = pools.singlet_sd(10, 12); print(f'pool {pool_size} operators, built in {time.perf_counter()-t:.1f}s', flush=True)
if "--nogc" in sys.argv:   # experiment: are the 70-ms evaluations full collections of the cyclic garbage collector?
    import gc; gc.collect(); gc.disable()
if "--freeze" in sys.argv:
    import gc; gc.collect(); gc.freeze()
t0 = time.perf_counter()
buf = io.StringIO()
marks = []
orig_screen = fav.return_signed_gradients
def timed_screen(*a, **k):
    t = time.perf_counter(); r = orig_screen(*a, **k); marks.append(("screen", time.perf_counter() - t)); return r
fav.return_signed_gradients = timed_screen
orig_action = fav.ucc_action
def timed_action(*a, **k):
    t = time.perf_counter(); r = orig_action(*a, **k); d = time.perf_counter() - t; marks.append(("energy", d))
    if d > 5e-3 and "--slow" in sys.argv:
        print(f"slow evaluation #{len(marks)}: {1e3 * d:.1f} ms", file=sys.stderr, flush=True)
    return r
fav.ucc_action = timed_action
def _timed(name):
    orig = getattr(fav, name)
    def wrapper(*a, **k):
        t = time.perf_counter(); r = orig(*a, **k); marks.append((name, time.perf_counter() - t)); return r
    setattr(fav, name, wrapper)
if "--slow" in sys.argv:   # which backend call the time of a slow evaluation went to
    from openvqe_amd.backend import Statevector as _SV
    def _wrap_sv(name):
        orig = getattr(_SV, name)
        def w(self, *a, **k):
            t = time.perf_counter(); r = orig(self, *a, **k); d = time.perf_counter() - t
            if d > 2e-3:
                print(f"   Statevector.{name}: {1e3 * d:.1f} ms", file=sys.stderr, flush=True)
            return r
        setattr(_SV, name, w)
    for name in ("set_hamiltonian", "set_ucc_program", "energy", "pool_gradients", "prepare_state", "get_support", "apply_exp_pauli_sum"):
        if hasattr(_SV, name):
            _wrap_sv(name)
for name in ("_ground_space", "fun_fidelity", "prepare_adapt_state", "prepare_state_ansatz", "hf_energy"):
    _timed(name)
try:
    with contextlib.redirect_stdout(buf):
        trace, result = fav.fermionic_adapt_vqe(None, None, None, ham, pool, hf, 1, -109.0745445341, "COBYLA", 1e-6, "norm", 1e-3, iters)
except Exception:
    print(buf.getvalue()[-1500:]); raise
wall = time.perf_counter() - t0
scr = [d for k, d in marks if k == "screen"]; en = [d for k, d in marks if k == "energy"]
slow = sorted(en, reverse=True)
print(f"energy evaluations: {sum(1 for d in en if d > 5e-3)} above 5 ms totalling {sum(d for d in en if d > 5e-3):.2f}s; largest {[round(1e3*d,1) for d in slow[:6]]} ms; "
      f"quartiles {[round(1e3*float(q),3) for q in np.percentile(en, [25, 50, 75, 95])]} ms")
print("screen ms by macro-iteration:", [round(1e3 * d, 1) for d in scr])
print(f"wall={wall:.1f}s screens={len(scr)} ({np.mean(scr)*1e3:.0f} ms each) energy evaluations={len(en)} (median {np.median(en)*1e3:.2f} ms, total {sum(en):.2f}s)")
for name in ("_ground_space", "fun_fidelity", "prepare_adapt_state", "prepare_state_ansatz", "hf_energy"):
    d = [x for k, x in marks if k == name]
    print(f"  {name}: {len(d)} calls, {sum(d):.2f}s")
import re
print("selected pool indices:", [int(m) for m in re.findall(r"sorted_index1:\s*\[(\d+)\]", buf.getvalue())])
print({k: v for k, v in result.items() if not isinstance(v, (list, dict))})
print("energies", trace.get("energies", trace)[:10] if isinstance(trace, dict) else trace)
