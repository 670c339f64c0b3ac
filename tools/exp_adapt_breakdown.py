"""Per macro-iteration breakdown of the 30-iteration fermionic-ADAPT mirror on N2 / cc-pVDZ (10e,12o): evaluations, the first energy call
after a new program (table build inside), the steady-state evaluation, exponentials, screen — and which path served the program"""
import os, sys, io, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openvqe_amd import chem, pools, backend
from openvqe_amd.adapt import fermionic_adapt_vqe as fav
mol = chem.molecule("N2-CCPVDZ"); mol.rhf()
prob = chem.cas_problem(mol, 2, 12)
ham = prob.jw_hamiltonian()
_, _, _, _, hf = prob.uccsd()
_, _, pool = pools.singlet_sd(10, 12)
fav.SECTOR_GROUND_SPACE = True
fav._FLAVOUR.optimiser_display = False
SV = backend.Statevector
log = {"iters": [], "cur": None}
def new_iter(sv):
    log["cur"] = {"evals": [], "set_program_s": 0.0, "sv": sv}
    log["iters"].append(log["cur"])
orig_energy, orig_setp, orig_exp, orig_pool = SV.energy, SV.set_rotation_program, SV.apply_exp_pauli_sum, SV.pool_gradients
def energy(self, theta):
    t = time.perf_counter(); e = orig_energy(self, theta); log["cur"]["evals"].append(time.perf_counter() - t); return e
def setp(self, *a, **k):
    new_iter(self)
    if os.environ.get('OVQE_LIB') == 'testing':
        self.set_option('sector_debug', 4 if len(log['iters']) in (12, 42, 60) else 0)
        if len(log['iters']) in (12, 42, 60): print('---- program', len(log['iters']) - 1, file=sys.stderr, flush=True)
    t = time.perf_counter(); r = orig_setp(self, *a, **k); log["cur"]["set_program_s"] = time.perf_counter() - t; return r
tot = {"exp": 0.0, "nexp": 0, "pool": 0.0}
def ex(self, *a, **k):
    t = time.perf_counter(); r = orig_exp(self, *a, **k); tot["exp"] += time.perf_counter() - t; tot["nexp"] += 1; return r
def pg(self, *a, **k):
    t = time.perf_counter(); r = orig_pool(self, *a, **k); tot["pool"] += time.perf_counter() - t; return r
SV.energy, SV.set_rotation_program, SV.apply_exp_pauli_sum, SV.pool_gradients = energy, setp, ex, pg
infos = []
orig_opt = None
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    fav.fermionic_adapt_vqe(None, None, None, ham, pool, hf, 1, -109.0745445341, "COBYLA", 1e-6, "norm", 1e-3, 30)
wall = time.perf_counter() - t0
print("wall %.3f s; exponentials %d calls %.3f s; screens %.3f s" % (wall, tot["nexp"], tot["exp"], tot["pool"]))
tb = ts = 0.0
for k, it in enumerate(log["iters"]):
    ev = np.array(it["evals"])
    if ev.size == 0:
        continue
    first = ev[0]; rest = ev[1:] if ev.size > 1 else ev
    tb += first + it["set_program_s"]; ts += rest.sum()
    info = it["sv"].program_info() if k == len(log["iters"]) - 1 else {}
    print("prog %2d: evals %4d  set_program %.2f ms  first %.2f ms  steady median %.1f us  mean %.1f us  sum %.3f s" %
          (k, ev.size, 1e3 * it["set_program_s"], 1e3 * first, 1e6 * np.median(rest), 1e6 * rest.mean(), ev.sum()))
print("builds (set_program + first call) %.3f s; steady evaluations %.3f s" % (tb, ts))
print({k: v for k, v in info.items() if v})
