"""One ADAPT-VQE engine for both flavours of the reference.

The reference carries two near-identical loops (ref:openvqe/adapt/fermionic_adapt_vqe.py:484-593,
ref:openvqe/adapt/qubit_adapt_vqe.py:446-604).  Here the loop exists once, as a small state machine

    screen -> rank -> stop? -> grow -> optimise -> record

and everything in which the two flavours differ is DATA on a ``Flavour`` record: the gradient mode of the device screen,
the rule for the new parameters, the stall threshold, which norm goes into the trace, the printed lines and the key
names of the result dictionaries (those are the contract the reference's callers and notebooks read:
ref:openvqe/algorithms/fermionic_adapt.py:57-75, ref:openvqe/algorithms/qubit_adapt.py:70-88).

Behaviour that is reproduced on purpose because it is observable in the stored traces:
  * ranking drops exact zeros and resolves ties to the lower pool index (common_files/sorted_gradient.py);
  * fermionic: every new parameter starts at 0.01; the recorded norm is the norm of the ranked (non-zero) gradients;
  * qubit: new parameters are the leading gradients divided by a "norm" whose square root is taken INSIDE the
    accumulation loop (ref:…qubit_adapt_vqe.py:530-532) — i.e. r_k = sqrt(r_{k-1} + g_k^2) — and the recorded norm is the
    plain one;
  * stopping on |norm - previous norm| below 1e-8 (fermionic) / 1e-7 (qubit) as well as on the threshold;
  * stopping before any optimisation has run is an error in the reference (it reads the last optimiser result,
    ref:…fermionic_adapt_vqe.py:531): raised here as the same NameError.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import scipy.optimize

from ..common_files.circuit import count
from ..common_files.host_threads import one_blas_thread
from ..common_files.sorted_gradient import abs_sort_desc, corresponding_index, index_without_0, value_without_0


@dataclass
class Flavour:
    title: str                                   # banner text of an iteration
    stall: float                                 # |norm - previous norm| below this also stops the growth
    trace_keys: Dict[str, str]                   # canonical name -> key in the ``iterations`` dictionary
    gate_keys: Sequence[str] = ("CNOTs", "Hadamard", "RY", "RX")
    gate_names: Sequence[str] = ("CNOT", "H", "RY", "RX")
    unknown_criterion: str = " type convergence is not defined"
    optimiser_display: bool = True


def rank_gradients(gradients: Sequence[float]):
    """(non-zero values by decreasing magnitude, matching pool indices) — exact zeros dropped, ties -> lower index"""
    kept = value_without_0(gradients)
    where = index_without_0(gradients)
    ordered = abs_sort_desc(list(kept))
    return ordered, corresponding_index(kept, where, ordered)


def iterated_root_norm(values: Sequence[float]) -> float:
    """r <- sqrt(r + v^2) over the values: the quantity the reference's qubit flavour divides by"""
    r = 0.0
    for v in values:
        r = float(np.sqrt(r + v * v))
    return r


@dataclass
class AdaptEngine:
    """state of one ADAPT run; the callables bind it to a backend (device screen, compiled energy, circuits)"""
    flavour: Flavour
    pool: Sequence                               # pool operators (what ``indices`` index)
    screen_gradients: Callable[[], List[float]]  # gradients of the whole pool on the current screen state
    energy: Callable[[Sequence, Sequence[float]], float]   # E(theta) of the Trotterised ansatz
    make_generator: Callable[[int], object]      # pool index -> generator of the energy circuit
    new_parameters: Callable[[List[float], List[int], int], List[float]]   # (ranked values, ranked idx, how many)
    rebuild: Callable[[List[int], List[float]], object]     # -> circuit of the optimised ansatz (+ refreshes the screen)
    on_iteration: Optional[Callable[[object], Dict[str, float]]] = None    # extra per-iteration records (fidelity)
    say: Callable[..., None] = print
    selected: List[int] = field(default_factory=list)
    generators: List = field(default_factory=list)
    theta: List[float] = field(default_factory=list)
    last_fit: Optional[scipy.optimize.OptimizeResult] = None
    circuit: object = None
    previous_norm: float = 0.0

    # -- phases ------------------------------------------------------------------------------------------------------
    def screen(self):
        grads = [float(g) for g in self.screen_gradients()]
        total = 0.0
        leader, leader_at = 0.0, 0
        for k, g in enumerate(grads):
            total += g * g
            if abs(g) > abs(leader):             # strict: the first maximum wins
                leader, leader_at = g, k
        return grads, float(np.sqrt(total)), leader, leader_at

    def should_stop(self, norm: float, criterion: str, threshold: float) -> bool:
        if criterion != "norm":
            self.say(self.flavour.unknown_criterion)
            raise SystemExit()
        return norm < threshold or abs(norm - self.previous_norm) < self.flavour.stall

    def require_fit(self):
        if self.last_fit is None:
            raise NameError("name 'opt_result' is not defined (the growth stopped before any optimisation ran)")
        return self.last_fit

    def grow(self, ranked_values, ranked_index, how_many: int):
        fresh = self.new_parameters(ranked_values, ranked_index, how_many)
        picks = [ranked_index[k] for k in range(how_many)]
        for idx, t0 in zip(picks, fresh):
            self.theta.append(t0)
            self.generators.append(self.make_generator(idx))
            self.selected.append(idx)
        return picks

    def optimise(self, method: str, tol: float):
        gens = self.generators
        with one_blas_thread():   # (host_threads.py: spinning BLAS workers between device calls)
            self.last_fit = scipy.optimize.minimize(lambda t: self.energy(gens, t), x0=self.theta, method=method, tol=tol,
                                                    options={"maxiter": 100000, "disp": self.flavour.optimiser_display})
        self.theta = [float(v) for v in self.last_fit.x[: len(gens)]]
        self.circuit = self.rebuild(self.selected, self.theta)
        return self.last_fit

    def record(self, trace: Dict[str, list], values: Dict[str, float]):
        keys = self.flavour.trace_keys
        for name, v in values.items():
            trace[keys[name]].append(v)
        ops = self.circuit.ops
        for key, gate in zip(self.flavour.gate_keys, self.flavour.gate_names):
            trace[key].append(count(gate, ops))
