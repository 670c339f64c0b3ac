"""Stand-ins for the slice of myQLM (``qat.*``) that OpenVQE's L1 hot path touches, executing on the
HIP backend.  With ``install()`` these modules are registered under the ``qat`` names so that the
reference's own ``openvqe/ucc_family/*.py`` and ``openvqe/adapt/*.py`` import and run unchanged
(INTEGRATION.md); the in-tree mirrors under ``openvqe_amd/ucc_family`` / ``openvqe_amd/adapt`` use
the same objects for circuits and gate counting.

Surface covered: ``Program().qalloc/apply/to_circ``, gates ``X H CNOT RX RY RZ``,
``build_ucc_ansatz([op], init, n_steps=1)([theta])``, ``Circuit.to_job(job_type="OBS", observable=H)``,
``Circuit.to_job()``, ``Circuit.ops``, ``get_default_qpu().submit(job)`` -> ``Result.value`` /
iteration over samples with ``.state.int`` and ``.amplitude``, ``Term`` / ``Hamiltonian``.

myQLM conventions restated here (third-party, not in /root/reference): qubit 0 is the most
significant bit of a basis index; ``build_ucc_ansatz`` prepares ``init`` with X gates and appends a
one-step Trotter slice prod_j exp(-i theta c_j P_j) in ``terms`` order, each factor synthesised as
basis change (H for X, RX(pi/2) for Y) + CNOT staircase + RZ(2 theta c) + uncompute — that synthesis
is what ``Circuit.ops`` lists for gate counting (ref:openvqe/common_files/circuit.py:186-205
``count``: 2(w-1) CNOT and 2 H per X per weight-w string, cf. the stored
``"CNOTs": [6, 12, ...]`` of ref:notebooks/demo_qubit_adapt.ipynb).
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np

from .operators import Hamiltonian, Observable, SpinHamiltonian, Term, pack_string  # noqa: F401


# ------------------------------------------------------------------------------------ parameters
class AffineParam:
    """scale * theta[index] + const — lets the gate templates of circuit.py be traced once
    (``RY(-2 * theta)``) instead of rebuilt for every energy evaluation."""

    __slots__ = ("index", "scale", "const")

    def __init__(self, index, scale=1.0, const=0.0):
        self.index, self.scale, self.const = int(index), float(scale), float(const)

    def __neg__(self):
        return AffineParam(self.index, -self.scale, -self.const)

    def __mul__(self, k):
        return AffineParam(self.index, self.scale * k, self.const * k)

    __rmul__ = __mul__

    def __truediv__(self, k):
        return self * (1.0 / k)

    def __add__(self, c):
        return AffineParam(self.index, self.scale, self.const + c)

    __radd__ = __add__

    def __sub__(self, c):
        return self + (-c)

    def value(self, theta):
        return self.scale * theta[self.index] + self.const


def _affine(angle):
    """-> (scale, const, index) with index -1 for a plain number."""
    if isinstance(angle, AffineParam):
        return angle.scale, angle.const, angle.index
    return 0.0, float(angle), -1


# ------------------------------------------------------------------------------------ gates
class Gate:
    def __init__(self, name, arity=1, angle=None):
        self.name, self.arity, self.angle = name, arity, angle

    def __repr__(self):
        return f"Gate({self.name})"


X = Gate("X")
Y = Gate("Y")
Z = Gate("Z")
H = Gate("H")
CNOT = Gate("CNOT", 2)


def RX(angle):
    return Gate("RX", 1, angle)


def RY(angle):
    return Gate("RY", 1, angle)


def RZ(angle):
    return Gate("RZ", 1, angle)


class Op:
    """One circuit instruction as listed by ``Circuit.ops``; ``str()`` carries ``gate='NAME'``."""

    __slots__ = ("gate", "qbits", "angle")

    def __init__(self, gate, qbits, angle=None):
        self.gate, self.qbits, self.angle = gate, list(qbits), angle

    def __str__(self):
        return f"Op(gate='{self.gate}', qbits={self.qbits}, type=0, cbits=None, formula=None, remap=None)"

    __repr__ = __str__


class PauliEvolution:
    """Routine returned by ``build_ucc_ansatz(ops, init, n_steps)(thetas)``."""

    def __init__(self, operators, init, thetas, nbqbits):
        self.operators, self.init, self.thetas, self.arity = operators, int(init), list(thetas), nbqbits

    def rotations(self):
        """[(op_string, qbits, angle)] in application order; angle = theta * coeff (may be affine)."""
        out = []
        for op, theta in zip(self.operators, self.thetas):
            for term in op.terms:
                c = complex(term.coeff)
                if abs(c.imag) > 1e-12 * max(1.0, abs(c.real)):
                    raise ValueError("build_ucc_ansatz: generator has a non-real Pauli coefficient "
                                     "(pass the Hermitian operator, i.e. cluster_op * 1j)")
                out.append((term.op, list(term.qbits), theta * c.real))
        return out


def build_ucc_ansatz(cluster_ops, ket_hf_init, n_steps=1):
    """Stand-in of qat.fermion.chemistry.ucc_deprecated.build_ucc_ansatz (call sites
    ref:openvqe/ucc_family/get_energy_ucc.py:44,86, ref:openvqe/adapt/fermionic_adapt_vqe.py:158,302,
    ref:openvqe/adapt/qubit_adapt_vqe.py:181,303).  Returns a callable taking the list of angles."""
    if n_steps != 1:
        raise NotImplementedError("only the single Trotter step used by OpenVQE is supported")
    nbq = cluster_ops[0].nbqbits

    def routine(thetas):
        return PauliEvolution(list(cluster_ops), ket_hf_init, thetas, nbq)

    return routine


# ------------------------------------------------------------------------------------ program / circuit
class Program:
    def __init__(self):
        self.nbqbits = 0
        self.items = []  # ("gate", Gate, [qubits]) | ("evolution", PauliEvolution, [qubits])

    def qalloc(self, n):
        start = self.nbqbits
        self.nbqbits += int(n)
        return list(range(start, start + int(n)))

    def apply(self, what, *qargs):
        qubits = []
        for q in qargs:
            qubits.extend(q if isinstance(q, (list, tuple)) else [q])
        if isinstance(what, PauliEvolution):
            if len(qubits) != what.arity:
                raise ValueError("routine arity does not match the register")
            self.items.append(("evolution", what, qubits))
        elif isinstance(what, Gate):
            if len(qubits) != what.arity:
                raise ValueError(f"gate {what.name} takes {what.arity} qubit(s)")
            self.items.append(("gate", what, qubits))
        elif isinstance(what, Program):
            self.items.extend(what.items)
        else:
            raise TypeError(f"cannot apply {what!r}")
        return self

    def to_circ(self, **_):
        return Circuit(self.nbqbits, list(self.items))


QRoutine = Program


class OpList(list):
    """the list ``Circuit.ops`` returns: a plain list of ``Op`` that also knows how many instructions carry each gate name, so
    that ``common_files.circuit.count`` answers its four questions per ADAPT iteration without formatting every instruction as text
    four times (76 000 instructions at the 30th iteration of the N2 run: 0.4 of its 3.1 s).  Round 6: the list is LAZY — the counts
    of a circuit of Pauli evolutions follow from the strings (2 #X Hadamards, 2 #Y RX, 2 (w - 1) CNOT, one RZ per string), cached per
    operator object, so an ADAPT iteration that only counts gates (ref:openvqe/adapt/fermionic_adapt_vqe.py:520-527) never builds the
    76 000 ``Op`` objects; anything that looks at the instructions themselves fills the list first."""

    def __init__(self, circuit=None):
        super().__init__()
        self._circuit = circuit          # pending synthesis

    def _fill(self):
        if self._circuit is not None:
            circuit, self._circuit = self._circuit, None
            super().extend(circuit._synthesise())

    def gate_count(self, gate):
        if self._circuit is not None:
            counts = self._circuit.gate_counts()
            if counts is not None:
                return counts.get(gate, 0)
            self._fill()
        counts = getattr(self, "_gate_counts", None)
        if counts is None or self._counted_len != list.__len__(self):
            counts = {}
            for op in list.__iter__(self):
                if type(op) is not Op:
                    return None   # foreign elements: the caller formats them
                counts[op.gate] = counts.get(op.gate, 0) + 1
            self._gate_counts, self._counted_len = counts, list.__len__(self)
        return counts.get(gate, 0)


def _lazy(name):
    method = getattr(list, name)

    def wrapper(self, *args, **kwargs):
        self._fill()
        return method(self, *args, **kwargs)
    wrapper.__name__ = name
    return wrapper


for _name in ("__len__", "__iter__", "__getitem__", "__setitem__", "__delitem__", "__contains__", "__reversed__", "__eq__", "__ne__",
              "__add__", "__iadd__", "__mul__", "__repr__", "append", "extend", "insert", "pop", "remove", "index", "count", "copy",
              "sort", "reverse", "clear"):
    setattr(OpList, _name, _lazy(_name))
OpList.__hash__ = None

_string_counts = {}   # id(operator) -> (operator, number of terms, {gate: count})


def _operator_gate_counts(op):
    """gates of the CNOT-staircase synthesis of every string of ``op`` (what ``Circuit._synthesise`` emits for it)"""
    hit = _string_counts.get(id(op))
    terms = op.terms
    if hit is not None and hit[0] is op and hit[1] == len(terms):
        return hit[2]
    h = rx = cnot = rz = 0
    for term in terms:
        c = complex(term.coeff)
        if abs(c.imag) > 1e-12 * max(1.0, abs(c.real)):
            return None          # (PauliEvolution.rotations raises for it: let the synthesis do so)
        s = term.op
        nx, ny = s.count("X"), s.count("Y")
        w = len(s) - s.count("I")
        if w == 0:
            continue
        h += 2 * nx
        rx += 2 * ny
        cnot += 2 * (w - 1)
        rz += 1
    counts = {"H": h, "RX": rx, "CNOT": cnot, "RZ": rz}
    if len(_string_counts) > 4096:
        _string_counts.clear()
    _string_counts[id(op)] = (op, len(terms), counts)
    return counts


class Circuit:
    def __init__(self, nbqbits, items):
        self.nbqbits, self.items = nbqbits, items

    @property
    def ops(self):
        """Gate list after synthesis of the Pauli evolutions (CNOT staircase), for ``count`` — synthesised when first looked at"""
        return OpList(self)

    def gate_counts(self):
        """{gate name: instructions} of ``ops`` without building them; None when a string cannot be counted from its text"""
        total = {}
        for kind, what, qubits in self.items:
            if kind == "gate":
                total[what.name] = total.get(what.name, 0) + 1
                continue
            nx = bin(what.init & ((1 << what.arity) - 1)).count("1")
            if nx:
                total["X"] = total.get("X", 0) + nx
            for op, _ in zip(what.operators, what.thetas):
                counts = _operator_gate_counts(op)
                if counts is None:
                    return None
                for g, c in counts.items():
                    if c:
                        total[g] = total.get(g, 0) + c
        return total

    def _synthesise(self):
        out = []
        for kind, what, qubits in self.items:
            if kind == "gate":
                out.append(Op(what.name, qubits, what.angle))
                continue
            n = what.arity
            for q in range(n):
                if (what.init >> (n - 1 - q)) & 1:
                    out.append(Op("X", [qubits[q]]))
            for pauli, qs, angle in what.rotations():
                qs = [qubits[q] for q in qs]
                act = [(q, p) for q, p in zip(qs, pauli) if p != "I"]
                if not act:
                    continue
                for q, p in act:
                    if p == "X":
                        out.append(Op("H", [q]))
                    elif p == "Y":
                        out.append(Op("RX", [q], math.pi / 2))
                for (a, _), (b, _) in zip(act[:-1], act[1:]):
                    out.append(Op("CNOT", [a, b]))
                out.append(Op("RZ", [act[-1][0]], 2 * angle if not isinstance(angle, AffineParam) else angle * 2))
                for (a, _), (b, _) in reversed(list(zip(act[:-1], act[1:]))):
                    out.append(Op("CNOT", [a, b]))
                for q, p in act:
                    if p == "X":
                        out.append(Op("H", [q]))
                    elif p == "Y":
                        out.append(Op("RX", [q], -math.pi / 2))
        return out

    def to_job(self, job_type="SAMPLE", observable=None, nbshots=0, **_):
        if job_type not in ("SAMPLE", "OBS"):
            raise ValueError(job_type)
        if job_type == "OBS" and observable is None:
            raise ValueError("OBS job without observable")
        return Job(self, job_type, observable)


class Job:
    def __init__(self, circuit, job_type, observable):
        self.circuit, self.type, self.observable = circuit, job_type, observable


class _State:
    __slots__ = ("int", "nbqbits")

    def __init__(self, value, nbqbits):
        self.int, self.nbqbits = int(value), nbqbits

    def __str__(self):
        return "|" + format(self.int, f"0{self.nbqbits}b") + ">"


class Sample:
    __slots__ = ("state", "amplitude", "probability")

    def __init__(self, index, amplitude, nbqbits):
        self.state = _State(index, nbqbits)
        self.amplitude = complex(amplitude)
        self.probability = abs(amplitude) ** 2


class Result:
    """value of an observable job, or the samples of a state-vector job.  The samples of a large register are kept as
    the arrays the device listed (``indices`` ascending, ``amplitudes``); Sample objects are made when somebody iterates."""

    def __init__(self, value=None, samples=None, indices=None, amplitudes=None, nbqbits=None):
        self.value = value
        self.indices = indices
        self.amplitudes = amplitudes
        self._nbqbits = nbqbits
        self._samples = samples if samples is not None else ([] if indices is None else None)

    @property
    def raw_data(self):
        if self._samples is None:
            self._samples = [Sample(int(i), a, self._nbqbits) for i, a in zip(self.indices, self.amplitudes)]
        return self._samples

    def __iter__(self):
        if self._samples is None:   # one pass over a long list: no need to keep the objects
            return (Sample(int(i), a, self._nbqbits) for i, a in zip(self.indices, self.amplitudes))
        return iter(self._samples)

    def __len__(self):
        return len(self.indices) if self._samples is None else len(self._samples)


# ------------------------------------------------------------------------------------ execution
def lower_circuit(circuit, theta=None):
    """-> (hf_index, kind, payload): kind 'rotations' -> (xs, zs, coeff, phi0, pidx);
    kind 'gates' -> list of (name, qubits, scale, const, pidx).  Leading X gates / the ``init`` of a leading
    evolution become the HF basis state.  A circuit mixing literal gates and evolutions is lowered to
    gates only if every evolution is first expanded by the caller (not needed by OpenVQE's L1)."""
    n = circuit.nbqbits
    hf = 0
    pristine = True
    rot = []
    gates = []
    for kind, what, qubits in circuit.items:
        if kind == "gate":
            if what.name == "X" and pristine and not rot and not gates:
                hf ^= 1 << (n - 1 - qubits[0])
                continue
            pristine = False
            sc, co, pi = _affine(what.angle) if what.angle is not None else (0.0, 0.0, -1)
            gates.append((what.name, list(qubits), sc, co, pi))
        else:
            if what.init:
                if not (pristine and not rot and not gates):
                    for q in range(what.arity):
                        if (what.init >> (what.arity - 1 - q)) & 1:
                            gates.append(("X", [qubits[q]], 0.0, 0.0, -1))
                else:
                    for q in range(what.arity):
                        if (what.init >> (what.arity - 1 - q)) & 1:
                            hf ^= 1 << (n - 1 - qubits[q])
            pristine = False
            for pauli, qs, angle in what.rotations():
                x, z = pack_string(n, pauli, [qubits[q] for q in qs])
                sc, co, pi = _affine(angle)
                rot.append((x, z, sc, co, pi))
    if rot and gates:
        raise NotImplementedError("circuit mixes Pauli evolutions and literal gates")
    if gates:
        return hf, "gates", gates
    xs = np.array([r[0] for r in rot], np.uint64)
    zs = np.array([r[1] for r in rot], np.uint64)
    sc = np.array([r[2] for r in rot], np.float64)
    co = np.array([r[3] for r in rot], np.float64)
    pi = np.array([r[4] for r in rot], np.int32)
    return hf, "rotations", (xs, zs, sc, co, pi)


class HipQPU:
    """``get_default_qpu()`` stand-in: exact (nbshots=0) simulation on the MI355X backend.

    The reference rebuilds Program / Circuit / Job for every energy evaluation (ref:openvqe/ucc_family/get_energy_ucc.py:35-50).
    A circuit made of Pauli evolutions of the SAME operator objects as the previous one (what an optimiser loop submits)
    differs from it only in its angles: it is compiled once with symbolic angles and later submissions only pass the new
    angle vector — so the compiled-program paths of the backend (table fusion, sector tables at 18+ qubits) serve the
    unchanged reference code too.  Circuits of literal gates (the QUCCSD templates) are cached by their structure — gate
    names, qubits, quarter-turn angles — with every other rotation gate as a parameter of its own, so the second submission
    of a template only passes its angles (and the Clifford-frame / sector paths keep their tables)."""

    def __init__(self, device=None):
        self.device = device    # None: this rank's GPU (replicas.device()); registers too large for one device are partitioned
        self._sv = {}
        self._compiled = {}   # nbqbits -> (skeleton key, operator objects kept alive, n_params)
        self._observable = {}  # nbqbits -> (observable object uploaded last, its content fingerprint)

    def _backend(self, n):
        from .partitioned import make_backend
        if n not in self._sv:
            self._sv[n] = make_backend(n, self.device)
        return self._sv[n]

    @staticmethod
    def _fingerprint(op):
        """cheap content check of an operator object next to its identity: callers of the reference may mutate an operator
        or an observable in place between submissions (it rebuilds everything per submit); term count, constant, the
        coefficient sum and a sample of whole terms catch an edited / appended term at a cost far below the submission's"""
        terms = op.terms
        acc = sum([t.coeff for t in terms])                                   # one pass, ~20 us per 1000 terms
        step = max(1, len(terms) // 8)
        chars = tuple((t.op, tuple(t.qbits), t.coeff) for t in terms[::step])  # ... and a few whole terms
        return (len(terms), complex(getattr(op, "constant_coeff", 0.0)), acc, chars)

    @staticmethod
    def _skeleton(circuit):
        """-> (key, angles, symbolic circuit items) for a circuit of leading X gates + Pauli evolutions with concrete angles,
        else None"""
        key, angles, items = [], [], []
        for kind, what, qubits in circuit.items:
            if kind == "gate":
                if what.name != "X" or any(k[0] == "E" for k in key):
                    return None
                key.append(("X", tuple(qubits)))
                items.append((kind, what, qubits))
                continue
            ops = list(what.operators)
            thetas = list(what.thetas)[:len(ops)]          # zip truncation, as PauliEvolution.rotations
            ops = ops[:len(thetas)]
            if any(isinstance(t, AffineParam) for t in thetas):
                return None
            key.append(("E", tuple(id(op) for op in ops), tuple(HipQPU._fingerprint(op) for op in ops), what.init, tuple(qubits)))
            sym = [AffineParam(len(angles) + k) for k in range(len(thetas))]
            angles.extend(float(t) for t in thetas)
            items.append((kind, PauliEvolution(ops, what.init, sym, what.arity), qubits))
        return tuple(key), angles, items

    @staticmethod
    def _gate_skeleton(circuit):
        """-> (key, angles, symbolic circuit items) for a circuit of literal gates with concrete angles, else None.  The
        reference's gate templates (ref:openvqe/common_files/circuit.py) are rebuilt per evaluation with the optimiser's
        angles inside: the STRUCTURE (gate names, qubits, which rotation gates carry a quarter turn — the basis changes of
        the templates, kept concrete so that they stay Clifford gates) repeats, every other rotation gate becomes a
        parameter of its own."""
        key, angles, items = [], [], []
        for kind, what, qubits in circuit.items:
            if kind != "gate" or isinstance(what.angle, AffineParam):
                return None
            if what.angle is None:
                key.append((what.name, tuple(qubits)))
                items.append((kind, what, qubits))
                continue
            a = float(what.angle)
            quarter = a / (0.5 * math.pi)
            if quarter == round(quarter):       # (bitwise k pi/2, as the templates write them)
                key.append((what.name, tuple(qubits), a))
                items.append((kind, what, qubits))
            else:
                key.append((what.name, tuple(qubits), None))
                items.append((kind, Gate(what.name, what.arity, AffineParam(len(angles))), qubits))
                angles.append(a)
        return tuple(key), angles, items

    def _load(self, sv, circuit):
        """-> angle vector to evaluate the handle's program with"""
        n = circuit.nbqbits
        gk = self._gate_skeleton(circuit)
        if gk is not None and gk[1]:
            key, angles, items = gk
            cached = self._compiled.get(n)
            if cached is not None and cached[0] == key and cached[2] == len(angles):
                return np.asarray(angles, dtype=float)
            hf, kind, payload = lower_circuit(Circuit(n, items))
            if kind == "gates":
                sv.set_gate_program(payload, len(angles), hf)
                self._compiled[n] = (key, None, len(angles))
                return np.asarray(angles, dtype=float)
        sk = self._skeleton(circuit)
        if sk is not None:
            key, angles, items = sk
            cached = self._compiled.get(n)
            if cached is not None and cached[0] == key and cached[2] == len(angles):
                return np.asarray(angles, dtype=float)
            symbolic = Circuit(n, items)
            hf, kind, payload = lower_circuit(symbolic)
            if kind == "rotations":
                xs, zs, sc, co, pi = payload
                sv.set_rotation_program(xs, zs, sc, pi, len(angles), hf, phi0=co)
                keep = [what.operators for kind_, what, _ in items if kind_ != "gate"]
                self._compiled[n] = (key, keep, len(angles))
                return np.asarray(angles, dtype=float)
        self._compiled.pop(n, None)
        hf, kind, payload = lower_circuit(circuit)
        if kind == "gates":
            npar = 1 + max([g[4] for g in payload] + [-1])
            if npar:
                raise ValueError("circuit still has free parameters")
            sv.set_gate_program(payload, 0, hf)
        else:
            xs, zs, sc, co, pi = payload
            if len(pi) and pi.max() >= 0:
                raise ValueError("circuit still has free parameters")
            sv.set_rotation_program(xs, zs, sc, pi, 0, hf, phi0=co)
        return np.zeros(0)

    def submit(self, job):
        circ = job.circuit
        n = circ.nbqbits
        sv = self._backend(n)
        theta = self._load(sv, circ)
        if job.type == "OBS":
            seen = self._observable.get(n)
            fp = self._fingerprint(job.observable)
            if seen is None or seen[0] is not job.observable or seen[1] != fp:
                sv.set_hamiltonian(job.observable)
                self._observable[n] = (job.observable, fp)
            return Result(value=sv.energy(theta))
        sv.prepare_state(theta)
        listed = sv.get_support() if n >= 16 else None   # non-zero amplitudes listed on the device (ovqe_get_support)
        if listed is not None:
            return Result(indices=listed[0].astype(np.int64), amplitudes=listed[1], nbqbits=circ.nbqbits)
        psi = sv.get_state()
        nz = np.nonzero(psi)[0]
        return Result(samples=[Sample(int(i), psi[i], circ.nbqbits) for i in nz])


_default_qpu = None


def get_default_qpu():
    global _default_qpu
    if _default_qpu is None:
        _default_qpu = HipQPU()
    return _default_qpu


# ------------------------------------------------------------------------------------ sys.modules hook
def install(force=False):
    """Register these stand-ins as ``qat.lang.AQASM``, ``qat.qpus``, ``qat.core``,
    ``qat.fermion(.chemistry.ucc_deprecated)`` so reference modules import unchanged.
    Does nothing when a real myQLM is importable unless ``force``."""
    if not force:
        try:
            import qat  # noqa: F401
            if not getattr(qat, "__ovqe_shim__", False):
                return False
        except ImportError:
            pass
    me = sys.modules[__name__]

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__ovqe_shim__ = True
        sys.modules[name] = m
        return m

    gates = dict(X=X, Y=Y, Z=Z, H=H, CNOT=CNOT, RX=RX, RY=RY, RZ=RZ)
    qat = mod("qat")
    qat.lang = mod("qat.lang")
    qat.lang.AQASM = mod("qat.lang.AQASM", Program=Program, QRoutine=QRoutine, **gates)
    qat.lang.AQASM.gates = mod("qat.lang.AQASM.gates", Gate=Gate, **gates)
    qat.qpus = mod("qat.qpus", get_default_qpu=get_default_qpu)
    qat.core = mod("qat.core", Term=Term, Observable=Observable, Circuit=Circuit, Job=Job, Result=Result)
    from . import fermionic as _fermionic
    from . import qat_chem as _chem

    qat.fermion = mod("qat.fermion", Hamiltonian=Hamiltonian, SpinHamiltonian=SpinHamiltonian,
                      FermionHamiltonian=_fermionic.FermionHamiltonian,
                      ElectronicStructureHamiltonian=_chem.ElectronicStructureHamiltonian)
    qat.fermion.transforms = mod("qat.fermion.transforms", transform_to_jw_basis=_chem.transform_to_jw_basis,
                                 transform_to_bk_basis=_chem.transform_to_bk_basis,
                                 transform_to_parity_basis=_chem.transform_to_parity_basis,
                                 get_jw_code=_chem.get_jw_code, get_bk_code=_chem.get_bk_code,
                                 get_parity_code=_chem.get_parity_code, recode_integer=_chem.recode_integer)
    qat.fermion.chemistry = mod("qat.fermion.chemistry")
    qat.fermion.chemistry.pyscf_tools = mod("qat.fermion.chemistry.pyscf_tools",
                                            perform_pyscf_computation=_chem.perform_pyscf_computation)
    qat.fermion.chemistry.ucc = mod("qat.fermion.chemistry.ucc", convert_to_h_integrals=_chem.convert_to_h_integrals,
                                    transform_integrals_to_new_basis=_chem.transform_integrals_to_new_basis)
    qat.fermion.chemistry.ucc_deprecated = mod("qat.fermion.chemistry.ucc_deprecated",
                                               build_ucc_ansatz=build_ucc_ansatz,
                                               get_cluster_ops_and_init_guess=_chem.get_cluster_ops_and_init_guess,
                                               get_active_space_hamiltonian=_chem.get_active_space_hamiltonian)
    qat.fermion.hamiltonians = mod("qat.fermion.hamiltonians", Hamiltonian=Hamiltonian,
                                   SpinHamiltonian=SpinHamiltonian, FermionHamiltonian=_fermionic.FermionHamiltonian)
    del me
    return True
