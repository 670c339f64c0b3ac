"""Gate-level QUCCSD / qubit-excitation templates and the gate counter — same names, arguments and
emitted gate sequences as ref:openvqe/common_files/circuit.py:13-205 (Yordanov's "efficient"
excitation circuits as coded there, including the ``RY(-2*theta)`` of the double template, SURVEY.md
§8a row a4: the backend executes the literal gate list).  ``q`` is a register (list of qubit
handles), ``qc`` anything with ``apply(gate, *qubits)``; ``theta`` may be a float or a traced
``AffineParam``.
"""
from math import pi

from ..qat_compat import CNOT, RY, RZ, H


def _ladder_up(q, qc, lo, hi):
    """CNOT chain q[lo+1]->q[lo+2]->...->q[hi-1] (parity of the qubits strictly between lo and hi)."""
    for i in range(lo + 1, hi - 1):
        qc.apply(CNOT, q[i], q[i + 1])


def _ladder_down(q, qc, lo, hi):
    for i in range(max(0, hi - lo - 2)):
        qc.apply(CNOT, q[hi - 2 - i], q[hi - 1 - i])


def _single_core(q, qc, a, b, theta):
    """exchange-type rotation between qubits a and b (controlled-RY construction)"""
    qc.apply(RZ(pi / 2), q[a])
    qc.apply(RY(-pi / 2), q[b])
    qc.apply(RZ(-pi / 2), q[b])
    qc.apply(CNOT, q[a], q[b])
    qc.apply(RY(theta), q[a])
    qc.apply(RZ(-pi / 2), q[b])
    qc.apply(CNOT, q[a], q[b])
    qc.apply(RY(-theta), q[a])
    qc.apply(H, q[b])
    qc.apply(CNOT, q[a], q[b])


def _double_core(q, qc, e, theta):
    a, b, c, d = e
    qc.apply(CNOT, q[a], q[c])
    seq = [("RY", +1), ("H", b), ("CX", b), ("RY", -1), ("H", d), ("CX", d), ("RY", +1), ("CX", b), ("RY", -1),
           ("H", c), ("CX", c), ("RY", +1), ("CX", b), ("RY", -1), ("CX", d), ("RY", +1), ("H", d), ("CX", b),
           ("RY", -2), ("H", b), ("CX", c), ("H", c)]
    for kind, arg in seq:
        if kind == "RY":
            qc.apply(RY(arg * theta), q[a])
        elif kind == "H":
            qc.apply(H, q[arg])
        else:
            qc.apply(CNOT, q[a], q[arg])
    qc.apply(CNOT, q[a], q[c])


def circuit_opt_simple(q, qc, exci, theta):
    """single fermionic-evolution template (circuit.py:13-38)"""
    _ladder_up(q, qc, exci[0], exci[1])
    _single_core(q, qc, exci[0], exci[1], theta)
    _ladder_down(q, qc, exci[0], exci[1])
    return qc


def circuit_opt_double(q, qc, exci, theta):
    """double fermionic-evolution template (circuit.py:40-93)"""
    qc.apply(CNOT, q[exci[0]], q[exci[1]])
    qc.apply(CNOT, q[exci[2]], q[exci[3]])
    _ladder_up(q, qc, exci[0], exci[1])
    _ladder_up(q, qc, exci[2], exci[3])
    _double_core(q, qc, exci, theta)
    _ladder_down(q, qc, exci[0], exci[1])
    _ladder_down(q, qc, exci[2], exci[3])
    qc.apply(CNOT, q[exci[0]], q[exci[1]])
    qc.apply(CNOT, q[exci[2]], q[exci[3]])
    return qc


def efficient_fermionic_ansatz(q, qc, list_exci, list_theta):
    """2-index lists get the single template, 4-index lists the double one (circuit.py:95-106)"""
    for i in range(len(list_exci)):
        if len(list_exci[i]) == 4:
            circuit_opt_double(q, qc, list_exci[i], list_theta[i])
        else:
            circuit_opt_simple(q, qc, list_exci[i], list_theta[i])
    return qc


def single_qubit_evo(q, qc, exci, theta):
    """single qubit-evolution template (circuit.py:108-127)"""
    _single_core(q, qc, exci[0], exci[1], theta)
    return qc


def double_qubit_evo(q, qc, exci, theta):
    """double qubit-evolution template (circuit.py:129-170)"""
    qc.apply(CNOT, q[exci[0]], q[exci[1]])
    qc.apply(CNOT, q[exci[2]], q[exci[3]])
    _double_core(q, qc, exci, theta)
    qc.apply(CNOT, q[exci[0]], q[exci[1]])
    qc.apply(CNOT, q[exci[2]], q[exci[3]])
    return qc


def efficient_qubit_ansatz(q, qc, list_exci, list_theta):
    for i in range(len(list_exci)):
        if len(list_exci[i]) == 4:
            double_qubit_evo(q, qc, list_exci[i], list_theta[i])
        else:
            single_qubit_evo(q, qc, list_exci[i], list_theta[i])
    return qc


def quccsd_gate_list(n_spatial, n_occ_spatial, stride=1, excitations=None):
    """literal gate list [(name, qubits, angle_scale, angle_const, param_index)] of the fermionic templates above on
    every ``stride``-th UCCSD excitation of n_spatial orbitals / n_occ_spatial occupied — or on the explicit index lists
    ``excitations`` (e.g. ``[op.terms[0].qbits for op in cluster_ops]``, what ref:openvqe/ucc_family/get_energy_qucc.py:46-49
    extracts) — traced with symbolic angles; the input of ``Statevector.set_gate_program`` -> (gates, n_params, hf_integer)"""
    from .. import fermion
    from ..qat_compat import AffineParam, Program, lower_circuit
    if excitations is None:
        singles, doubles = fermion.uccsd_excitations(n_spatial, n_occ_spatial)
        excitations = [[i, a] for i, a in singles] + [[i, j, a, b] for i, j, a, b in doubles]
    exci = [list(e) for e in excitations][::stride]
    prog = Program()
    reg = prog.qalloc(2 * n_spatial)
    efficient_fermionic_ansatz(reg, prog, exci, [AffineParam(k) for k in range(len(exci))])
    _, kind, gates = lower_circuit(prog.to_circ())
    assert kind == "gates"
    return gates, len(exci), fermion.hf_integer(2 * n_spatial, 2 * n_occ_spatial)


def count(gate, mylist):
    """number of instructions whose text contains gate='<GATE>' (circuit.py:186-205); lower-case
    names are upper-cased first."""
    gate = str(gate)
    if gate == gate.lower():
        gate = gate.upper()
    fast = getattr(mylist, "gate_count", None)   # qat_compat.OpList: one pass over the gate names (same answer: an instruction's
    if fast is not None:                          # text carries exactly one gate='NAME')
        n = fast(gate)
        if n is not None:
            return n
    needle = "gate='{}'".format(gate)
    return sum(1 for op in mylist if needle in str(op))
