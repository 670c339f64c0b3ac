"""Second-quantised operators on the host: the objects the reference's pool generators build from ``qat.core.Term`` +
``qat.fermion.FermionHamiltonian`` (ref:openvqe/common_files/generator_excitations.py:4-6) and hand to
``transform_to_jw_basis`` (ref:…generator_excitations.py:16-36, ref:openvqe/common_files/qubit_pool.py:43-47).

A term is ``coeff * o_0 o_1 …`` with ``op`` a string over {"C" = creation, "c" = annihilation} and ``qbits`` the
spin-orbital of each ladder operator (spin orbitals interleaved: even = alpha, odd = beta; spin orbital p <-> qubit p
under Jordan-Wigner).  SURVEY.md §8f rows 1-2: front-end data producers, CPU, once per job.

Normal ordering is a restatement of what ref:openvqe/common_files/fermion_util.py:5-114 computes — creation operators
left of annihilation operators, each group by increasing orbital, signs and contraction terms from the
anticommutation relations, terms with a repeated operator dropped — by a different procedure: the ladder operators
are inserted one at a time into an already ordered product (``_insert``), instead of searching the string for
out-of-order neighbours.
"""
from __future__ import annotations

from .operators import Hamiltonian, Term


class FermionHamiltonian:
    """Sum of ladder-operator products.  Equal (op, qbits) terms are merged in first-appearance order (coefficients
    that cancel stay as zero-coefficient terms — the reference's pool sizes, ref:tests/test_main_fermionic_adapt.py:11,
    count such operators: SURVEY.md §8)."""

    def __init__(self, nqbits, terms=(), constant_coeff=0.0, do_clean_up=True):
        self.nbqbits = int(nqbits)
        self.constant_coeff = constant_coeff
        self.terms = [Term(t.coeff, t.op, list(t.qbits)) for t in terms]
        for t in self.terms:
            if set(t.op) - {"C", "c"}:
                raise ValueError("fermionic term over an alphabet other than 'C'/'c'")
            if t.qbits and max(t.qbits) >= self.nbqbits:
                raise ValueError("ladder operator outside the register")
        if do_clean_up:
            merged, order = {}, []
            for t in self.terms:
                key = (t.op, tuple(t.qbits))
                if not key[0]:
                    self.constant_coeff = self.constant_coeff + t.coeff
                elif key in merged:
                    merged[key] = merged[key] + t.coeff
                else:
                    merged[key] = t.coeff
                    order.append(key)
            self.terms = [Term(merged[k], k[0], list(k[1])) for k in order]

    def copy(self):
        return FermionHamiltonian(self.nbqbits, self.terms, self.constant_coeff, do_clean_up=False)

    def __mul__(self, scalar):
        return FermionHamiltonian(self.nbqbits, [Term(t.coeff * scalar, t.op, t.qbits) for t in self.terms],
                                  self.constant_coeff * scalar, do_clean_up=False)

    __rmul__ = __mul__

    def __truediv__(self, scalar):
        return self * (1.0 / scalar)

    def __neg__(self):
        return self * (-1.0)

    def __add__(self, other):
        if isinstance(other, FermionHamiltonian):
            if other.nbqbits != self.nbqbits:
                raise ValueError("register size mismatch")
            return FermionHamiltonian(self.nbqbits, self.terms + other.terms, self.constant_coeff + other.constant_coeff)
        return FermionHamiltonian(self.nbqbits, self.terms, self.constant_coeff + other, do_clean_up=False)

    __radd__ = __add__  # lets ``0 + op`` start a sum (ref:…generator_excitations.py:232-236)

    def __sub__(self, other):
        return self + other * (-1.0)

    def dag(self):
        flip = {"C": "c", "c": "C"}
        return FermionHamiltonian(self.nbqbits, [Term(complex(t.coeff).conjugate(), "".join(flip[o] for o in reversed(t.op)),
                                                      list(reversed(t.qbits))) for t in self.terms],
                                  complex(self.constant_coeff).conjugate(), do_clean_up=False)

    def get_matrix(self, sparse=False):
        """2^n x 2^n matrix of the Jordan-Wigner image (what the reference's sparse factory asks of every pool operator,
        ref:openvqe/common_files/molecule_factory_with_sparse.py:601-617)"""
        return transform_to_jw_basis(self).get_matrix(sparse=sparse)

    def to_spin(self, transform="JW"):
        return spin_operator(self, transform)

    def __repr__(self):
        return " + ".join([f"{self.constant_coeff}"] + [f"{t.coeff}*{t.op}{t.qbits}" for t in self.terms])


def _no_transform(name):
    raise NotImplementedError(f"transform '{name}': 'JW', 'Bravyi-Kitaev' and 'parity_basis' are restated")


# ---------------------------------------------------------------------------------------------- normal ordering
def _insert(cre, ann, dagger, p):
    """Right-multiply the ordered product  C[cre...] c[ann...]  (both strictly increasing) by one ladder operator
    -> list of (sign, cre, ann) of ordered products; contraction (delta) terms come first, like the reference's
    neighbour permutation (ref:…fermion_util.py:24-27)."""
    if not dagger:
        if p in ann:
            return []                                   # c_p c_p = 0
        k = sum(1 for q in ann if q > p)                # hops to the left over larger annihilators
        pos = len(ann) - k
        return [((-1) ** k, cre, ann[:pos] + (p,) + ann[pos:])]
    out = []
    # move C_p to the left through every annihilator: c_q C_p = delta_qp - C_p c_q
    sign = 1
    for j in range(len(ann) - 1, -1, -1):
        if ann[j] == p:
            rest = ann[:j] + ann[j + 1:]
            out.append((sign, cre, rest))
        sign = -sign
    if p not in cre:
        k = sum(1 for q in cre if q > p)
        pos = len(cre) - k
        out.append((sign * (-1) ** k, cre[:pos] + (p,) + cre[pos:], ann))
    return out


def normal_ordered_terms(term):
    """One fermionic Term -> list of Terms "C…Cc…c" with increasing orbitals inside each group (what
    ``order_fermionic_term`` returns, ref:…fermion_util.py:98-114).  Products that reduce to a pure number are not
    representable in that form and are dropped there too (the reference's ``pauli_op.index('c')`` requires a 'c')."""
    states = [(term.coeff, (), ())]
    for o, p in zip(term.op, term.qbits):
        nxt = []
        for coeff, cre, ann in states:
            for sign, c2, a2 in _insert(cre, ann, o == "C", p):
                nxt.append((coeff * sign, c2, a2))
        states = nxt
    return [Term(coeff, "C" * len(cre) + "c" * len(ann), list(cre) + list(ann)) for coeff, cre, ann in states
            if ann]


def normal_ordered(nqbits, terms):
    """FermionHamiltonian of the normal-ordered images of ``terms`` (merged)"""
    out = []
    for t in terms:
        out.extend(normal_ordered_terms(t))
    return FermionHamiltonian(nqbits, out)


# ---------------------------------------------------------------------------------------------- Jordan-Wigner
def transform_to_jw_basis(op):
    """FermionHamiltonian -> spin Hamiltonian, a_p = Z_0…Z_{p-1} (X_p + iY_p)/2 (``qat.fermion.transforms``; call sites
    ref:…generator_excitations.py:17-30, ref:…qubit_pool.py:43-47).  Strings whose coefficients cancel are dropped, so
    an identically-zero operator has an empty term list here.

    ORDER of the Pauli strings (it decides "the first string on each qubit support" of the derived qubit pools,
    ref:…qubit_pool.py:233-274, and the Trotter order inside a generator): the stored second run of
    ref:notebooks/demo_puccgsd.ipynb (E(0.01) = -1.1286907548863794 and its 18 forward differences, reproduced to 3e-10 /
    7e-6 in tests/test_reference_traces.py) fixes the first odd-Y strings of the sUPCCGSD operators as "YX" on (q < p) and
    "YXYY" on (p, p+1, q, q+1), alpha support before beta support.  Expanding every ladder product X-before-Y with the
    last operator varying fastest (as ``fermion.jw_product`` does) reproduces all of it when the fermionic terms are
    expanded in increasing order of their index lists; expanding them in construction order gives "YYYX", in reversed
    order the beta supports first, and no Y-before-X variant gives the strings at all.  myQLM's own ordering is not in
    the reference tree, so this is a fitted convention, stated as such."""
    from . import fermion
    total = {}
    for t in sorted(op.terms, key=lambda t: t.qbits):
        fermion.psum_iadd(total, fermion.jw_product([(p, o == "C") for o, p in zip(t.op, t.qbits)]), t.coeff)
    return fermion.psum_to_hamiltonian(op.nbqbits, total, op.constant_coeff, tol=1e-13)


def transform_to_encoding(op, transform):
    """FermionHamiltonian -> spin Hamiltonian under "Bravyi-Kitaev" / "parity_basis" (``transform_to_bk_basis`` /
    ``transform_to_parity_basis`` of myQLM; linear encodings of fermion.encoding_matrix).  Isospectral with the JW image;
    myQLM's own index conventions for these two encodings are not pinned by anything the reference stores."""
    from . import fermion
    total = {}
    for t in sorted(op.terms, key=lambda t: t.qbits):
        fermion.psum_iadd(total, fermion.encoded_product(op.nbqbits, [(p, o == "C") for o, p in zip(t.op, t.qbits)],
                                                         transform), t.coeff)
    return fermion.psum_to_hamiltonian(op.nbqbits, total, op.constant_coeff, tol=1e-13)


def spin_operator(op, transform="JW"):
    if transform == "JW":
        return transform_to_jw_basis(op)
    if transform in ("Bravyi-Kitaev", "parity_basis"):
        return transform_to_encoding(op, transform)
    _no_transform(transform)


__all__ = ["FermionHamiltonian", "Hamiltonian", "Term", "normal_ordered_terms", "normal_ordered", "transform_to_jw_basis",
           "spin_operator"]
