/* TEST INFRASTRUCTURE ONLY — plain-C CPU restatement of the OpenVQE statevector hot path.
 *
 * Not part of the product: only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py load this library, as the checker / reported baseline.
 *
 * Two restatements of one energy evaluation
 *   E(theta) = <HF| U^+ H U |HF>,  U = prod_k prod_j exp(-i theta_k c_kj P_kj)
 * (ref:openvqe/ucc_family/get_energy_ucc.py:35-50; clones
 *  ref:openvqe/adapt/fermionic_adapt_vqe.py:126-162, ref:openvqe/adapt/qubit_adapt_vqe.py:271-307):
 *
 *  C1 "gate level"  — what one reference evaluation does algorithmically inside myQLM
 *     (build_ucc_ansatz one-step Trotter slice synthesised as basis change (H for X,
 *     RX(pi/2) for Y) + CNOT staircase + RZ(2 phi) + uncompute; observable evaluated term by
 *     term; one full-state sweep per gate / per term).  SURVEY.md §3.1, BASELINE.md §3 "C1".
 *  C2 "fused"       — one sweep per Pauli rotation with (x,z) bit masks, x-mask-grouped
 *     expectation; OpenMP over host cores.  Same algorithm as the HIP kernels.
 *
 * Conventions: reference qubit q <-> basis-index bit (n-1-q)
 * (ref:openvqe/ucc_family/get_energy_qucc.py:40-45,
 *  ref:openvqe/common_files/molecule_factory_with_sparse.py:622-642); a Pauli string is two
 * index-space masks (x,z): I=(0,0) X=(1,0) Z=(0,1) Y=(1,1), P = i^{|x&z|} X^x Z^z.
 * Gates: RX(a)=exp(-iaX/2), RY(a)=exp(-iaY/2), RZ(a)=diag(e^{-ia/2},e^{ia/2}), CNOT(control,target)
 * (gate set of ref:openvqe/common_files/circuit.py:1).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

static inline int parity64(uint64_t v) { return __builtin_parityll(v); }

/* 1: sweeps are OpenMP-parallel (large n); 0: sweeps run serially (batch mode: one thread per evaluation) */
static int g_par = 1;

void orc_init_basis(cplx *psi, int n, uint64_t index) {
    uint64_t dim = 1ull << n;
    memset(psi, 0, dim * sizeof(cplx));
    psi[index] = 1.0;
}

/* ---------------------------------------------------------------- C2: fused mask sweeps */
void orc_pauli_rotation(cplx *psi, int n, uint64_t x, uint64_t z, double phi) {
    const uint64_t dim = 1ull << n;
    const double c = cos(phi), s = sin(phi);
    const int ny = __builtin_popcountll(x & z) & 3;
    static const cplx ipow[4] = {1.0, I, -1.0, -I};
    const cplx mis = -I * s * ipow[ny]; /* -i sin(phi) i^ny */
    if (x == 0) {
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static)
        for (uint64_t i = 0; i < dim; ++i) {
            double sg = parity64(i & z) ? -1.0 : 1.0;
            psi[i] = (c + mis * sg) * psi[i];
        }
        return;
    }
    const int p = 63 - __builtin_clzll(x); /* pivot: highest x bit */
    const uint64_t low = (1ull << p) - 1;
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static)
    for (uint64_t k = 0; k < dim / 2; ++k) {
        uint64_t i = ((k & ~low) << 1) | (k & low); /* bit p clear */
        uint64_t j = i ^ x;
        double si = parity64(j & z) ? -1.0 : 1.0; /* sign of <i|P|j> */
        double sj = parity64(i & z) ? -1.0 : 1.0; /* sign of <j|P|i> */
        cplx a = psi[i], b = psi[j];
        psi[i] = c * a + mis * si * b;
        psi[j] = c * b + mis * sj * a;
    }
}

/* Re sum_t coeff_t <psi|P_t|psi>, one sweep per term (term by term, as myQLM's OBS job) */
double orc_expectation_termwise(const cplx *psi, int n, int64_t T, const uint64_t *xs, const uint64_t *zs,
                                const double *coeffs) {
    const uint64_t dim = 1ull << n;
    static const cplx ipow[4] = {1.0, I, -1.0, -I};
    double total = 0.0;
    for (int64_t t = 0; t < T; ++t) {
        const uint64_t x = xs[t], z = zs[t];
        const cplx ph = ipow[__builtin_popcountll(x & z) & 3];
        double re = 0.0, im = 0.0;
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static) reduction(+ : re, im)
        for (uint64_t i = 0; i < dim; ++i) {
            uint64_t j = i ^ x;
            double sg = parity64(j & z) ? -1.0 : 1.0;
            cplx v = conj(psi[i]) * psi[j] * sg;
            re += creal(v);
            im += cimag(v);
        }
        total += coeffs[t] * creal(ph * (re + I * im));
    }
    return total;
}

/* same value, terms grouped by x mask: one sweep per distinct x (terms must be sorted by x) */
double orc_expectation_grouped(const cplx *psi, int n, int64_t T, const uint64_t *xs, const uint64_t *zs,
                               const double *coeffs) {
    const uint64_t dim = 1ull << n;
    double total = 0.0;
    int64_t t0 = 0;
    while (t0 < T) {
        int64_t t1 = t0;
        while (t1 < T && xs[t1] == xs[t0]) ++t1;
        const uint64_t x = xs[t0];
        double acc = 0.0;
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static) reduction(+ : acc)
        for (uint64_t i = 0; i < dim; ++i) {
            uint64_t j = i ^ x;
            cplx v = conj(psi[i]) * psi[j];
            double dre = 0.0, dim_ = 0.0; /* D(i) = sum_t c_t i^ny (-1)^{|j&z|} */
            for (int64_t t = t0; t < t1; ++t) {
                double sg = parity64(j & zs[t]) ? -coeffs[t] : coeffs[t];
                switch (__builtin_popcountll(x & zs[t]) & 3) {
                case 0: dre += sg; break;
                case 1: dim_ += sg; break;
                case 2: dre -= sg; break;
                default: dim_ -= sg; break;
                }
            }
            acc += dre * creal(v) - dim_ * cimag(v);
        }
        total += acc;
        t0 = t1;
    }
    return total;
}

/* ---------------------------------------------------------------- gate level primitives */
void orc_gate_1q(cplx *psi, int n, int bit, const double *m /* row-major 2x2 as re,im pairs */) {
    const uint64_t dim = 1ull << n, stride = 1ull << bit, low = stride - 1;
    const cplx m00 = m[0] + I * m[1], m01 = m[2] + I * m[3], m10 = m[4] + I * m[5], m11 = m[6] + I * m[7];
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static)
    for (uint64_t k = 0; k < dim / 2; ++k) {
        uint64_t i = ((k & ~low) << 1) | (k & low), j = i | stride;
        cplx a = psi[i], b = psi[j];
        psi[i] = m00 * a + m01 * b;
        psi[j] = m10 * a + m11 * b;
    }
}

void orc_gate_cnot(cplx *psi, int n, int cbit, int tbit) {
    const uint64_t dim = 1ull << n, cm = 1ull << cbit, tm = 1ull << tbit;
#pragma omp parallel for if (g_par && dim >= (1ull << 18)) schedule(static)
    for (uint64_t i = 0; i < dim; ++i) {
        if ((i & cm) && !(i & tm)) {
            cplx a = psi[i];
            psi[i] = psi[i | tm];
            psi[i | tm] = a;
        }
    }
}

static void mat_h(double *m) {
    const double r = 0.70710678118654752440;
    double v[8] = {r, 0, r, 0, r, 0, -r, 0};
    memcpy(m, v, sizeof v);
}
static void mat_rx(double a, double *m) {
    double c = cos(a / 2), s = sin(a / 2);
    double v[8] = {c, 0, 0, -s, 0, -s, c, 0};
    memcpy(m, v, sizeof v);
}
static void mat_ry(double a, double *m) {
    double c = cos(a / 2), s = sin(a / 2);
    double v[8] = {c, 0, -s, 0, s, 0, c, 0};
    memcpy(m, v, sizeof v);
}
static void mat_rz(double a, double *m) {
    double c = cos(a / 2), s = sin(a / 2);
    double v[8] = {c, -s, 0, 0, 0, 0, c, s};
    memcpy(m, v, sizeof v);
}

/* opcode: 0 X, 1 H, 2 RX, 3 RY, 4 RZ, 5 CNOT(q0 control, q1 target); q are index BITS */
void orc_apply_gate(cplx *psi, int n, int opcode, int b0, int b1, double angle) {
    double m[8];
    switch (opcode) {
    case 0: { double v[8] = {0, 0, 1, 0, 1, 0, 0, 0}; memcpy(m, v, sizeof v); orc_gate_1q(psi, n, b0, m); break; }
    case 1: mat_h(m); orc_gate_1q(psi, n, b0, m); break;
    case 2: mat_rx(angle, m); orc_gate_1q(psi, n, b0, m); break;
    case 3: mat_ry(angle, m); orc_gate_1q(psi, n, b0, m); break;
    case 4: mat_rz(angle, m); orc_gate_1q(psi, n, b0, m); break;
    case 5: orc_gate_cnot(psi, n, b0, b1); break;
    default: break;
    }
}

/* C1: exp(-i phi P) as the CNOT-staircase circuit, one sweep per gate.
 * Qubits are visited in ascending reference-qubit order = descending index bit. */
void orc_pauli_rotation_gates(cplx *psi, int n, uint64_t x, uint64_t z, double phi) {
    int bits[64], w = 0;
    double m[8];
    for (int b = n - 1; b >= 0; --b)
        if (((x | z) >> b) & 1) bits[w++] = b;
    if (w == 0) { /* exp(-i phi I): global phase */
        const uint64_t dim = 1ull << n;
        cplx g = cos(phi) - I * sin(phi);
        for (uint64_t i = 0; i < dim; ++i) psi[i] *= g;
        return;
    }
    for (int k = 0; k < w; ++k) {
        int b = bits[k];
        int isx = (x >> b) & 1, isz = (z >> b) & 1;
        if (isx && !isz) { mat_h(m); orc_gate_1q(psi, n, b, m); }
        else if (isx && isz) { mat_rx(M_PI / 2, m); orc_gate_1q(psi, n, b, m); }
    }
    for (int k = 0; k + 1 < w; ++k) orc_gate_cnot(psi, n, bits[k], bits[k + 1]);
    mat_rz(2.0 * phi, m);
    orc_gate_1q(psi, n, bits[w - 1], m);
    for (int k = w - 2; k >= 0; --k) orc_gate_cnot(psi, n, bits[k], bits[k + 1]);
    for (int k = 0; k < w; ++k) {
        int b = bits[k];
        int isx = (x >> b) & 1, isz = (z >> b) & 1;
        if (isx && !isz) { mat_h(m); orc_gate_1q(psi, n, b, m); }
        else if (isx && isz) { mat_rx(-M_PI / 2, m); orc_gate_1q(psi, n, b, m); }
    }
}

/* One full energy evaluation.
 *   rotations r = 0..R-1 applied in order: phi_r = theta[pidx[r]] * rcoef[r]
 *   mode 0 = C2 fused + grouped expectation (H terms must be sorted by x),
 *   mode 1 = C1 gate level + term-wise expectation.
 * psi is caller-provided scratch of 2^n amplitudes (left holding the final state). */
double orc_ucc_energy(cplx *psi, int n, uint64_t hf_index, int64_t R, const uint64_t *rx, const uint64_t *rz,
                      const double *rcoef, const int32_t *pidx, const double *theta, int64_t T, const uint64_t *hx,
                      const uint64_t *hz, const double *hc, double constant, int mode) {
    orc_init_basis(psi, n, hf_index);
    for (int64_t r = 0; r < R; ++r) {
        double phi = theta[pidx[r]] * rcoef[r];
        if (mode == 0) orc_pauli_rotation(psi, n, rx[r], rz[r], phi);
        else orc_pauli_rotation_gates(psi, n, rx[r], rz[r], phi);
    }
    double e = mode == 0 ? orc_expectation_grouped(psi, n, T, hx, hz, hc) : orc_expectation_termwise(psi, n, T, hx, hz, hc);
    return e + constant;
}

/* Gate-program evaluation (QUCCSD templates, ref:openvqe/common_files/circuit.py:13-106):
 * gate g: opcode[g], index bits b0[g], b1[g], angle = ascale[g]*theta[gpidx[g]] + aconst[g] (gpidx<0: constant). */
double orc_gate_energy(cplx *psi, int n, uint64_t hf_index, int64_t G, const int32_t *opcode, const int32_t *b0,
                       const int32_t *b1, const double *ascale, const double *aconst, const int32_t *gpidx,
                       const double *theta, int64_t T, const uint64_t *hx, const uint64_t *hz, const double *hc,
                       double constant) {
    orc_init_basis(psi, n, hf_index);
    for (int64_t g = 0; g < G; ++g) {
        double a = aconst[g] + (gpidx[g] >= 0 ? ascale[g] * theta[gpidx[g]] : 0.0);
        orc_apply_gate(psi, n, opcode[g], b0[g], b1[g], a);
    }
    return orc_expectation_termwise(psi, n, T, hx, hz, hc) + constant;
}

/* B independent evaluations, one host thread each (the fastest CPU arrangement when the state is
 * small: sweeps run single-threaded inside a thread, evaluations run in parallel across cores).
 * scratch: nthreads * 2^n amplitudes. */
void orc_ucc_energy_batch(cplx *scratch, int nthreads, int n, uint64_t hf_index, int64_t R, const uint64_t *rx,
                          const uint64_t *rz, const double *rcoef, const int32_t *pidx, int64_t B, int K,
                          const double *thetas, int64_t T, const uint64_t *hx, const uint64_t *hz, const double *hc,
                          double constant, int mode, double *energies) {
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
#endif
    g_par = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int64_t b = 0; b < B; ++b) {
#ifdef _OPENMP
        int tid = omp_get_thread_num();
#else
        int tid = 0;
#endif
        energies[b] = orc_ucc_energy(scratch + ((uint64_t)tid << n), n, hf_index, R, rx, rz, rcoef, pidx,
                                     thetas + b * K, T, hx, hz, hc, constant, mode);
    }
    g_par = 1;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_threads(int t) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}
