// sv_small.hpp — whole-circuit, parameter-batched energy kernel for small registers (n <= 16).
//
// At n = 12..14 (LiH / H2O) the state is 64..256 KiB: one sweep is far too short to amortise a
// launch, so one workgroup owns one statevector for the WHOLE evaluation: |HF> init, every fused
// Pauli-rotation pass / literal gate of the ansatz, and <psi|H|psi> — one launch for B parameter
// vectors (a finite-difference gradient is B = K+1).
//
// Residency: the state lives in LDS whenever it fits 128 KiB of the CU's 160 KiB.
//   * REAL mode: if every rotation string has an odd number of Y (-i phi P is then a real
//     antisymmetric matrix — every UCC / ADAPT generator) and only X/H/CNOT/RY gates occur, the
//     state stays real: 8 B/amplitude, so n <= 14 fits LDS and the arithmetic halves.  Hamiltonian
//     strings with odd #Y have zero expectation on a real state and are dropped by the host.
//   * complex mode: 16 B/amplitude, LDS for n <= 13, else a per-workgroup global slice (L2/MALL).
// cos/sin of every rotation angle are tabulated in LDS once per segment, so a sweep is pure FMA
// work on LDS operands.
//
// <psi|H|psi>: per x-group one pass over the pairs k (i = insert_zero(k, pivot), j = i ^ x):
//   E_g = 2 Re sum_k D(k) conj(a_i) a_j,  D(k) = sum_t c_t (-1)^{parity(k & z_t)}   (Hermitian pair trick;
// coefficient already carries i^ny and the parity of x & z).  Each thread owns 8 pair indices that differ
// only in 3 index bits, takes their 8-point Walsh-Hadamard transform in registers, and then needs ONE
// sign evaluation + FMA per Hamiltonian term (terms bucketed by those 3 bits of z on the host) instead
// of one per term per pair.
#pragma once
#include "sv_kernels.hpp"

namespace ovqe {

enum SmallOpKind : int32_t { OP_PAIR = 0, OP_DIAG = 1, OP_X = 2, OP_H = 3, OP_CNOT = 4 };

struct SmallOp {
    uint64_t x;       // OP_PAIR: x mask;  OP_X/OP_H: 1<<bit;  OP_CNOT: 1<<target
    int32_t kind;
    int32_t first;    // first rotation entry (OP_PAIR / OP_DIAG);  OP_CNOT: control bit
    int32_t count;    // rotations in the fused run;                OP_CNOT: target bit
    int32_t pivot;    // highest x bit (OP_PAIR); bit (OP_X/OP_H)
};

struct SmallRot {
    uint64_t z;
    double coeff;   // phi = coeff * theta[pidx] + phi0
    double phi0;
    int32_t pidx;   // < 0: constant angle
    int32_t ny;     // popcount(x & z)
};

// segment = run of ops whose rotation entries fit the LDS cos/sin table
struct SmallSeg {
    int32_t op0, op1, rot0, rot1;
};

// expectation tables in PAIR-INDEX space (k): built by the host for a given thread count (lbits)
struct ExpTerm {
    uint32_t zk;   // z mask with the pivot bit removed (diag group: z itself)
    uint32_t pad;
    double cr, ci; // coefficient * i^ny * (-1)^{parity(x & z)}
};
struct ExpGroup {
    uint32_t x;        // 0 for the diagonal group
    int32_t pivot;
    int32_t t0;        // first term
    int32_t off[9];    // bucket b (bits [lbits, lbits+3) of zk == b) = terms [t0+off[b], t0+off[b+1])
};

struct SmallArgs {
    int n;
    int64_t B;
    const double *theta;  // B x K
    int K;
    const SmallOp *ops;
    const SmallRot *rots;
    const SmallSeg *segs;
    int nsegs;
    const ExpGroup *groups;
    int ngroups;
    const ExpTerm *terms;
    double constant;
    uint64_t hf;
    void *workspace;    // gridDim.x slices of 2^n amplitudes (global-state variant)
    double *energies;   // B
    int cs_capacity;    // entries of the LDS cos/sin table
};

// ---- amplitude helpers, REAL = double, complex = double2 -----------------------------------------
template <bool REAL> struct Amp;
template <> struct Amp<true> {
    typedef double T;
    static __device__ __forceinline__ T basis(bool one) { return one ? 1.0 : 0.0; }
};
template <> struct Amp<false> {
    typedef double2 T;
    static __device__ __forceinline__ T basis(bool one) { return make_double2(one ? 1.0 : 0.0, 0.0); }
};

// real mixing (odd ny): u' = c u + si v ; v' = c v + sj u
__device__ __forceinline__ void mix_real(double &u, double &v, double c, double si, double sj) {
    const double nu = c * u + si * v;
    const double nv = c * v + sj * u;
    u = nu;
    v = nv;
}

template <bool REAL, int NT, typename A>
__device__ __forceinline__ void small_pass_pair(A *st, uint32_t npairs, const SmallOp &op,
                                                const SmallRot *__restrict__ rots, const double2 *cs, int rot_base) {
    constexpr int U = 4;
    const uint32_t x = (uint32_t)op.x;
    const int pivot = op.pivot;
    for (uint32_t k0 = threadIdx.x; k0 < npairs; k0 += NT * U) {
        A u[U], v[U];
        uint32_t ii[U];
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint32_t k = k0 + m * NT;
            ii[m] = (uint32_t)insert_zero(k, pivot);
            if (k < npairs) {
                u[m] = st[ii[m]];
                v[m] = st[ii[m] ^ x];
            }
        }
        for (int r = 0; r < op.count; ++r) {
            const uint32_t z = (uint32_t)rots[op.first + r].z;
            const int odd = rots[op.first + r].ny & 1;
            const double2 c = cs[op.first + r - rot_base];
#pragma unroll
            for (int m = 0; m < U; ++m) {
                const int pi = __popc(ii[m] & z) & 1;
                const int pj = pi ^ odd;
                const double si = pj ? -c.y : c.y;
                const double sj = pi ? -c.y : c.y;
                if constexpr (REAL) {
                    mix_real(u[m], v[m], c.x, si, sj);
                } else {
                    if (odd) {
                        mix_real(u[m].x, v[m].x, c.x, si, sj);
                        mix_real(u[m].y, v[m].y, c.x, si, sj);
                    } else {
                        const double ux = c.x * u[m].x + si * v[m].y, uy = c.x * u[m].y - si * v[m].x;
                        const double vx = c.x * v[m].x + sj * u[m].y, vy = c.x * v[m].y - sj * u[m].x;
                        u[m] = make_double2(ux, uy);
                        v[m] = make_double2(vx, vy);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint32_t k = k0 + m * NT;
            if (k < npairs) {
                st[ii[m]] = u[m];
                st[ii[m] ^ x] = v[m];
            }
        }
    }
}

// x == 0 runs are complex phases: only reachable in complex mode
template <int NT>
__device__ __forceinline__ void small_pass_diag(double2 *st, uint32_t namps, const SmallOp &op,
                                                const SmallRot *__restrict__ rots, const double2 *cs, int rot_base) {
    for (uint32_t i = threadIdx.x; i < namps; i += NT) {
        double2 a = st[i];
        for (int r = 0; r < op.count; ++r) {
            const uint32_t z = (uint32_t)rots[op.first + r].z;
            const double2 c = cs[op.first + r - rot_base];
            const double s = (__popc(i & z) & 1) ? -c.y : c.y;
            a = make_double2(c.x * a.x + s * a.y, c.x * a.y - s * a.x);
        }
        st[i] = a;
    }
}

template <bool REAL, int NT, typename A>
__device__ __forceinline__ void small_pass_gate(A *st, uint32_t namps, const SmallOp &op) {
    if (op.kind == OP_CNOT) {
        const int cb = op.first, tb = op.count;
        const int lo = cb < tb ? cb : tb, hi = cb < tb ? tb : cb;
        for (uint32_t k = threadIdx.x; k < (namps >> 2); k += NT) {
            const uint32_t i = (uint32_t)insert_zero(insert_zero(k, lo), hi) | (1u << cb), j = i | (1u << tb);
            const A a = st[i], b = st[j];
            st[i] = b;
            st[j] = a;
        }
        return;
    }
    const uint32_t bit = (uint32_t)op.x;
    for (uint32_t k = threadIdx.x; k < (namps >> 1); k += NT) {
        const uint32_t i = (uint32_t)insert_zero(k, op.pivot), j = i | bit;
        const A a = st[i], b = st[j];
        if (op.kind == OP_H) {
            const double r = 0.70710678118654752440;
            if constexpr (REAL) {
                st[i] = (a + b) * r;
                st[j] = (a - b) * r;
            } else {
                st[i] = make_double2((a.x + b.x) * r, (a.y + b.y) * r);
                st[j] = make_double2((a.x - b.x) * r, (a.y - b.y) * r);
            }
        } else {
            st[i] = b;
            st[j] = a;
        }
    }
}

// 8-point Walsh-Hadamard transform in registers (natural ordering: W[h] = sum_m (-1)^{parity(m&h)} w[m])
__device__ __forceinline__ void wht8(double *w) {
#pragma unroll
    for (int s = 1; s < 8; s <<= 1) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (!(m & s)) {
                const double a = w[m], b = w[m | s];
                w[m] = a + b;
                w[m | s] = a - b;
            }
        }
    }
}

// sum over the groups of 2 Re sum_k D(k) conj(a_i) a_j  (x = 0 group: sum_i D(i) |a_i|^2)
template <bool REAL, int NT, int LBITS, typename A>
__device__ __forceinline__ double small_expectation(const A *st, int n, const ExpGroup *__restrict__ groups, int ngroups,
                                                    const ExpTerm *__restrict__ terms) {
    double acc = 0.0;
    const uint32_t tid = threadIdx.x;
    for (int g = 0; g < ngroups; ++g) {
        const ExpGroup gr = groups[g];
        const bool diag = gr.x == 0;
        const uint32_t nk = diag ? (1u << n) : (1u << (n - 1));
        const double weight = diag ? 1.0 : 2.0;
        if (nk >= 8u * NT) {
            // chunks of 8*NT pair indices: k = cbase + m*NT + tid, m = 0..7
            for (uint32_t cbase = 0; cbase < nk; cbase += 8u * NT) {
                double wr[8], wi[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const uint32_t k = cbase + m * NT + tid;
                    if (diag) {
                        const A a = st[k];
                        if constexpr (REAL) wr[m] = a * a; else wr[m] = a.x * a.x + a.y * a.y;
                        wi[m] = 0.0;
                    } else {
                        const uint32_t i = (uint32_t)insert_zero(k, gr.pivot), j = i ^ gr.x;
                        const A a = st[i], c = st[j];
                        if constexpr (REAL) {
                            wr[m] = a * c;
                            wi[m] = 0.0;
                        } else {
                            wr[m] = a.x * c.x + a.y * c.y;  // conj(a_i) a_j
                            wi[m] = a.x * c.y - a.y * c.x;
                        }
                    }
                }
                wht8(wr);
                if constexpr (!REAL) wht8(wi);
                double part = 0.0;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    for (int t = gr.t0 + gr.off[h]; t < gr.t0 + gr.off[h + 1]; ++t) {
                        const ExpTerm et = terms[t];
                        // sign from the thread bits and the chunk bits; the 3 m-bits are in the bucket
                        const uint32_t kk = (cbase | tid) & et.zk;
                        const bool neg = __popc(kk) & 1;
                        double v;
                        if constexpr (REAL) v = et.cr * wr[h]; else v = et.cr * wr[h] - et.ci * wi[h];
                        part += neg ? -v : v;
                    }
                }
                acc += weight * part;
            }
        } else {
            // small registers: direct evaluation
            double part = 0.0;
            for (uint32_t k = tid; k < nk; k += NT) {
                double wr, wi = 0.0;
                if (diag) {
                    const A a = st[k];
                    if constexpr (REAL) wr = a * a; else wr = a.x * a.x + a.y * a.y;
                } else {
                    const uint32_t i = (uint32_t)insert_zero(k, gr.pivot), j = i ^ gr.x;
                    const A a = st[i], c = st[j];
                    if constexpr (REAL) {
                        wr = a * c;
                    } else {
                        wr = a.x * c.x + a.y * c.y;
                        wi = a.x * c.y - a.y * c.x;
                    }
                }
                double dr = 0.0, di = 0.0;
                for (int t = gr.t0; t < gr.t0 + gr.off[8]; ++t) {
                    const ExpTerm et = terms[t];
                    const bool neg = __popc(k & et.zk) & 1;
                    dr += neg ? -et.cr : et.cr;
                    di += neg ? -et.ci : et.ci;
                }
                part += dr * wr - di * wi;
            }
            acc += weight * part;
        }
    }
    return acc;
}

template <bool REAL, bool LDS_STATE, int NT, int LBITS>
__global__ __launch_bounds__(NT) void k_small_vqe(SmallArgs A) {
    typedef typename Amp<REAL>::T amp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t namps = 1u << A.n;
    // LDS layout: [state (LDS_STATE)] [cos/sin table] [reduction scratch]
    amp *st = LDS_STATE ? reinterpret_cast<amp *>(smem) : reinterpret_cast<amp *>(A.workspace) + (size_t)blockIdx.x * namps;
    double2 *cs = reinterpret_cast<double2 *>(smem + (LDS_STATE ? (size_t)namps * sizeof(amp) : 0));
    double2 *red = cs + A.cs_capacity;

    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        const double *th = A.theta + b * A.K;
        for (uint32_t i = threadIdx.x; i < namps; i += NT) st[i] = Amp<REAL>::basis(i == (uint32_t)A.hf);
        for (int sgi = 0; sgi < A.nsegs; ++sgi) {
            const SmallSeg sg = A.segs[sgi];
            __syncthreads();  // previous users of cs / state writes done
            for (int r = sg.rot0 + threadIdx.x; r < sg.rot1; r += NT) {
                const SmallRot sr = A.rots[r];
                const double phi = sr.phi0 + (sr.pidx >= 0 ? sr.coeff * th[sr.pidx] : 0.0);
                double s, c;
                sincos(phi, &s, &c);
                cs[r - sg.rot0] = make_double2(c, (sr.ny & 2) ? -s : s);
            }
            __syncthreads();
            for (int o = sg.op0; o < sg.op1; ++o) {
                const SmallOp op = A.ops[o];
                if (op.kind == OP_PAIR) {
                    small_pass_pair<REAL, NT>(st, namps >> 1, op, A.rots, cs, sg.rot0);
                } else if (op.kind == OP_DIAG) {
                    if constexpr (!REAL) small_pass_diag<NT>(st, namps, op, A.rots, cs, sg.rot0);
                } else {
                    small_pass_gate<REAL, NT>(st, namps, op);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        const double acc = small_expectation<REAL, NT, LBITS>(st, A.n, A.groups, A.ngroups, A.terms);
        const double2 tot = block_sum<NT>(make_double2(acc, 0.0), red);
        if (threadIdx.x == 0) A.energies[b] = tot.x + A.constant;
        __syncthreads();
    }
}

}  // namespace ovqe
