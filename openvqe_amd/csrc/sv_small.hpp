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

enum SmallOpKind : int32_t { OP_PAIR = 0, OP_DIAG = 1, OP_X = 2, OP_H = 3, OP_CNOT = 4, OP_TAB = 5 };

// OP_TAB — "commuting-run fusion": a run of rotations sharing the x mask, the parameter and the z part
// outside x (the JW strings of one fermionic excitation) commutes, so its product is ONE rotation of each
// pair (i, i^x) by chainsign(i) * theta * K_e, where e = the bits of i on the x positions and K_e a
// host-computed constant.  Patterns with K_e == 0 exactly (the fermionic selection rule: 7 of 8 for a
// double, 1 of 2 for a single excitation) are untouched pairs and are never visited.
struct SmallOp {
    uint64_t x;       // OP_PAIR/OP_TAB: x mask;  OP_X/OP_H: 1<<bit;  OP_CNOT: 1<<target
    int32_t kind;
    int32_t first;    // first table entry (rotations of the run / active patterns);  OP_CNOT: control bit
    int32_t count;    // entries;                                                    OP_CNOT: target bit
    int32_t pivot;    // highest x bit (OP_PAIR); bit (OP_X/OP_H)
    uint32_t zc;      // OP_TAB: z mask outside x (sign of the pair = parity(i & zc))
    uint32_t fixmask; // OP_TAB: index bits that are fixed by the pattern (= x)
    int32_t stream;   // OP_TAB: offset of the precomputed index stream (-1: compute indices in the kernel)
    uint32_t lognk;   // OP_TAB: log2 of the pairs per pattern
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
    uint32_t zk;   // z mask in free-index (k) space: the fixed positions squeezed out
    uint32_t pad;
    double cr, ci; // coefficient * i^ny * (-1)^{parity(x & z)}
};
// one (x-group, active pattern of the x-position bits) pair: the free index k runs over the bits NOT in
// fixmask; i = deposit(k) | ibits, j = i ^ x.  Fallback for wide x masks: fixmask = {pivot}, ibits = 0.
struct ExpGroup {
    uint32_t x;        // 0 for the diagonal group
    uint32_t ibits;    // values of the fixed bits
    int32_t t0;        // first term
    int32_t off[9];    // bucket b (bits [lbits, lbits+3) of zk == b) = terms [t0+off[b], t0+off[b+1])
    uint32_t fixmask;  // index bits fixed by the pattern
};

// a (x-group, pattern) entry with ONE real-coefficient term, sliced over lanes: this lane visits the free
// indices k = slice + nslices*m.  Entries are laid out entry-per-lane (adjacent lanes = adjacent slices of one
// entry, so their amplitudes differ in the low free bits -> different LDS banks).
struct FlatItem {
    uint16_t x;       // pair mask (also the fixed positions)
    uint16_t ibits;   // pattern bits on the non-pivot x positions
    uint16_t zc;      // sign mask OUTSIDE x, index space
    uint16_t slice;   // first free index
    uint32_t count;   // free indices visited by this lane
    uint32_t stride;  // nslices (power of two)
    double c;         // 2 * coefficient (pair trick), signs folded
};

// run of general groups whose terms fit the LDS staging area
struct ExpChunk {
    int32_t g0, g1, t0, t1;
};

constexpr int SMALL_OPS_CAP = 192;  // ops per segment staged in LDS

struct SmallArgs {
    int n;
    int K;
    int nsegs;
    int ngroups;
    int nchunks;
    int nflat;
    int cs_capacity;    // entries of the LDS rotation table
    int64_t B;
    double constant;
    uint64_t hf;
};

// per-rotation entry of the LDS table: angles resolved for this parameter vector + the z mask
struct RotLds {
    double c, s;     // cos(phi), sin(phi) * (ny&2 ? -1 : 1)
    uint32_t z;
    uint32_t odd;    // ny & 1
    uint64_t pad;
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

__device__ __forceinline__ uint32_t insert_zero32(uint32_t k, uint32_t lowmask) {
    return ((k & ~lowmask) << 1) | (k & lowmask);
}

// spread the free index k over the bit positions NOT in fixmask (zeros at the fixed positions)
__device__ __forceinline__ uint32_t deposit_index(uint32_t k, uint32_t fixmask) {
    while (fixmask) {  // wave-uniform: scalar loop, ascending positions
        const uint32_t lowbit = fixmask & (0u - fixmask);
        k = insert_zero32(k, lowbit - 1u);
        fixmask ^= lowbit;
    }
    return k;
}

// ``st`` is a pointer to the amplitudes or any view with operator[] (the tile kernels pass a bank-swizzled view)
template <bool REAL, int NT, int U, typename V>
__device__ __forceinline__ void small_pass_pair(V st, uint32_t npairs, const SmallOp &op, const RotLds *tab) {
    typedef typename Amp<REAL>::T A;
    const uint32_t x = (uint32_t)op.x;
    const int pivot = op.pivot;
    for (uint32_t k0 = threadIdx.x; k0 < npairs; k0 += NT * U) {
        A u[U], v[U];
        uint32_t ii[U];
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint32_t k = k0 + m * NT;
            ii[m] = insert_zero32(k, (1u << pivot) - 1u);
            if (k < npairs) {
                u[m] = st[ii[m]];
                v[m] = st[ii[m] ^ x];
            }
        }
        for (int r = 0; r < op.count; ++r) {
            const RotLds rl = tab[r];
#pragma unroll
            for (int m = 0; m < U; ++m) {
                const int pi = __popc(ii[m] & rl.z) & 1;
                const int pj = pi ^ (int)rl.odd;
                const double si = pj ? -rl.s : rl.s;
                const double sj = pi ? -rl.s : rl.s;
                if constexpr (REAL) {
                    mix_real(u[m], v[m], rl.c, si, sj);
                } else {
                    if (rl.odd) {
                        mix_real(u[m].x, v[m].x, rl.c, si, sj);
                        mix_real(u[m].y, v[m].y, rl.c, si, sj);
                    } else {
                        const double ux = rl.c * u[m].x + si * v[m].y, uy = rl.c * u[m].y - si * v[m].x;
                        const double vx = rl.c * v[m].x + sj * u[m].y, vy = rl.c * v[m].y - sj * u[m].x;
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
template <int NT, typename V>
__device__ __forceinline__ void small_pass_diag(V st, uint32_t namps, const SmallOp &op, const RotLds *tab) {
    for (uint32_t i = threadIdx.x; i < namps; i += NT) {
        double2 a = st[i];
        for (int r = 0; r < op.count; ++r) {
            const RotLds rl = tab[r];
            const double s = (__popc(i & rl.z) & 1) ? -rl.s : rl.s;
            a = make_double2(rl.c * a.x + s * a.y, rl.c * a.y - s * a.x);
        }
        st[i] = a;
    }
}

// OP_TAB with a host-precomputed index stream: entry e = pattern * 2^lognk + k holds (sign << 15) | i, so the pass
// is load-index, two LDS reads, one real rotation, two LDS writes — no per-op index arithmetic
template <bool REAL, int NT, typename A>
__device__ __forceinline__ void small_pass_tab_stream(A *st, const SmallOp &op, const RotLds *tab,
                                                      const uint16_t *__restrict__ stream) {
    const uint32_t x = (uint32_t)op.x;
    const uint32_t nent = (uint32_t)op.count << op.lognk;
    const uint16_t *sp = stream + op.stream;
    for (uint32_t e = threadIdx.x; e < nent; e += NT) {
        const uint32_t v = sp[e];
        const RotLds rl = tab[e >> op.lognk];
        const uint32_t i = v & 0x7fffu, j = i ^ x;
        const double s = (v & 0x8000u) ? -rl.s : rl.s;
        A u = st[i], w = st[j];
        if constexpr (REAL) {
            mix_real(u, w, rl.c, s, -s);
        } else {
            mix_real(u.x, w.x, rl.c, s, -s);
            mix_real(u.y, w.y, rl.c, s, -s);
        }
        st[i] = u;
        st[j] = w;
    }
}

template <bool REAL, int NT, typename V>
__device__ __forceinline__ void small_pass_tab(V st, int n, const SmallOp &op, const RotLds *tab) {
    typedef typename Amp<REAL>::T A;
    const uint32_t nk = 1u << (n - __popc(op.fixmask));
    const uint32_t x = (uint32_t)op.x;
    for (int p = 0; p < op.count; ++p) {
        const RotLds rl = tab[p];  // z = the pattern's fixed bits
        for (uint32_t k = threadIdx.x; k < nk; k += NT) {
            const uint32_t i = deposit_index(k, op.fixmask) | rl.z, j = i ^ x;
            const double s = (__popc(i & op.zc) & 1) ? -rl.s : rl.s;
            A u = st[i], v = st[j];
            // a pair of exact zeros stays a pair of exact zeros: nothing to rotate, and — what matters — nothing to
            // store (LDS stores run at a third of the read rate).  Sector-sparse states (UCC: a few percent of the
            // register) skip most of their pairs here; dense states pay one compare.
            if constexpr (REAL) {
                if (u == 0.0 && v == 0.0) continue;
                mix_real(u, v, rl.c, s, -s);
            } else {
                if (u.x == 0.0 && u.y == 0.0 && v.x == 0.0 && v.y == 0.0) continue;
                mix_real(u.x, v.x, rl.c, s, -s);
                mix_real(u.y, v.y, rl.c, s, -s);
            }
            st[i] = u;
            st[j] = v;
        }
    }
}

template <bool REAL, int NT, typename V>
__device__ __forceinline__ void small_pass_gate(V st, uint32_t namps, const SmallOp &op) {
    typedef typename Amp<REAL>::T A;
    if (op.kind == OP_CNOT) {
        const int cb = op.first, tb = op.count;
        const int lo = cb < tb ? cb : tb, hi = cb < tb ? tb : cb;
        for (uint32_t k = threadIdx.x; k < (namps >> 2); k += NT) {
            const uint32_t i = insert_zero32(insert_zero32(k, (1u << lo) - 1u), (1u << hi) - 1u) | (1u << cb), j = i | (1u << tb);
            const A a = st[i], b = st[j];
            st[i] = b;
            st[j] = a;
        }
        return;
    }
    const uint32_t bit = (uint32_t)op.x;
    for (uint32_t k = threadIdx.x; k < (namps >> 1); k += NT) {
        const uint32_t i = insert_zero32(k, (1u << op.pivot) - 1u), j = i | bit;
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

// M-point Walsh-Hadamard transform in registers (natural ordering: W[h] = sum_m (-1)^{parity(m&h)} w[m])
template <int M>
__device__ __forceinline__ void wht(double *w) {
#pragma unroll
    for (int s = 1; s < M; s <<= 1) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (!(m & s)) {
                const double a = w[m], b = w[m | s];
                w[m] = a + b;
                w[m | s] = a - b;
            }
        }
    }
}

// flat single-term entries, entry-per-lane: sum c * (-1)^{parity(i & zc)} Re(conj(a_i) a_j)
template <bool REAL, int NT, typename A>
__device__ __forceinline__ double small_expectation_flat(const A *st, const FlatItem *__restrict__ items, int nflat) {
    double acc = 0.0;
    for (int it = threadIdx.x; it < nflat; it += NT) {
        const FlatItem fi = items[it];
        const uint32_t x = fi.x, keep = ~x;
        // first index: deposit(slice) | ibits ; step: deposit(stride) (a single bit at a free position)
        uint32_t i = fi.slice, step = fi.stride, m = x;
        while (m) {
            const uint32_t lowbit = m & (0u - m);
            i = insert_zero32(i, lowbit - 1u);
            step = insert_zero32(step, lowbit - 1u);
            m ^= lowbit;
        }
        i |= fi.ibits;
        double part = 0.0;
        for (uint32_t c = 0; c < fi.count; ++c) {
            const A a = st[i], b = st[i ^ x];
            double w;
            if constexpr (REAL) w = a * b; else w = a.x * b.x + a.y * b.y;
            part += (__popc(i & fi.zc) & 1) ? -w : w;
            i = ((((i | x) + step) & keep) | fi.ibits);
        }
        acc += fi.c * part;
    }
    return acc;
}

// chunks of M*NT free indices: k = cbase + m*NT + tid, m = 0..M-1.  The M values a thread owns differ only in
// log2(M) index bits: after their M-point WHT one sign evaluation + FMA per term suffices (terms are bucketed
// on the host by exactly those bits of their mask).
template <bool REAL, int NT, int M, typename A>
__device__ __forceinline__ double exp_chunked(const A *st, const ExpGroup &gr, const ExpTerm *gt, uint32_t nk, bool diag) {
    const uint32_t tid = threadIdx.x;
    double part = 0.0;
    for (uint32_t cbase = 0; cbase < nk; cbase += (uint32_t)M * NT) {
        double wr[M], wi[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint32_t k = cbase + m * NT + tid;
            const uint32_t i = deposit_index(k, gr.fixmask) | gr.ibits;
            const A a = st[i];
            if (diag) {
                if constexpr (REAL) wr[m] = a * a; else wr[m] = a.x * a.x + a.y * a.y;
                wi[m] = 0.0;
            } else {
                const A c = st[i ^ gr.x];
                if constexpr (REAL) {
                    wr[m] = a * c;
                    wi[m] = 0.0;
                } else {
                    wr[m] = a.x * c.x + a.y * c.y;  // conj(a_i) a_j
                    wi[m] = a.x * c.y - a.y * c.x;
                }
            }
        }
        wht<M>(wr);
        if constexpr (!REAL) wht<M>(wi);
#pragma unroll
        for (int h = 0; h < M; ++h) {
            for (int t = gr.off[h]; t < gr.off[h + 1]; ++t) {
                const ExpTerm et = gt[t];
                // sign from the thread bits and the chunk bits; the m-bits are in the bucket
                const uint32_t kk = (cbase | tid) & et.zk;
                const bool neg = __popc(kk) & 1;
                double v;
                if constexpr (REAL) v = et.cr * wr[h]; else v = et.cr * wr[h] - et.ci * wi[h];
                part += neg ? -v : v;
            }
        }
    }
    return part;
}

// general (group, pattern) entries: 2 Re sum_k D(k) conj(a_i) a_j  (x = 0 entry: sum_i D(i) |a_i|^2);
// the term tables are staged chunk-wise into LDS (``lterms``) so the per-term reads are LDS broadcasts
template <bool REAL, int NT, int LBITS, typename A>
__device__ __forceinline__ double small_expectation(const A *st, int n, const ExpGroup *__restrict__ groups,
                                                    const ExpChunk *__restrict__ chunks, int nchunks,
                                                    const ExpTerm *__restrict__ terms, ExpTerm *lterms) {
    double acc = 0.0;
    const uint32_t tid = threadIdx.x;
    for (int ch = 0; ch < nchunks; ++ch) {
        const ExpChunk ck = chunks[ch];
        __syncthreads();
        for (int t = ck.t0 + (int)tid; t < ck.t1; t += NT) lterms[t - ck.t0] = terms[t];
        __syncthreads();
        for (int g = ck.g0; g < ck.g1; ++g) {
            const ExpGroup gr = groups[g];
            const ExpTerm *gt = lterms + (gr.t0 - ck.t0);
            const bool diag = gr.x == 0;
            const uint32_t nk = 1u << (n - __popc(gr.fixmask));
            const double weight = diag ? 1.0 : 2.0;
            if (nk >= 8u * NT) {
                acc += weight * exp_chunked<REAL, NT, 8>(st, gr, gt, nk, diag);
            } else if (nk >= 4u * NT) {
                acc += weight * exp_chunked<REAL, NT, 4>(st, gr, gt, nk, diag);
            } else if (nk >= 2u * NT) {
                acc += weight * exp_chunked<REAL, NT, 2>(st, gr, gt, nk, diag);
            } else {
                // few free indices per thread: direct evaluation
                double part = 0.0;
                for (uint32_t k = tid; k < nk; k += NT) {
                    const uint32_t i = deposit_index(k, gr.fixmask) | gr.ibits;
                    const A a = st[i];
                    double wr, wi = 0.0;
                    if (diag) {
                        if constexpr (REAL) wr = a * a; else wr = a.x * a.x + a.y * a.y;
                    } else {
                        const A c = st[i ^ gr.x];
                        if constexpr (REAL) {
                            wr = a * c;
                        } else {
                            wr = a.x * c.x + a.y * c.y;
                            wi = a.x * c.y - a.y * c.x;
                        }
                    }
                    double dr = 0.0, di = 0.0;
                    for (int t = 0; t < gr.off[8]; ++t) {
                        const ExpTerm et = gt[t];
                        const bool neg = __popc(k & et.zk) & 1;
                        dr += neg ? -et.cr : et.cr;
                        di += neg ? -et.ci : et.ci;
                    }
                    part += dr * wr - di * wi;
                }
                acc += weight * part;
            }
        }
    }
    return acc;
}

template <bool REAL, bool LDS_STATE, int NT, int LBITS>
__global__ __launch_bounds__(NT) void k_small_vqe(SmallArgs A, const double *__restrict__ theta,
                                                  const SmallOp *__restrict__ ops, const SmallRot *__restrict__ rots,
                                                  const SmallSeg *__restrict__ segs,
                                                  const ExpGroup *__restrict__ groups,
                                                  const ExpChunk *__restrict__ chunks,
                                                  const ExpTerm *__restrict__ terms,
                                                  const FlatItem *__restrict__ flat,
                                                  const uint16_t *__restrict__ stream, void *__restrict__ workspace,
                                                  double *__restrict__ energies) {
    typedef typename Amp<REAL>::T amp;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t namps = 1u << A.n;
    // LDS layout: [state (LDS_STATE)] [rotation table] [reduction scratch]
    amp *st = LDS_STATE ? reinterpret_cast<amp *>(smem) : reinterpret_cast<amp *>(workspace) + (size_t)blockIdx.x * namps;
    RotLds *tab = reinterpret_cast<RotLds *>(smem + (LDS_STATE ? (size_t)namps * sizeof(amp) : 0));
    SmallOp *lops = reinterpret_cast<SmallOp *>(tab + A.cs_capacity);  // ops of the current segment
    double2 *red = reinterpret_cast<double2 *>(lops + SMALL_OPS_CAP);

    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        const double *th = theta + b * A.K;
        for (uint32_t i = threadIdx.x; i < namps; i += NT) st[i] = Amp<REAL>::basis(i == (uint32_t)A.hf);
        for (int sgi = 0; sgi < A.nsegs; ++sgi) {
            const SmallSeg sg = segs[sgi];
            __syncthreads();  // previous users of the table / state writes done
            for (int r = sg.rot0 + threadIdx.x; r < sg.rot1; r += NT) {
                const SmallRot sr = rots[r];
                const double phi = sr.phi0 + (sr.pidx >= 0 ? sr.coeff * th[sr.pidx] : 0.0);
                double s, c;
                sincos(phi, &s, &c);
                RotLds rl;
                rl.c = c;
                rl.s = (sr.ny & 2) ? -s : s;
                rl.z = (uint32_t)sr.z;
                rl.odd = sr.ny & 1;
                rl.pad = 0;
                tab[r - sg.rot0] = rl;
            }
            for (int o = sg.op0 + threadIdx.x; o < sg.op1; o += NT) lops[o - sg.op0] = ops[o];
            __syncthreads();
            for (int o = sg.op0; o < sg.op1; ++o) {
                // LDS broadcast -> make the fields wave-uniform scalars again (scalar loops / branches below)
                SmallOp op = lops[o - sg.op0];
                op.x = (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)op.x);
                op.kind = __builtin_amdgcn_readfirstlane(op.kind);
                op.first = __builtin_amdgcn_readfirstlane(op.first);
                op.count = __builtin_amdgcn_readfirstlane(op.count);
                op.pivot = __builtin_amdgcn_readfirstlane(op.pivot);
                op.zc = __builtin_amdgcn_readfirstlane(op.zc);
                op.fixmask = __builtin_amdgcn_readfirstlane(op.fixmask);
                op.stream = __builtin_amdgcn_readfirstlane(op.stream);
                op.lognk = __builtin_amdgcn_readfirstlane(op.lognk);
                if (op.kind == OP_PAIR) {
                    small_pass_pair<REAL, NT, REAL ? 8 : 4>(st, namps >> 1, op, tab + (op.first - sg.rot0));
                } else if (op.kind == OP_TAB) {
                    if (op.stream >= 0) small_pass_tab_stream<REAL, NT>(st, op, tab + (op.first - sg.rot0), stream);
                    else small_pass_tab<REAL, NT>(st, A.n, op, tab + (op.first - sg.rot0));
                } else if (op.kind == OP_DIAG) {
                    if constexpr (!REAL) small_pass_diag<NT>(st, namps, op, tab + (op.first - sg.rot0));
                } else {
                    small_pass_gate<REAL, NT>(st, namps, op);
                }
                if constexpr (LDS_STATE) {
                    // the state is in LDS: the barrier only has to order LDS traffic
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        double acc = small_expectation_flat<REAL, NT>(st, flat, A.nflat);
        // the rotation table is idle now: its LDS space stages the term tables of the general entries
        acc += small_expectation<REAL, NT, LBITS>(st, A.n, groups, chunks, A.nchunks, terms,
                                                  reinterpret_cast<ExpTerm *>(tab));
        const double2 tot = block_sum<NT>(make_double2(acc, 0.0), red);
        if (threadIdx.x == 0) energies[b] = tot.x + A.constant;
        __syncthreads();
    }
}

}  // namespace ovqe
