// sv_sparse.hpp — support-compacted evaluation: E(theta) on the REACHABLE SUPPORT of the circuit.
//
// A fused OP_TAB pass (sv_small.hpp) never touches a pair whose rotation angle is structurally zero, so an
// amplitude that is not reachable from |HF> through the active pairs of the program is exactly 0.0 for every
// parameter vector.  The host propagates that support through the program once (ovqe_sv.hip,
// build_sparse_program): S_0 = {hf}, S_k = S_{k-1} + partners of S_{k-1} under the active pairs of op k.  For a
// particle-number / spin conserving ansatz on a Hartree-Fock determinant this is the CI space of the sector
// (H2O/STO-3G: 441 of 16384 amplitudes); nothing about fermions is assumed — the analysis is on Pauli masks.
//
// When the support is small the whole evaluation is re-expressed on compact indices 0..m-1:
//   * per op: the list of active pairs inside the support, (ci, cj, chain sign, pattern id);
//   * <H>: the list of (ci, cj, coefficient) with H_{ij} != 0 inside the support — the quadratic form of the
//     Hamiltonian restricted to S (coefficients D_g(i) summed on the host, real mode).
// Results are identical to the dense kernels up to the summation order of <H> (skipped work is exact zeros).
// One WAVE owns SPW evaluations: no workgroup barrier exists anywhere; LDS traffic of a wave is in order.
#pragma once
#include "sv_small.hpp"

namespace ovqe {

struct SpOp {
    int32_t first;   // first pair
    int32_t npairs;
    int32_t tab0;    // first table entry of the op (pattern p -> tab0 + p)
    int32_t pad;
};

// pair word: ci (12 bits) | cj (12 bits) << 12 | sign << 24 | pattern << 25
struct SpEntry {
    uint32_t ij;  // ci | cj << 12   (ci == cj: diagonal entry)
    uint32_t pad;
    double c;     // 2 * H_ij (off-diagonal, pair counted once) or H_ii
};

struct SparseArgs {
    int m;        // support size
    int mpad;     // per-evaluation stride of the LDS state in doubles: m rounded up to even, so that the double2
                  // cos/sin table behind SPW states starts on a 16-byte boundary (ds_read/write_b128)
    int K;
    int nops;
    int ntab;
    int nent;
    int npairs;   // all active pairs of the program (STAGE: copied to LDS once per launch)
    int hf;       // compact slot of |hf>
    int64_t B;
    double constant;
    int dbg = 0;  // measurements ("sparse_dbg"; k_sparse_vqe_rows only): 1 no sincos, 2 no circuit rows, 3 no Hamiltonian entries
};

// STAGE = true (small batches, the latency path of one-evaluation-per-call optimisers): the op table and the pair
// words are copied to LDS at kernel start, so the chain op -> pair word -> amplitudes never waits for global memory.
template <int SPW, bool STAGE>
__global__ __launch_bounds__(64) void k_sparse_vqe(SparseArgs A, const double *__restrict__ theta,
                                                   const SmallRot *__restrict__ tabrots,
                                                   const SpOp *__restrict__ ops, const uint32_t *__restrict__ pairs,
                                                   const SpEntry *__restrict__ entries,
                                                   double *__restrict__ energies) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *st = reinterpret_cast<double *>(smem);                          // [SPW][mpad]
    double2 *cs = reinterpret_cast<double2 *>(st + (size_t)SPW * A.mpad);   // [SPW][ntab], 16-byte aligned
    const int lane = threadIdx.x;
    SpOp *lops = reinterpret_cast<SpOp *>(cs + (size_t)SPW * A.ntab);     // [nops]   (STAGE)
    uint32_t *lpairs = reinterpret_cast<uint32_t *>(lops + A.nops);       // [npairs] (STAGE)
    if constexpr (STAGE) {
        for (int i = lane; i < A.nops; i += 64) lops[i] = ops[i];
        for (int i = lane; i < A.npairs; i += 64) lpairs[i] = pairs[i];
    }
    // the wave's lanes are split evenly over its SPW evaluations (no index division in the inner loops)
    constexpr int LPS = 64 / SPW;
    const int s = lane / LPS, l = lane % LPS;
    double *base = st + (size_t)s * A.mpad;
    const double2 *csb = cs + (size_t)s * A.ntab;
    const int64_t nwork = (A.B + SPW - 1) / SPW;
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const int64_t b0 = w * SPW;
        // |HF> (compact index 0) and the cos/sin table of every active pattern, per evaluation
        for (int i = lane; i < SPW * A.mpad; i += 64) st[i] = (i % A.mpad == A.hf) ? 1.0 : 0.0;
        {
            const int64_t b = b0 + s < A.B ? b0 + s : A.B - 1;
            const double *th = theta + b * A.K;
            for (int e = l; e < A.ntab; e += LPS) {
                const SmallRot sr = tabrots[e];
                double sn, c;
                sincos(sr.coeff * th[sr.pidx], &sn, &c);
                cs[(size_t)s * A.ntab + e] = make_double2(c, sn);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (STAGE) {
            for (int o = 0; o < A.nops; ++o) {
                SpOp op = lops[o];
                op.first = __builtin_amdgcn_readfirstlane(op.first);
                op.npairs = __builtin_amdgcn_readfirstlane(op.npairs);
                op.tab0 = __builtin_amdgcn_readfirstlane(op.tab0);
                for (int pe = l; pe < op.npairs; pe += LPS) {
                    const uint32_t pw = lpairs[op.first + pe];
                    const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu;
                    const double2 t = csb[op.tab0 + (int)(pw >> 25)];
                    const double sn = (pw & (1u << 24)) ? -t.y : t.y;
                    const double u = base[ci], v = base[cj];
                    base[ci] = t.x * u + sn * v;
                    base[cj] = t.x * v - sn * u;
                }
                // pairs of one op are disjoint; the next op may touch them from other lanes of this wave: the LDS unit
                // executes a wave's DS instructions in issue order, so only the COMPILER must not reorder across ops
                asm volatile("" ::: "memory");
            }
        } else {
            // op records and pair words stream from L2: the chain record -> pair word -> amplitudes of consecutive ops is
            // what bounds this kernel, so both are fetched AHEAD — the record of op o + 3 and the first pair words of ops o + 1,
            // o + 2 are in flight while op o rotates its pairs (most ops have fewer pairs than lanes: one word per lane)
            const int last = A.nops - 1;
            SpOp op1 = ops[0], op2 = ops[min(1, last)], op3 = ops[min(2, last)];
            uint32_t pw1 = l < op1.npairs ? pairs[op1.first + l] : 0u;
            uint32_t pw2 = (last >= 1 && l < op2.npairs) ? pairs[op2.first + l] : 0u;
            for (int o = 0; o < A.nops; ++o) {
                const SpOp op = op1;
                uint32_t pw = pw1;
                op1 = op2;
                pw1 = pw2;
                op2 = op3;
                op3 = ops[min(o + 3, last)];
                pw2 = (o + 2 <= last && l < op2.npairs) ? pairs[op2.first + l] : 0u;
                for (int pe = l; pe < op.npairs; pe += LPS) {
                    if (pe != l) pw = pairs[op.first + pe];
                    const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu;
                    const double2 t = csb[op.tab0 + (int)(pw >> 25)];
                    const double sn = (pw & (1u << 24)) ? -t.y : t.y;
                    const double u = base[ci], v = base[cj];
                    base[ci] = t.x * u + sn * v;
                    base[cj] = t.x * v - sn * u;
                }
                asm volatile("" ::: "memory");
            }
        }
        double acc[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) acc[s] = 0.0;
#pragma unroll 4  // four entries in flight: the loop is bound by the latency of its loads, not by issue
        for (int e = lane; e < A.nent; e += 64) {
            const SpEntry en = entries[e];
            const uint32_t ci = en.ij & 0xfffu, cj = (en.ij >> 12) & 0xfffu;
#pragma unroll
            for (int s = 0; s < SPW; ++s) acc[s] += en.c * st[(size_t)s * A.mpad + ci] * st[(size_t)s * A.mpad + cj];
        }
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const double tot = wave_sum(acc[s]);
            if (lane == 0 && b0 + s < A.B) energies[b0 + s] = tot + A.constant;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---- throughput form of the circuit (round 3): rows of 32 padded 64-bit words ------------------------------------------------
// rocprofv3 counters of k_sparse_vqe<2> on the H2O workload: the LDS pipe is active 6 % of the wave cycles and bank-conflict
// cycles can be cut by a third without any change in run time; 39 % of the wave cycles are instruction issue — about 60
// instructions per op and wave (25 scalar: op records three ops ahead, loop bounds; 20 vector: unpacking the pair word,
// addresses, sign) around 4 f64 operations and 5 LDS accesses.  This form removes the bookkeeping: the program is a flat
// list of ROWS of 32 words (an op's pairs in chunks of 32, padded), one word per lane and row, everything pre-computed in it:
//   bits 0-15 byte offset of amplitude i | 16-31 byte offset of amplitude j | 32-47 byte offset of the cos/sin entry | 63 sign
// Padded lanes rotate two private spare slots behind the support by the identity entry behind the table (no branch).  Rows are
// fetched four ahead in registers; per row: one 8-byte load, three LDS reads, four f64 operations, two LDS writes.
template <int SPW, int DBG = 0>   // DBG: measurements ("sparse_dbg": 1 no sincos, 2 no circuit rows, 3 no Hamiltonian entries) — a
// run-time test of A.dbg in the loop bounds cost the product kernel 12 % (0.714 -> 0.80 ms per 65 536 evaluations), hence compile time
__global__ __launch_bounds__(64) void k_sparse_vqe_rows(SparseArgs A, const double *__restrict__ theta,
                                                        const SmallRot *__restrict__ tabrots, const uint64_t *__restrict__ rows,
                                                        int nrows4, const SpEntry *__restrict__ entries,
                                                        double *__restrict__ energies) {
    static_assert(SPW == 2, "rows are built for two evaluations per wave (32 lanes each)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *st = reinterpret_cast<double *>(smem);                          // [SPW][mpad], mpad = support + 64 spare slots
    double2 *cs = reinterpret_cast<double2 *>(st + (size_t)SPW * A.mpad);   // [SPW][ntab + 1]: the last entry is the identity
    const int lane = threadIdx.x;
    constexpr int LPS = 64 / SPW;
    const int s = lane / LPS, l = lane % LPS;
    const int ntab1 = A.ntab + 1;
    unsigned char *sbase = smem + (size_t)s * A.mpad * sizeof(double);
    unsigned char *cbase = smem + (size_t)SPW * A.mpad * sizeof(double) + (size_t)s * ntab1 * sizeof(double2);
    const int64_t nwork = (A.B + SPW - 1) / SPW;
    const uint64_t *rp = rows + l;
    for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
        const int64_t b0 = w * SPW;
        uint64_t w0 = rp[0], w1 = rp[32], w2 = rp[64], w3 = rp[96];   // (the table ends with four spare rows)
        for (int i = lane; i < SPW * A.mpad; i += 64) st[i] = (i % A.mpad == A.hf) ? 1.0 : 0.0;
        {
            const int64_t b = b0 + s < A.B ? b0 + s : A.B - 1;
            const double *th = theta + b * A.K;
            for (int e = l; e < A.ntab; e += LPS) {
                const SmallRot sr = tabrots[e];
                double sn, c;
                if (DBG == 1) {
                    sn = sr.coeff * th[sr.pidx];
                    c = 1.0;
                } else {
                    sincos(sr.coeff * th[sr.pidx], &sn, &c);
                }
                cs[(size_t)s * ntab1 + e] = make_double2(c, sn);
            }
            if (l == 0) cs[(size_t)s * ntab1 + A.ntab] = make_double2(1.0, 0.0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto apply = [&](uint64_t word) {
            const uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
            double *pi = reinterpret_cast<double *>(sbase + (lo & 0xffffu));
            double *pj = reinterpret_cast<double *>(sbase + (lo >> 16));
            const double2 t = *reinterpret_cast<const double2 *>(cbase + (hi & 0xffffu));
            const double sn = __hiloint2double(__double2hiint(t.y) ^ (int)(hi & 0x80000000u), __double2loint(t.y));
            const double u = *pi, v = *pj;
            *pi = t.x * u + sn * v;
            *pj = t.x * v - sn * u;
            // pairs of one row are disjoint; the next row may touch them from other lanes of this wave: the LDS unit executes a
            // wave's DS instructions in issue order, so only the COMPILER must not reorder across rows
            asm volatile("" ::: "memory");
        };
        for (int r = 0; r < (DBG == 2 ? 0 : nrows4); r += 4) {
            const uint64_t *nx = rp + (size_t)(r + 4) * 32;
            apply(w0);
            w0 = nx[0];
            apply(w1);
            w1 = nx[32];
            apply(w2);
            w2 = nx[64];
            apply(w3);
            w3 = nx[96];
        }
        double acc[SPW];
#pragma unroll
        for (int q = 0; q < SPW; ++q) acc[q] = 0.0;
        // (Measured and dropped: the next trip's entries in flight while this trip's are contracted — 0.69 ms per 65 536 against 0.67:
        // with nine waves per CU the kernel is bound by LDS instructions, ~5 per pair and 2 per entry and state at ~3 ns each per CU,
        // tools/micro/lds_atomic.hip, not by this loop's latency.  Also measured and dropped: the restricted Hamiltonian in row format
        // (lane = row, a_i once per slice of 64 rows, ONE amplitude read per entry and state instead of two, entries ordered
        // against bank conflicts): 0.673 ms against 0.675 — the entry loop's LDS reads are not what bounds the kernel either;
        // per wave and pair of evaluations ~6600 VALU, 3700 SALU, 1400 LDS instructions at 2.25 waves per SIMD.)
        // (Round 4, measured and dropped: the two states copied side by side behind the circuit (slot -> double2) so that ONE 16-byte
        // LDS read per amplitude serves both states and one address is computed instead of two — bit-identical energies, 0.885 ms
        // against 0.755: random 16-byte reads conflict more than twice as often as random 8-byte reads.  tools/exp_value_phases.py:
        // of the 0.79 ms per 65 536 evaluations through host buffers this loop is 0.40, the circuit rows 0.23, the sincos 0.03.)
#pragma unroll 4
        for (int e = lane; e < (DBG == 3 ? 0 : A.nent); e += 64) {
            const SpEntry en = entries[e];
            const uint32_t ci = en.ij & 0xfffu, cj = (en.ij >> 12) & 0xfffu;
#pragma unroll
            for (int q = 0; q < SPW; ++q) acc[q] += en.c * st[(size_t)q * A.mpad + ci] * st[(size_t)q * A.mpad + cj];
        }
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const double tot = wave_sum(acc[q]);
            if (lane == 0 && b0 + q < A.B) energies[b0 + q] = tot + A.constant;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---- latency form (round 3): ONE evaluation per workgroup of NT threads ---------------------------------------------------------
// What scipy's one-evaluation-per-call optimisers see.  rocprofv3: the single-wave kernel above takes 45 us per H2O evaluation
// — 26 us of it its wave walking the 9443 entries of the restricted Hamiltonian four loads at a time, the rest 140 ops of ~60
// instructions plus staging.  Here the workgroup's NT threads prepare everything in parallel (state, ONE sincos per thread) and
// pull the Hamiltonian's entries into REGISTERS while wave 0 runs the circuit — rows of 64 padded words straight from memory,
// eight rows ahead — then all waves contract their entries against the state in LDS.
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_sparse_vqe_wg(SparseArgs A, const double *__restrict__ theta,
                                                      const SmallRot *__restrict__ tabrots, const uint64_t *__restrict__ rows,
                                                      int nrows8, const SpEntry *__restrict__ entries,
                                                      double *__restrict__ energies) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *st = reinterpret_cast<double *>(smem);                 // [mpad] = support + 128 spare slots
    double2 *cs = reinterpret_cast<double2 *>(st + A.mpad);        // [ntab + 1]: the last entry is the identity
    __shared__ double2 red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef OVQE_TESTING
    uint64_t stamp[6] = {};   // "sparse_dbg" = 9: phase times of the first evaluation of workgroup 0 (s_memrealtime ticks of 10 ns)
#define OVQE_STAMP(k) do { if (A.dbg == 9) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define OVQE_STAMP(k) do { } while (0)
#endif
    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        OVQE_STAMP(0);
        const double *th = theta + b * A.K;
        // Hamiltonian entries of this thread: in flight from here on
        SpEntry ent[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) ent[k] = entries[min(tid + k * NT, A.nent - 1)];
        uint64_t w[8];
        const uint64_t *rp = rows + lane;
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = rp[k * 64];          // (the table ends with eight spare rows)
        }
        for (int i = tid; i < A.mpad; i += NT) st[i] = i == A.hf ? 1.0 : 0.0;
        for (int e = tid; e < A.ntab; e += NT) {
            const SmallRot sr = tabrots[e];
            double sn, c;
            sincos(sr.coeff * th[sr.pidx], &sn, &c);
            cs[e] = make_double2(c, sn);
        }
        if (tid == 0) cs[A.ntab] = make_double2(1.0, 0.0);
        OVQE_STAMP(1);
        __syncthreads();
        OVQE_STAMP(2);
        if (wave == 0) {
            auto apply = [&](uint64_t word) {
                const uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
                double *pi = reinterpret_cast<double *>(smem + (lo & 0xffffu));
                double *pj = reinterpret_cast<double *>(smem + (lo >> 16));
                const double2 t = *reinterpret_cast<const double2 *>(reinterpret_cast<const unsigned char *>(cs) + (hi & 0xffffu));
                const double sn = __hiloint2double(__double2hiint(t.y) ^ (int)(hi & 0x80000000u), __double2loint(t.y));
                const double u = *pi, v = *pj;
                *pi = t.x * u + sn * v;
                *pj = t.x * v - sn * u;
                asm volatile("" ::: "memory");   // a wave's DS instructions execute in issue order (see k_sparse_vqe)
            };
            for (int r = 0; r < nrows8; r += 8) {
                const uint64_t *nx = rp + (size_t)(r + 8) * 64;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    apply(w[k]);
                    w[k] = nx[k * 64];
                }
            }
        }
        OVQE_STAMP(3);
        __syncthreads();
        OVQE_STAMP(4);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (tid + k * NT < A.nent) acc += ent[k].c * st[ent[k].ij & 0xfffu] * st[(ent[k].ij >> 12) & 0xfffu];
        for (int e = tid + EPT * NT; e < A.nent; e += NT) {       // (restricted Hamiltonians beyond EPT x NT entries)
            const SpEntry en = entries[e];
            acc += en.c * st[en.ij & 0xfffu] * st[(en.ij >> 12) & 0xfffu];
        }
        const double2 tot = block_sum<NT>(make_double2(acc, 0.0), red);
        if (tid == 0) energies[b] = tot.x + A.constant;
#ifdef OVQE_TESTING
        if (A.dbg == 9 && tid == 0 && blockIdx.x == 0 && b == 0) {
            stamp[5] = __builtin_amdgcn_s_memrealtime();
            printf("k_sparse_vqe_wg phases (10-ns ticks): prologue %llu, barrier %llu, circuit (wave 0) %llu, barrier %llu, <H> + reduce %llu; rows %d entries %d\n",
                   (unsigned long long)(stamp[1] - stamp[0]), (unsigned long long)(stamp[2] - stamp[1]), (unsigned long long)(stamp[3] - stamp[2]),
                   (unsigned long long)(stamp[4] - stamp[3]), (unsigned long long)(stamp[5] - stamp[4]), nrows8, A.nent);
        }
#endif
    }
#undef OVQE_STAMP
}

// ---- exact gradient on the compact support (round 3) ---------------------------------------------------------------------
// E(theta) and dE/dtheta_k for ALL K parameters of one parameter vector per wave, in ONE launch: forward circuit as above,
// lambda = H psi from the restricted Hamiltonian's entries (f64 LDS atomics: lambda_i += H_ij a_j, lambda_j += H_ij a_i),
// then the ops BACKWARDS on psi and lambda together — per pair g = lambda_i psi_j - lambda_j psi_i on the states after the op,
// w[table entry] += +-g, both states rotated back — and dE/dtheta_k = sum over the table entries of parameter k of
// 2 coeff w.  (The adjoint method of ovqe_energy_gradient's streaming and sector paths on the support: the streaming form
// needs ~6 launches per generator — H2O: 1.6 ms for 140 derivatives.)  The order of the atomic additions is not fixed: the
// gradient reproduces to rounding, the energy bit for bit.
template <bool STAGE>
__global__ __launch_bounds__(64) void k_sparse_grad(SparseArgs A, const double *__restrict__ theta, const SmallRot *__restrict__ tabrots,
                                                    const SpOp *__restrict__ ops, const uint32_t *__restrict__ pairs,
                                                    const SpEntry *__restrict__ entries, double *__restrict__ energies,
                                                    double *__restrict__ grads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *psi = reinterpret_cast<double *>(smem);                 // [mpad]
    double *lam = psi + A.mpad;                                     // [mpad]
    double2 *cs = reinterpret_cast<double2 *>(lam + A.mpad);        // [ntab]
    double *w = reinterpret_cast<double *>(cs + A.ntab);            // [ntab]
    double *gk = w + A.ntab;                                        // [K]
    SpOp *lops = reinterpret_cast<SpOp *>(gk + ((A.K + 1) & ~1));   // [nops]   (STAGE)
    uint32_t *lpairs = reinterpret_cast<uint32_t *>(lops + A.nops); // [npairs] (STAGE)
    const int lane = threadIdx.x;
    if constexpr (STAGE) {
        for (int i = lane; i < A.nops; i += 64) lops[i] = ops[i];
        for (int i = lane; i < A.npairs; i += 64) lpairs[i] = pairs[i];
    }
    const SpOp *opv = STAGE ? lops : ops;
    const uint32_t *pv = STAGE ? lpairs : pairs;
    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        const double *th = theta + b * A.K;
        for (int i = lane; i < A.mpad; i += 64) {
            psi[i] = i == A.hf ? 1.0 : 0.0;
            lam[i] = 0.0;
        }
        for (int e = lane; e < A.ntab; e += 64) {
            const SmallRot sr = tabrots[e];
            double sn, c;
            sincos(sr.coeff * th[sr.pidx], &sn, &c);
            cs[e] = make_double2(c, sn);
            w[e] = 0.0;
        }
        for (int k = lane; k < A.K; k += 64) gk[k] = 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // forward
        for (int o = 0; o < A.nops; ++o) {
            SpOp op = opv[o];
            op.first = __builtin_amdgcn_readfirstlane(op.first);
            op.npairs = __builtin_amdgcn_readfirstlane(op.npairs);
            op.tab0 = __builtin_amdgcn_readfirstlane(op.tab0);
            for (int pe = lane; pe < op.npairs; pe += 64) {
                const uint32_t pw = pv[op.first + pe];
                const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu;
                const double2 t = cs[op.tab0 + (int)(pw >> 25)];
                const double sn = (pw & (1u << 24)) ? -t.y : t.y;
                const double u = psi[ci], v = psi[cj];
                psi[ci] = t.x * u + sn * v;
                psi[cj] = t.x * v - sn * u;
            }
            asm volatile("" ::: "memory");   // a wave's DS instructions execute in issue order (see k_sparse_vqe)
        }
        // lambda = H psi, E = <psi|lambda>
        double acc = 0.0;
#pragma unroll 4
        for (int e = lane; e < A.nent; e += 64) {
            const SpEntry en = entries[e];
            const uint32_t ci = en.ij & 0xfffu, cj = (en.ij >> 12) & 0xfffu;
            const double ai = psi[ci], aj = psi[cj];
            acc += en.c * ai * aj;
            if (ci == cj) {
                __hip_atomic_fetch_add(&lam[ci], en.c * ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {   // c = 2 H_ij, the pair counted once
                __hip_atomic_fetch_add(&lam[ci], 0.5 * en.c * aj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&lam[cj], 0.5 * en.c * ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        const double etot = wave_sum(acc);
        if (lane == 0) energies[b] = etot + A.constant;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // backward
        for (int o = A.nops - 1; o >= 0; --o) {
            SpOp op = opv[o];
            op.first = __builtin_amdgcn_readfirstlane(op.first);
            op.npairs = __builtin_amdgcn_readfirstlane(op.npairs);
            op.tab0 = __builtin_amdgcn_readfirstlane(op.tab0);
            for (int pe = lane; pe < op.npairs; pe += 64) {
                const uint32_t pw = pv[op.first + pe];
                const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu;
                const int e = op.tab0 + (int)(pw >> 25);
                const double2 t = cs[e];
                const bool neg = (pw & (1u << 24)) != 0u;
                const double sn = neg ? -t.y : t.y;
                const double u1 = psi[ci], v1 = psi[cj], lu = lam[ci], lv = lam[cj];
                const double g = lu * v1 - lv * u1;
                __hip_atomic_fetch_add(&w[e], neg ? -g : g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                psi[ci] = t.x * u1 - sn * v1;
                psi[cj] = t.x * v1 + sn * u1;
                lam[ci] = t.x * lu - sn * lv;
                lam[cj] = t.x * lv + sn * lu;
            }
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int e = lane; e < A.ntab; e += 64) {
            const SmallRot sr = tabrots[e];
            if (sr.pidx >= 0) __hip_atomic_fetch_add(&gk[sr.pidx], 2.0 * sr.coeff * w[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = lane; k < A.K; k += 64) grads[b * A.K + k] = gk[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ... and with ONE parameter vector per workgroup (latency form, as k_sparse_vqe_wg): all threads prepare and hold the restricted
// Hamiltonian's entries in registers, wave 0 runs the circuit forwards over rows of 64 padded words, all waves form
// E = <psi|H|psi> and lambda = H psi (f64 LDS atomics), wave 0 walks the rows BACKWARDS on psi and lambda — per row
// g = lambda_i psi_j - lambda_j psi_i summed over the wave by DPP row operations when the row has one cos/sin entry (the rows of a
// JW excitation do, their padded lanes carry it too), by LDS atomics otherwise — and all threads fold w into dE/dtheta.
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_sparse_grad_wg(SparseArgs A, const double *__restrict__ theta, const SmallRot *__restrict__ tabrots,
                                                       const uint64_t *__restrict__ rows, int nrows8, const SpEntry *__restrict__ entries,
                                                       double *__restrict__ energies, double *__restrict__ grads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *psi = reinterpret_cast<double *>(smem);                 // [mpad]
    double *lam = psi + A.mpad;                                     // [mpad]
    double2 *cs = reinterpret_cast<double2 *>(lam + A.mpad);        // [ntab + 1]
    double *w = reinterpret_cast<double *>(cs + A.ntab + 1);        // [ntab + 1]
    double *gk = w + ((A.ntab + 2) & ~1);                           // [K]
    __shared__ double2 red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lam_off = (uint32_t)A.mpad * (uint32_t)sizeof(double);
    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        const double *th = theta + b * A.K;
        SpEntry ent[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) ent[k] = entries[min(tid + k * NT, A.nent - 1)];
        uint64_t wd[8];
        const uint64_t *rp = rows + lane;
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) wd[k] = rp[k * 64];
        }
        for (int i = tid; i < A.mpad; i += NT) {
            psi[i] = i == A.hf ? 1.0 : 0.0;
            lam[i] = 0.0;
        }
        for (int e = tid; e < A.ntab; e += NT) {
            const SmallRot sr = tabrots[e];
            double sn, c;
            sincos(sr.coeff * th[sr.pidx], &sn, &c);
            cs[e] = make_double2(c, sn);
            w[e] = 0.0;
        }
        if (tid == 0) {
            cs[A.ntab] = make_double2(1.0, 0.0);
            w[A.ntab] = 0.0;
        }
        for (int k = tid; k < A.K; k += NT) gk[k] = 0.0;
        __syncthreads();
        if (wave == 0) {   // forward
            for (int r = 0; r < nrows8; r += 8) {
                const uint64_t *nx = rp + (size_t)(r + 8) * 64;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t lo = (uint32_t)wd[k], hi = (uint32_t)(wd[k] >> 32);
                    double *pi = reinterpret_cast<double *>(smem + (lo & 0xffffu));
                    double *pj = reinterpret_cast<double *>(smem + (lo >> 16));
                    const double2 t = *reinterpret_cast<const double2 *>(reinterpret_cast<const unsigned char *>(cs) + (hi & 0xffffu));
                    const double sn = __hiloint2double(__double2hiint(t.y) ^ (int)(hi & 0x80000000u), __double2loint(t.y));
                    const double u = *pi, v = *pj;
                    *pi = t.x * u + sn * v;
                    *pj = t.x * v - sn * u;
                    asm volatile("" ::: "memory");
                    wd[k] = nx[k * 64];
                }
            }
            // the last eight rows, for the way back (the spare rows behind the table were fetched last)
#pragma unroll
            for (int k = 0; k < 8; ++k) wd[k] = rp[(size_t)(nrows8 - 1 - k) * 64];
        }
        __syncthreads();
        double acc = 0.0;
        auto entry = [&](const SpEntry &en) {
            const uint32_t ci = en.ij & 0xfffu, cj = (en.ij >> 12) & 0xfffu;
            const double ai = psi[ci], aj = psi[cj];
            acc += en.c * ai * aj;
            if (ci == cj) {
                __hip_atomic_fetch_add(&lam[ci], en.c * ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {   // c = 2 H_ij, the pair counted once
                __hip_atomic_fetch_add(&lam[ci], 0.5 * en.c * aj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&lam[cj], 0.5 * en.c * ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
#pragma unroll
        for (int k = 0; k < EPT; ++k)
            if (tid + k * NT < A.nent) entry(ent[k]);
        for (int e = tid + EPT * NT; e < A.nent; e += NT) entry(entries[e]);
        const double2 tot = block_sum<NT>(make_double2(acc, 0.0), red);   // (its barriers also complete lambda)
        if (tid == 0) energies[b] = tot.x + A.constant;
        if (wave == 0) {   // backward: rows nrows8 - 1 ... 0 (the rows of one op are independent of each other)
            for (int r = nrows8 - 1; r >= 0; r -= 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t lo = (uint32_t)wd[k], hi = (uint32_t)(wd[k] >> 32);
                    const uint32_t oi = lo & 0xffffu, oj = lo >> 16, oe = hi & 0xffffu;
                    double *pi = reinterpret_cast<double *>(smem + oi), *pj = reinterpret_cast<double *>(smem + oj);
                    double *li = reinterpret_cast<double *>(smem + lam_off + oi), *lj = reinterpret_cast<double *>(smem + lam_off + oj);
                    const double2 t = *reinterpret_cast<const double2 *>(reinterpret_cast<const unsigned char *>(cs) + oe);
                    const bool neg = (hi & 0x80000000u) != 0u;
                    const double sn = neg ? -t.y : t.y;
                    const double u1 = *pi, v1 = *pj, lu = *li, lv = *lj;
                    const double g = lu * v1 - lv * u1, gs = neg ? -g : g;
                    const uint32_t oe0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)oe);
                    if (__all(oe == oe0)) {
                        const double rowsum = sec_wave_sum63(gs);
                        if (lane == 63) w[oe0 >> 4] += rowsum;
                    } else {
                        __hip_atomic_fetch_add(&w[oe >> 4], gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    *pi = t.x * u1 - sn * v1;
                    *pj = t.x * v1 + sn * u1;
                    *li = t.x * lu - sn * lv;
                    *lj = t.x * lv + sn * lu;
                    asm volatile("" ::: "memory");
                    const int nr = r - 8 - k;
                    wd[k] = rp[(size_t)max(nr, 0) * 64];
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < A.ntab; e += NT) {
            const SmallRot sr = tabrots[e];
            if (sr.pidx >= 0) __hip_atomic_fetch_add(&gk[sr.pidx], 2.0 * sr.coeff * w[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        for (int k = tid; k < A.K; k += NT) grads[b * A.K + k] = gk[k];
        __syncthreads();
    }
}

}  // namespace ovqe
