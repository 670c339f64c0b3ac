// sv_sector.hpp — sector path: energies of a real-amplitude program on the SUPPORT of its states, registers of 18+ qubits.
//
// A particle-number / spin conserving ansatz on a Hartree-Fock determinant keeps the state inside one symmetry sector: a few
// percent of the register (24 qubits, 5 alpha + 5 beta electrons in 12 orbitals: 627 264 of 16.8 M amplitudes).  Nothing about
// fermions is assumed here — the support is whatever the dense kernels leave non-zero at a generic parameter vector — but
// that is the case this path is made for.  The support-compacted kernel of sv_sparse.hpp does this for supports that fit ONE
// workgroup's LDS (<= 4096 amplitudes); here the support is large, so the compact state lives in HBM / the caches and is
// processed in TILES as sv_tile.hpp does with the dense state:
//
//   * circuit: consecutive ops whose mixing bits fit a set S of M index bits form a sweep; the support is sorted by
//     (index bits outside S, index bits inside S), so the amplitudes of one dense tile are CONTIGUOUS — a few thousand doubles
//     instead of 2^M.  One workgroup gathers its tile from the previous sweep's order, applies the sweep's ops from
//     precomputed pair lists (slot_i, slot_j, sign, pattern) in LDS and writes the tile back contiguously.  Compact tiles are
//     small, so M is 16 where the dense tiles stop at 12-13: a third of the sweeps, and 27x less data per sweep.
//   * <H>: the Hamiltonian restricted to the support is MATERIALISED once per (program, Hamiltonian): for every sweep of a
//     tile cover of the x-groups and every tile, the list of (slot_i, slot_j, H_ij) with H_ij != 0.  An evaluation streams
//     that list once against the tile's amplitudes in LDS: HBM-bound, no Pauli arithmetic left.  The matrix elements of the
//     x-groups with three or more mixing bits (double excitations: 96 % of the elements) take few distinct magnitudes per
//     sweep, so they are stored as ONE 32-bit word (slot_i, slot_j, sign, index into the sweep's dictionary of magnitudes:
//     4 bytes per element); the others (diagonal, single-excitation-like groups: occupation-dependent values) keep an
//     explicit double (12 bytes).  <H> tiles are smaller than the circuit's (<= 1023 amplitudes: 10-bit slots).
//
// All tables are built on the device (radix sort of the permuted indices, binary search of partners inside a tile).
#pragma once
#include "sv_kernels.hpp"

namespace ovqe {

// pair word, sb = slot bits of the engine (13, 14 or 15: tiles of up to 2^sb - 1 amplitudes; the all-ones slot means "partner
// outside the support"): slot_i | slot_j << sb | sign << 2 sb | pattern << (2 sb + 1), i.e. 32 / 8 / 2 active patterns per
// OP_TAB op;  u' = c u + s v, v' = c v - s u, s = sign ? -sin : sin
constexpr int SEC_MAX_PAT = 32;
constexpr uint32_t SEC_TILE_LDS_CAP = 14000;      // amplitudes of a circuit tile that fit LDS next to the staging buffers
constexpr uint32_t SEC_STAGE_WORDS = 4096;        // pair words per staging buffer
struct SecBuildOp {   // one compact op = one OP_TAB op, or one rotation of an OP_PAIR run
    uint64_t x;       // mixing mask
    uint64_t zs;      // sign = parity(i & zs) ^ flip, i = the pair's member that matches the pattern
    int32_t pat0, npat;
    int32_t flip, tab0;  // rotation table entries tab0 + pattern
};
struct SecPat {
    uint64_t pm, pv;  // i is the first member of an active pair when (i & pm) == pv
};
struct SecGroup {     // x-group of the Hamiltonian, global masks
    uint64_t x;
    int32_t t0, t1;
};
constexpr int SEC_HSLOT_BITS = 10;
constexpr uint32_t SEC_HSLOT_MASK = (1u << SEC_HSLOT_BITS) - 1u;
constexpr uint32_t SEC_HMAX_TILE = SEC_HSLOT_MASK;   // entries per <H> tile
constexpr int SEC_DICT_MAX = 2048;                   // magnitudes per sweep (11 bits of the coded word)
constexpr int SEC_CODED_MIN_WEIGHT = 3;              // x-groups with at least this many mixing bits go through the dictionary
// coded word: slot_i | slot_j << 10 | sign << 20 | dictionary index << 21;  explicit word: slot_i | slot_j << 10
struct SecHSweep {    // one sweep of the materialised <H>: device pointers
    const uint32_t *src;    // [K] position of the entry in the circuit's final order
    const uint32_t *off;    // [ntiles + 1]
    const uint32_t *cbase;  // [K + 1] first coded element of every entry
    const uint32_t *cwords;
    const double *cvals;    // the coded stream's values when the sweep has no dictionary (ndict == 0), else unused
    const double *dict;     // [ndict] magnitudes (ascending)
    const uint32_t *xbase;  // [K + 1] first explicit element of every entry
    const uint32_t *xwords;
    const double *xvals;    // H_ii, or 2 H_ij (pair counted once)
    int32_t ndict, ntiles;
};

__device__ __forceinline__ uint32_t sec_pext(uint32_t v, uint32_t mask) {  // mask is wave-uniform
    uint32_t r = 0;
    int k = 0;
    while (mask) {
        const int p = __ffs((int)mask) - 1;
        r |= ((v >> p) & 1u) << k;
        ++k;
        mask &= mask - 1u;
    }
    return r;
}

// ---- support of a real state ------------------------------------------------------------------------------------------
constexpr int SEC_NZ_PER_THREAD = 16;
__global__ __launch_bounds__(256) void k_sec_count_nz(const double *__restrict__ st, uint64_t namps, uint32_t *__restrict__ counts) {
    __shared__ uint32_t w[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * SEC_NZ_PER_THREAD;
    uint32_t c = 0;
    for (int k = 0; k < SEC_NZ_PER_THREAD; ++k)
        if (i0 + k < namps && st[i0 + k] != 0.0) ++c;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
// ascending indices of the non-zero amplitudes; base[b] = exclusive sum of counts
__global__ __launch_bounds__(256) void k_sec_fill_nz(const double *__restrict__ st, uint64_t namps, const uint32_t *__restrict__ base,
                                                     uint32_t *__restrict__ sup) {
    __shared__ uint32_t w[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * SEC_NZ_PER_THREAD;
    uint32_t c = 0;
    for (int k = 0; k < SEC_NZ_PER_THREAD; ++k)
        if (i0 + k < namps && st[i0 + k] != 0.0) ++c;
    uint32_t incl = c;   // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t pos = base[blockIdx.x] + incl - c;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) pos += w[k];
    for (int k = 0; k < SEC_NZ_PER_THREAD; ++k)
        if (i0 + k < namps && st[i0 + k] != 0.0) sup[pos++] = (uint32_t)(i0 + k);
}

// ---- layouts ----------------------------------------------------------------------------------------------------------
// key = (index bits outside the tile set) << M | (index bits inside): sorting the keys sorts the support by tile
__global__ __launch_bounds__(256) void k_sec_keys(const uint32_t *__restrict__ sup, uint32_t K, uint32_t smask, uint32_t outside,
                                                  int M, uint32_t *__restrict__ keys, uint32_t *__restrict__ ids) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= K) return;
    const uint32_t i = sup[e];
    keys[e] = (sec_pext(i, outside) << M) | sec_pext(i, smask);
    ids[e] = e;
}
__global__ __launch_bounds__(256) void k_sec_offsets(const uint32_t *__restrict__ keys, uint32_t K, int M, uint32_t ntiles,
                                                     uint32_t *__restrict__ off) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= K) return;
    const int64_t t = keys[e] >> M, tp = e ? (int64_t)(keys[e - 1] >> M) : -1;
    for (int64_t tt = tp + 1; tt <= t; ++tt) off[tt] = e;
    if (e == K - 1)
        for (int64_t tt = t + 1; tt <= (int64_t)ntiles; ++tt) off[tt] = K;
}
__global__ __launch_bounds__(256) void k_sec_inverse(const uint32_t *__restrict__ cid, uint32_t K, uint32_t *__restrict__ inv) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) inv[cid[e]] = e;
}
__global__ __launch_bounds__(256) void k_sec_compose(const uint32_t *__restrict__ cid, const uint32_t *__restrict__ inv_prev,
                                                     uint32_t K, uint32_t *__restrict__ src) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) src[e] = inv_prev[cid[e]];
}
// per-tile maximum of the entry counts, as one number
__global__ __launch_bounds__(256) void k_sec_max_tile(const uint32_t *__restrict__ off, uint32_t ntiles, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < ntiles; t += gridDim.x * 256u) m = max(m, off[t + 1] - off[t]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__device__ __forceinline__ int sec_find(const uint32_t *lk, int n, uint32_t key) {  // sorted local keys of the tile
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = lk[mid];
        if (v == key) return mid;
        if (v < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// ---- pair lists of one circuit sweep ----------------------------------------------------------------------------------
// One workgroup per tile.  FILL = false: cnt[tile * nops + o] = pairs of op o in the tile;  FILL = true: the pair words at
// poff[tile * (nops + 1) + o], in ascending slot order (a fixed order: results are reproducible).  A member of an active
// pair whose partner is not in the support is recorded with the all-ones slot: its amplitude is structurally zero at that point of
// the circuit (otherwise the partner would have been populated at the probe parameters) and the sweep checks just that.
template <bool FILL, int NT>
__global__ __launch_bounds__(NT) void k_sec_pairs(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                  const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                  uint32_t smask, const SecBuildOp *__restrict__ ops, int nops,
                                                  const SecPat *__restrict__ pats, uint32_t *__restrict__ cnt,
                                                  const uint32_t *__restrict__ poff, uint32_t *__restrict__ pairs, int sb) {
    const uint32_t orphan = (1u << sb) - 1u;
    extern __shared__ uint32_t sec_lk[];
    __shared__ uint32_t wtot[NT / 64];
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) {
        if (!FILL)
            for (int o = threadIdx.x; o < nops; o += NT) cnt[(size_t)t * nops + o] = 0u;
        return;
    }
    const uint32_t lmask = (1u << M) - 1u;
    for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 0; o < nops; ++o) {
        const SecBuildOp op = ops[o];
        const uint32_t xl = sec_pext((uint32_t)op.x, smask);
        const uint32_t base = FILL ? poff[(size_t)t * (nops + 1) + o] : 0u;
        uint32_t running = 0;
        for (int k0 = 0; k0 < n; k0 += NT) {
            const int k = k0 + (int)threadIdx.x;
            bool emit = false;
            uint32_t word = 0;
            if (k < n) {
                const uint64_t i = sup[cid[e0 + k]];
                for (int p = 0; p < op.npat; ++p) {
                    const SecPat pt = pats[op.pat0 + p];
                    const uint64_t bits = i & pt.pm;
                    if (bits == pt.pv) {
                        const int sj = sec_find(sec_lk, n, sec_lk[k] ^ xl);
                        const uint32_t sign = (uint32_t)(parity64(i & op.zs) ^ op.flip);
                        word = (uint32_t)k | ((sj < 0 ? orphan : (uint32_t)sj) << sb) | (sign << (2 * sb)) | ((uint32_t)p << (2 * sb + 1));
                        emit = true;
                        break;
                    }
                    if (bits == (pt.pv ^ (op.x & pt.pm))) {   // second member: only its orphans are recorded
                        if (sec_find(sec_lk, n, sec_lk[k] ^ xl) < 0) {
                            word = (uint32_t)k | (orphan << sb) | ((uint32_t)p << (2 * sb + 1));
                            emit = true;
                        }
                        break;
                    }
                }
            }
            const uint64_t bal = __ballot(emit);
            if (lane == 0) wtot[wave] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < NT / 64; ++w) {
                if (w < wave) before += wtot[w];
                total += wtot[w];
            }
            if (FILL && emit) pairs[base + running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = word;
            running += total;
            __syncthreads();
        }
        if (!FILL && threadIdx.x == 0) cnt[(size_t)t * nops + o] = running;
    }
}

// ---- one sweep of the circuit -----------------------------------------------------------------------------------------
// in: the compact state in the previous sweep's order (src = position of every entry there; nullptr: |hf> at hf_pos);
// out: the state in this sweep's order.  The chain pair offsets -> pair word -> cos/sin -> amplitudes would cost three
// dependent trips to L2 per op (measured: 1.7 us per op), so the tile's pair offsets and the sweep's cos/sin table are
// staged in LDS up front and the first pair word of op o + 1 is fetched while op o rotates its pairs.
// NT == 64: one wave owns the tile and the LDS unit keeps its accesses in order, so no barrier separates the ops.
struct SecOpLds {
    uint32_t p0;    // first pair of the op in this tile
    int32_t tab;    // first table entry, relative to the sweep's table
};
template <int NT>
__device__ __forceinline__ void sec_rotate(double *tile, const double2 *tab, uint32_t pw, bool &bad, int sb) {
    const uint32_t mask = (1u << sb) - 1u;
    const uint32_t si = pw & mask, sj = (pw >> sb) & mask;
    if (sj == mask) {
        bad |= tile[si] != 0.0;
        return;
    }
    const double2 r = tab[pw >> (2 * sb + 1)];
    const double s = ((pw >> (2 * sb)) & 1u) ? -r.y : r.y;
    const double u = tile[si], v = tile[sj];
    tile[si] = r.x * u + s * v;
    tile[sj] = r.x * v - s * u;
}
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_sweep(const double *__restrict__ in, double *__restrict__ out,
                                                     const uint32_t *__restrict__ src, const uint32_t *__restrict__ off,
                                                     const int32_t *__restrict__ tab0, int nops,
                                                     const uint32_t *__restrict__ poff, const uint32_t *__restrict__ pairs,
                                                     const RotParam *__restrict__ rp, int rot0, int nrot, uint32_t tile_cap,
                                                     uint32_t hf_pos, int *__restrict__ flag, int dbg, int sb) {
    // The pair words of the tile are consecutive in memory (op after op): they are staged in LDS in op-aligned chunks of
    // at most W words, the loads of chunk c + 1 in flight (registers) while chunk c is processed — an op never waits for
    // global memory.  Barriers wait for LDS only (s_waitcnt lgkmcnt), not for those loads.
    constexpr uint32_t W = SEC_STAGE_WORDS;
    constexpr int SEC_WORDS_PER_THREAD = SEC_STAGE_WORDS / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    double2 *cs = reinterpret_cast<double2 *>(tile + ((tile_cap + 1u) & ~1u));
    SecOpLds *lop = reinterpret_cast<SecOpLds *>(cs + nrot);
    uint32_t *wbuf = reinterpret_cast<uint32_t *>(lop + nops + 2);
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) return;
    const uint32_t *po = poff + (size_t)t * (nops + 1);
    for (int o = threadIdx.x; o <= nops; o += NT) lop[o] = SecOpLds{po[o], o < nops ? tab0[o] - rot0 : 0};
    for (int r = threadIdx.x; r < nrot; r += NT) {
        const RotParam rr = rp[rot0 + r];
        cs[r] = make_double2(rr.c, rr.s);
    }
    if (src) {
        for (int k = threadIdx.x; k < n; k += NT) tile[k] = in[src[e0 + k]];
    } else {
        for (int k = threadIdx.x; k < n; k += NT) tile[k] = (e0 + (uint32_t)k == hf_pos) ? 1.0 : 0.0;
    }
    __syncthreads();
    auto chunk_end = [&](int oa) {   // ops [oa, ob) whose pair words fit one buffer (ob == oa: op oa alone is larger)
        int ob = oa;
        const uint32_t pa = lop[oa].p0;
        while (ob < nops && lop[ob + 1].p0 - pa <= W) ++ob;
        return ob;
    };
    uint32_t regs[SEC_WORDS_PER_THREAD];
    auto fetch = [&](int oa, int ob) {
        const uint32_t base = lop[oa].p0, cnt = lop[ob].p0 - base;
#pragma unroll
        for (int r = 0; r < SEC_WORDS_PER_THREAD; ++r) {
            const uint32_t idx = threadIdx.x + (uint32_t)r * NT;
            regs[r] = idx < cnt ? pairs[base + idx] : 0u;
        }
    };
    auto stash = [&](uint32_t *buf) {
#pragma unroll
        for (int r = 0; r < SEC_WORDS_PER_THREAD; ++r) buf[threadIdx.x + (uint32_t)r * NT] = regs[r];
    };
    bool bad = false;
    int oa = 0, ob = chunk_end(0), cb = 0;
    if (dbg == 1) oa = ob = nops;   // measurement only: the sweep without its ops
    if (ob > oa) {
        fetch(oa, ob);
        stash(wbuf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    while (oa < nops) {
        if (ob == oa) {   // one op with more pairs in this tile than a buffer holds: straight from memory
            const uint32_t p0 = lop[oa].p0, p1 = lop[oa + 1].p0;
            const double2 *tab = cs + lop[oa].tab;
            for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) sec_rotate<NT>(tile, tab, pairs[k], bad, sb);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ++oa;
            if (oa < nops) {
                ob = chunk_end(oa);
                if (ob > oa) {
                    fetch(oa, ob);
                    stash(wbuf + (size_t)cb * W);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
            continue;
        }
        const int na = ob, nb = na < nops ? chunk_end(na) : na;
        if (nb > na) fetch(na, nb);   // in flight while this chunk is processed
        const uint32_t *wb = wbuf + (size_t)cb * W;
        const uint32_t base = lop[oa].p0;
        uint32_t p0 = 0;
        for (int o = oa; o < ob; ++o) {
            const uint32_t p1 = lop[o + 1].p0 - base;
            const double2 *tab = cs + lop[o].tab;
            for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) sec_rotate<NT>(tile, tab, wb[k], bad, sb);
            p0 = p1;
            if (NT > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("" ::: "memory");
        }
        if (nb > na) {
            stash(wbuf + (size_t)(cb ^ 1) * W);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        cb ^= 1;
        oa = na;
        ob = nb;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int k = threadIdx.x; k < n; k += NT) out[e0 + k] = tile[k];
    if (bad) atomicOr(flag, 1);
}

// ---- materialised <H> -------------------------------------------------------------------------------------------------
// One thread per entry of the tile; for every x-group of the sweep D_g(i) = sum_t c_t (-1)^{|j & z_t|}, j = i ^ x, the pair
// taken from its member with a clear pivot bit.  FILL = false: ccnt[e] / xcnt[e] = non-zero elements of entry e in the two
// streams; FILL = true: they are written at cbase[e] / xbase[e] (entry-major: a wave of the evaluation kernel reads one
// amplitude for a run of elements).  The coded stream is written with its values; k_sec_encode replaces them.
// ROWS = true: the symmetric matrix row by row — every off-diagonal element under BOTH of its entries, explicit values
// not doubled (the coded stream keeps the doubled magnitudes of the pair format so that the sweep's dictionary serves
// both) — in SLICES of 64 consecutive entries of a tile: element q of the slice's row l sits at base + 64 q + l, rows padded
// to the longest of the slice with null elements (value 0).  lambda = H psi then is one gather per element, a register
// sum per row, coalesced loads, no scattered additions (k_sector_apply_rows).  cbase / xbase are per (tile, slice) then,
// clen / xlen the slice lengths.
constexpr int SEC_HSLICES = (SEC_HMAX_TILE + 64) / 64;   // slices per tile
template <bool FILL, int NT, bool ROWS = false>
__global__ __launch_bounds__(NT) void k_sec_hbuild(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                   uint32_t smask, const SecGroup *__restrict__ groups, int ngroups,
                                                   const HTerm *__restrict__ terms, uint32_t *__restrict__ ccnt,
                                                   uint32_t *__restrict__ xcnt, const uint32_t *__restrict__ cbase,
                                                   const uint32_t *__restrict__ xbase, uint32_t *__restrict__ cwords,
                                                   double *__restrict__ cvals, uint32_t *__restrict__ xwords,
                                                   double *__restrict__ xvals, const uint32_t *__restrict__ clen = nullptr,
                                                   const uint32_t *__restrict__ xlen = nullptr) {
    extern __shared__ uint32_t sec_lk[];
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) return;
    const uint32_t lmask = (1u << M) - 1u;
    for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    __syncthreads();
    const int nrows = (ROWS && FILL) ? ((n + 63) & ~63) : n;   // the lanes past the tile's last row write padding only
    for (int k = threadIdx.x; k < nrows; k += NT) {
        uint32_t cc = 0, xc = 0;
        const size_t sl = (size_t)t * SEC_HSLICES + (size_t)(k >> 6);
        const uint32_t cstride = (ROWS && FILL) ? 64u : 1u;
        uint32_t cb = 0, xb = 0;
        if (FILL) {
            cb = ROWS ? cbase[sl] + (uint32_t)(k & 63) : cbase[e0 + k];
            xb = ROWS ? xbase[sl] + (uint32_t)(k & 63) : xbase[e0 + k];
        }
        if (k < n) {
            const uint64_t i = sup[cid[e0 + k]];
            const uint32_t li = sec_lk[k];
            for (int g = 0; g < ngroups; ++g) {
                const SecGroup gr = groups[g];
                int sj = k;
                uint64_t j = i;   // the pair's member with the pivot bit set: its index enters the signs
                if (gr.x) {
                    const uint64_t pbit = 1ull << (63 - __clzll((long long)gr.x));
                    if (i & pbit) {
                        if (!ROWS) continue;
                    } else {
                        j = i ^ gr.x;
                    }
                    sj = sec_find(sec_lk, n, li ^ sec_pext((uint32_t)gr.x, smask));
                    if (sj < 0) continue;
                }
                double d = 0.0;
                for (int tt = gr.t0; tt < gr.t1; ++tt) {
                    const HTerm ht = terms[tt];
                    if (__popcll(gr.x & ht.z) & 1) continue;   // odd number of Y: <P> = 0 on a real state
                    d += parity64(j & ht.z) ? -ht.cr : ht.cr;
                }
                if (d == 0.0) continue;
                const uint32_t word = (uint32_t)k | ((uint32_t)sj << SEC_HSLOT_BITS);
                if (__popcll(gr.x) >= SEC_CODED_MIN_WEIGHT) {
                    if (FILL) {
                        cwords[cb + cc * cstride] = word;
                        cvals[cb + cc * cstride] = 2.0 * d;
                    }
                    ++cc;
                } else {
                    if (FILL) {
                        xwords[xb + xc * cstride] = word;
                        xvals[xb + xc * cstride] = (gr.x && !ROWS) ? 2.0 * d : d;
                    }
                    ++xc;
                }
            }
        }
        if (!FILL) {
            ccnt[e0 + k] = cc;
            xcnt[e0 + k] = xc;
        } else if (ROWS) {   // pad the row to the slice's length
            for (uint32_t q = cc; q < clen[sl]; ++q) {
                cwords[cb + q * 64u] = 0u;
                cvals[cb + q * 64u] = 0.0;
            }
            for (uint32_t q = xc; q < xlen[sl]; ++q) {
                xwords[xb + q * 64u] = 0u;
                xvals[xb + q * 64u] = 0.0;
            }
        }
    }
}
// slice lengths of the row format: len[tile * SEC_HSLICES + s] = longest row of the slice (one wave per slice)
__global__ __launch_bounds__(256) void k_sec_slice_len(const uint32_t *__restrict__ off, const uint32_t *__restrict__ ccnt,
                                                       const uint32_t *__restrict__ xcnt, uint32_t *__restrict__ clen,
                                                       uint32_t *__restrict__ xlen, uint32_t *__restrict__ cwords64,
                                                       uint32_t *__restrict__ xwords64) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t s = wave; s < (uint32_t)SEC_HSLICES; s += 4u) {
        const uint32_t k = 64u * s + lane;
        uint32_t c = k < n ? ccnt[e0 + k] : 0u, x = k < n ? xcnt[e0 + k] : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            c = max(c, (uint32_t)__shfl_xor(c, o, 64));
            x = max(x, (uint32_t)__shfl_xor(x, o, 64));
        }
        if (lane == 0) {
            const size_t sl = (size_t)t * SEC_HSLICES + s;
            clen[sl] = c;
            xlen[sl] = x;
            cwords64[sl] = 64u * c;   // words of the slice: input of the scan that lays the slices out
            xwords64[sl] = 64u * x;
        }
    }
}
// dictionary of a sweep: sort keys = |value| (the bit pattern of a non-negative double orders like the number)
__global__ __launch_bounds__(256) void k_sec_abs_keys(const double *__restrict__ vals, uint32_t n, uint64_t *__restrict__ keys) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < n) keys[e] = (uint64_t)__double_as_longlong(fabs(vals[e]));
}
__global__ __launch_bounds__(256) void k_sec_encode(uint32_t *__restrict__ words, const double *__restrict__ vals, uint32_t n,
                                                    const uint64_t *__restrict__ dict, int ndict) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    const double v = vals[e];
    if (v == 0.0) {   // padding of the row format: the zero magnitude appended to the dictionary
        words[e] = (uint32_t)ndict << 21;
        return;
    }
    const uint64_t key = (uint64_t)__double_as_longlong(fabs(v));
    int lo = 0, hi = ndict - 1;
    while (lo < hi) {   // the key is in the dictionary
        const int mid = (lo + hi) >> 1;
        if (dict[mid] < key) lo = mid + 1; else hi = mid;
    }
    words[e] |= (v < 0.0 ? 1u << 20 : 0u) | ((uint32_t)lo << 21);
}

// E = sum over the sweeps and tiles of sum_e H_e a[slot_i] a[slot_j]: blockIdx.y = sweep; the workgroups of a sweep share
// its tiles round robin (the dictionary is staged once per workgroup, the amplitudes of the next tile are in flight while
// the elements of the current one stream)
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_expect(const double *__restrict__ state, const SecHSweep *__restrict__ sweeps,
                                                      double2 *__restrict__ partials) {
    constexpr int PER = (SEC_HMAX_TILE + NT) / NT;   // amplitudes of a tile per thread
    __shared__ double tile[SEC_HMAX_TILE + 1];
    __shared__ double dict[SEC_DICT_MAX];
    __shared__ double2 red[NT / 64];
    const SecHSweep sw = sweeps[blockIdx.y];
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = sw.dict[k];
    double nxt[PER];
    uint32_t e0 = 0, n = 0;
    auto fetch = [&](uint32_t t) {
        e0 = sw.off[t];
        n = sw.off[t + 1] - e0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t k = threadIdx.x + (uint32_t)q * NT;
            nxt[q] = k < n ? state[sw.src[e0 + k]] : 0.0;
        }
    };
    uint32_t t = blockIdx.x;
    if (t < (uint32_t)sw.ntiles) fetch(t);
    for (; t < (uint32_t)sw.ntiles; t += gridDim.x) {
        const uint32_t ce0 = e0, cn = n;
        __syncthreads();   // the previous tile's elements are done with the LDS copy
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t k = threadIdx.x + (uint32_t)q * NT;
            if (k < cn) tile[k] = nxt[q];
        }
        __syncthreads();
        if (t + gridDim.x < (uint32_t)sw.ntiles) fetch(t + gridDim.x);
        if (cn == 0) continue;
        {
            const uint32_t b1 = sw.cbase[ce0 + cn];
            uint32_t e = sw.cbase[ce0] + threadIdx.x;
            if (sw.ndict) {
                auto term = [&](uint32_t w) {
                    const double v = dict[w >> 21] * tile[w & SEC_HSLOT_MASK] * tile[(w >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
                    return (w & (1u << 20)) ? -v : v;
                };
                for (; e + 7u * NT < b1; e += 8u * NT) {
                    uint32_t w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = __builtin_nontemporal_load(&sw.cwords[e + q * NT]);
                    acc0 += term(w[0]) + term(w[4]);
                    acc1 += term(w[1]) + term(w[5]);
                    acc2 += term(w[2]) + term(w[6]);
                    acc3 += term(w[3]) + term(w[7]);
                }
                for (; e < b1; e += NT) acc0 += term(sw.cwords[e]);
            } else {
                for (; e < b1; e += NT) {
                    const uint32_t w = sw.cwords[e];
                    acc0 += sw.cvals[e] * tile[w & SEC_HSLOT_MASK] * tile[(w >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
                }
            }
        }
        {
            const uint32_t b1 = sw.xbase[ce0 + cn];
            uint32_t e = sw.xbase[ce0] + threadIdx.x;
            for (; e + NT < b1; e += 2u * NT) {
                const uint32_t w0 = sw.xwords[e], w1 = sw.xwords[e + NT];
                const double v0 = sw.xvals[e], v1 = sw.xvals[e + NT];
                acc1 += v0 * tile[w0 & SEC_HSLOT_MASK] * tile[(w0 >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
                acc2 += v1 * tile[w1 & SEC_HSLOT_MASK] * tile[(w1 >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
            }
            for (; e < b1; e += NT) {
                const uint32_t w0 = sw.xwords[e];
                acc3 += sw.xvals[e] * tile[w0 & SEC_HSLOT_MASK] * tile[(w0 >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
            }
        }
    }
    __syncthreads();
    const double2 tsum = block_sum<NT>(make_double2((acc0 + acc1) + (acc2 + acc3), 0.0), red);
    if (threadIdx.x == 0) partials[slot] = tsum;
}

// ---- exact gradient on the sector tables (adjoint method) ----------------------------------------------------------------
// lambda = H psi restricted to the support, from the same element streams as k_sector_expect: an off-diagonal element
// (pair counted once, value 2 H_ij) adds H_ij a_j to lambda_i and H_ij a_i to lambda_j — two f64 LDS atomics per element on
// an LDS copy of the tile (the lanes of a wave mostly share slot_i in entry-major order: the LDS unit serialises those);
// the tile's result is added to lambda in the circuit's final order with f64 atomics.  The order of these additions is not
// fixed: the gradient reproduces to rounding, not bit for bit.
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_apply(const double *__restrict__ state, const SecHSweep *__restrict__ sweeps,
                                                     double *__restrict__ lam_out) {
    __shared__ double tile[SEC_HMAX_TILE + 1];
    __shared__ double lam[SEC_HMAX_TILE + 1];
    __shared__ double dict[SEC_DICT_MAX];
    const SecHSweep sw = sweeps[blockIdx.y];
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = sw.dict[k];
    auto add = [&](uint32_t w, double hv) {   // hv = H_ij (off-diagonal) or H_ii
        const uint32_t si = w & SEC_HSLOT_MASK, sj = (w >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK;
        const double ai = tile[si], aj = tile[sj];
        __hip_atomic_fetch_add(&lam[si], hv * aj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (si != sj) __hip_atomic_fetch_add(&lam[sj], hv * ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (uint32_t t = blockIdx.x; t < (uint32_t)sw.ntiles; t += gridDim.x) {
        const uint32_t e0 = sw.off[t], n = sw.off[t + 1] - e0;
        if (n == 0) continue;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += NT) {
            tile[k] = state[sw.src[e0 + k]];
            lam[k] = 0.0;
        }
        __syncthreads();
        {
            const uint32_t b1 = sw.cbase[e0 + n];
            uint32_t e = sw.cbase[e0] + threadIdx.x;
            if (sw.ndict) {
                for (; e + 3u * NT < b1; e += 4u * NT) {
                    uint32_t w[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = __builtin_nontemporal_load(&sw.cwords[e + q * NT]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double v = 0.5 * dict[w[q] >> 21];
                        add(w[q], (w[q] & (1u << 20)) ? -v : v);
                    }
                }
                for (; e < b1; e += NT) {
                    const uint32_t w = sw.cwords[e];
                    const double v = 0.5 * dict[w >> 21];
                    add(w, (w & (1u << 20)) ? -v : v);
                }
            } else {
                for (; e < b1; e += NT) add(sw.cwords[e], 0.5 * sw.cvals[e]);
            }
        }
        {
            const uint32_t b1 = sw.xbase[e0 + n];
            for (uint32_t e = sw.xbase[e0] + threadIdx.x; e < b1; e += NT) {
                const uint32_t w = sw.xwords[e];
                const bool diag = (w & SEC_HSLOT_MASK) == ((w >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK);
                add(w, diag ? sw.xvals[e] : 0.5 * sw.xvals[e]);
            }
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += NT) unsafeAtomicAdd(&lam_out[sw.src[e0 + k]], lam[k]);
    }
}
// lambda = H psi from the row format: one wave per slice, lane = row; per element a coalesced word, the dictionary and
// a_j from LDS, a register sum; the row's result goes straight to lambda in the circuit's final order (an f64 atomic: the
// sweeps of the cover add up there).  Gradients use this when the row tables fit the budget, else k_sector_apply.
struct SecHRows {
    const uint32_t *src, *off;
    const uint32_t *cbase, *clen, *cwords;   // [ntiles * SEC_HSLICES] slice bases / lengths; coded words
    const double *cvals, *dict;              // cvals: sweeps without a dictionary
    const uint32_t *xbase, *xlen, *xwords;
    const double *xvals;
    int32_t ndict, ntiles;
};
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_apply_rows(const double *__restrict__ state, const SecHRows *__restrict__ sweeps,
                                                          double *__restrict__ lam_out) {
    constexpr int NW = NT / 64;
    __shared__ double tile[SEC_HMAX_TILE + 1];
    __shared__ double dict[SEC_DICT_MAX + 1];
    const SecHRows sw = sweeps[blockIdx.y];
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = 0.5 * sw.dict[k];   // doubled magnitudes -> H_ij
    if (threadIdx.x == 0) dict[sw.ndict] = 0.0;                                  // the null element
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t t = blockIdx.x; t < (uint32_t)sw.ntiles; t += gridDim.x) {
        const uint32_t e0 = sw.off[t], n = sw.off[t + 1] - e0;
        if (n == 0) continue;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += NT) tile[k] = state[sw.src[e0 + k]];
        __syncthreads();
        for (uint32_t s = wave; 64u * s < n; s += NW) {
            const size_t sl = (size_t)t * SEC_HSLICES + s;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            {
                const uint32_t L = sw.clen[sl];
                const uint32_t *wp = sw.cwords + sw.cbase[sl] + lane;
                if (sw.ndict) {
                    auto term = [&](uint32_t w) {
                        const double v = dict[w >> 21] * tile[(w >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
                        return (w & (1u << 20)) ? -v : v;
                    };
                    uint32_t q = 0;
                    for (; q + 7u < L; q += 8u) {
                        uint32_t w[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) w[u] = __builtin_nontemporal_load(wp + 64u * (q + (uint32_t)u));
                        a0 += term(w[0]) + term(w[4]);
                        a1 += term(w[1]) + term(w[5]);
                        a2 += term(w[2]) + term(w[6]);
                        a3 += term(w[3]) + term(w[7]);
                    }
                    for (; q < L; ++q) a0 += term(wp[64u * q]);
                } else {
                    const double *vp = sw.cvals + sw.cbase[sl] + lane;
                    for (uint32_t q = 0; q < L; ++q) a0 += 0.5 * vp[64u * q] * tile[(wp[64u * q] >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
                }
            }
            {
                const uint32_t L = sw.xlen[sl];
                const uint32_t *wp = sw.xwords + sw.xbase[sl] + lane;
                const double *vp = sw.xvals + sw.xbase[sl] + lane;
                for (uint32_t q = 0; q < L; ++q) a1 += vp[64u * q] * tile[(wp[64u * q] >> SEC_HSLOT_BITS) & SEC_HSLOT_MASK];
            }
            const uint32_t row = 64u * s + lane;
            if (row < n) unsafeAtomicAdd(&lam_out[sw.src[e0 + row]], (a0 + a1) + (a2 + a3));
        }
    }
}
// <a|b> over the compact state: partials per workgroup (fixed order)
__global__ __launch_bounds__(256) void k_sec_dot(const double *__restrict__ a, const double *__restrict__ b, uint32_t K,
                                                 double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) acc += a[k] * b[k];
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// wave-wide sum by DPP row operations (no LDS crossbar: __shfl_xor on doubles costs two ds_bpermute per step); the total
// is valid in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double sec_dpp_add(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sec_wave_sum63(double v) {
    v = sec_dpp_add<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v = sec_dpp_add<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v = sec_dpp_add<0x141, 0xf>(v);   // row_half_mirror
    v = sec_dpp_add<0x140, 0xf>(v);   // row_mirror
    v = sec_dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = sec_dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// One sweep of the circuit BACKWARDS on psi and lambda together (both in this sweep's order): for every op, last to
// first, w[entry] += sum over its pairs of sigma (lambda_i psi_j - lambda_j psi_i) on the states after the op (dE/dtheta =
// 2 coeff w), then both states are rotated back.  Output in the PREVIOUS sweep's order (scatter through src; the first
// sweep needs no output).  Partial sums: one per wave and table entry in LDS, added per tile in wave order, then over the
// tiles by k_sec_reduce_w — a fixed order.
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_adjoint(const double *__restrict__ psi_in, const double *__restrict__ lam_in,
                                                       double *__restrict__ psi_out, double *__restrict__ lam_out,
                                                       const uint32_t *__restrict__ src, const uint32_t *__restrict__ off,
                                                       const int32_t *__restrict__ tab0, int nops,
                                                       const uint32_t *__restrict__ poff, const uint32_t *__restrict__ pairs,
                                                       const RotParam *__restrict__ rp, int rot0, int nrot, uint32_t tile_cap,
                                                       double *__restrict__ wpart, int wstride, int sb) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    const uint32_t capp = (tile_cap + 1u) & ~1u;
    double *tp = reinterpret_cast<double *>(sec_smem);
    double *tl = tp + capp;
    double2 *cs = reinterpret_cast<double2 *>(tl + capp);
    double *wacc = reinterpret_cast<double *>(cs + nrot);          // [NW][nrot]
    SecOpLds *lop = reinterpret_cast<SecOpLds *>(wacc + (size_t)NW * nrot);
    uint32_t *wbuf = reinterpret_cast<uint32_t *>(lop + nops + 2);
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    double *wp = wpart + (size_t)t * wstride + rot0;   // [tile][table entry], zeroed by the host
    if (n == 0) return;
    const uint32_t *po = poff + (size_t)t * (nops + 1);
    for (int o = threadIdx.x; o <= nops; o += NT) lop[o] = SecOpLds{po[o], o < nops ? tab0[o] - rot0 : 0};
    for (int r = threadIdx.x; r < nrot; r += NT) {
        const RotParam rr = rp[rot0 + r];
        cs[r] = make_double2(rr.c, rr.s);
    }
    for (int r = threadIdx.x; r < NW * nrot; r += NT) wacc[r] = 0.0;
    for (int k = threadIdx.x; k < n; k += NT) {
        tp[k] = psi_in[e0 + k];
        tl[k] = lam_in[e0 + k];
    }
    __syncthreads();
    const uint32_t mask = (1u << sb) - 1u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *mine = wacc + (size_t)wave * nrot;
    // pair words staged in LDS in op-aligned chunks as in k_sector_sweep, walked from the last op to the first
    constexpr uint32_t W = SEC_STAGE_WORDS;
    constexpr int WPT = SEC_STAGE_WORDS / NT;
    auto chunk_begin = [&](int ob) {   // ops [oa, ob) whose pair words fit one buffer (oa == ob: op ob - 1 alone is larger)
        int oa = ob;
        const uint32_t pb = lop[ob].p0;
        while (oa > 0 && pb - lop[oa - 1].p0 <= W) --oa;
        return oa;
    };
    uint32_t regs[WPT];
    auto fetch = [&](int oa, int ob) {
        const uint32_t base = lop[oa].p0, cnt = lop[ob].p0 - base;
#pragma unroll
        for (int r = 0; r < WPT; ++r) {
            const uint32_t idx = threadIdx.x + (uint32_t)r * NT;
            regs[r] = idx < cnt ? pairs[base + idx] : 0u;
        }
    };
    auto stash = [&](uint32_t *buf) {
#pragma unroll
        for (int r = 0; r < WPT; ++r) buf[threadIdx.x + (uint32_t)r * NT] = regs[r];
    };
    // one op backwards: gradient sums of its (at most few) patterns, then both states rotated back
    auto back_op = [&](int o, const uint32_t *words, uint32_t p0, uint32_t p1) {
        const int tb = lop[o].tab;
        const int npat = (o + 1 < nops ? lop[o + 1].tab : nrot) - tb;   // the table entries of an op are consecutive
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) {
            const uint32_t pw = words[k];
            const uint32_t si = pw & mask, sj = (pw >> sb) & mask;
            if (sj == mask) continue;
            const uint32_t pat = pw >> (2 * sb + 1);
            const double2 r = cs[tb + pat];
            const bool neg = (pw >> (2 * sb)) & 1u;
            const double s = neg ? -r.y : r.y;
            const double u1 = tp[si], v1 = tp[sj], lu = tl[si], lv = tl[sj];
            const double g = lu * v1 - lv * u1;
            const double gs = neg ? -g : g;
            if (pat < 8) {
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p] += (pat == (uint32_t)p) ? gs : 0.0;
            } else {
                __hip_atomic_fetch_add(&mine[tb + pat], gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // rare
            }
            tp[si] = r.x * u1 - s * v1;
            tp[sj] = r.x * v1 + s * u1;
            tl[si] = r.x * lu - s * lv;
            tl[sj] = r.x * lv + s * lu;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p < npat) {   // wave-uniform
                const double tsum = sec_wave_sum63(acc[p]);
                if (lane == 63 && tsum != 0.0) mine[tb + p] += tsum;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    int ob = nops, oa = chunk_begin(nops), cb = 0;
    if (oa < ob) {
        fetch(oa, ob);
        stash(wbuf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    while (ob > 0) {
        if (oa == ob) {   // one op with more pairs in this tile than a buffer holds: straight from memory
            back_op(ob - 1, pairs, lop[ob - 1].p0, lop[ob].p0);
            --ob;
            if (ob > 0) {
                oa = chunk_begin(ob);
                if (oa < ob) {
                    fetch(oa, ob);
                    stash(wbuf + (size_t)cb * W);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            } else {
                oa = 0;
            }
            continue;
        }
        const int nb = oa, na = nb > 0 ? chunk_begin(nb) : nb;
        if (na < nb) fetch(na, nb);   // in flight while this chunk is processed
        const uint32_t *wb = wbuf + (size_t)cb * W;
        const uint32_t base = lop[oa].p0;
        for (int o = ob - 1; o >= oa; --o) back_op(o, wb, lop[o].p0 - base, lop[o + 1].p0 - base);
        if (na < nb) {
            stash(wbuf + (size_t)(cb ^ 1) * W);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        cb ^= 1;
        ob = nb;
        oa = na;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < nrot; r += NT) {
        double tsum = 0.0;
        for (int w = 0; w < NW; ++w) tsum += wacc[(size_t)w * nrot + r];
        wp[r] = tsum;
    }
    if (src) {
        for (int k = threadIdx.x; k < n; k += NT) {
            const uint32_t d = src[e0 + k];
            psi_out[d] = tp[k];
            lam_out[d] = tl[k];
        }
    }
}
// w[r] = sum over the tiles of wpart[tile][r] (fixed order), r over the whole angle table
__global__ __launch_bounds__(256) void k_sec_reduce_w(const double *__restrict__ wpart, uint32_t ntiles, int nrot,
                                                      double *__restrict__ w) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrot) return;
    double tsum = 0.0;
    for (uint32_t t = 0; t < ntiles; ++t) tsum += wpart[(size_t)t * nrot + r];
    w[r] = tsum;
}

// ---- Lanczos on the support (ovqe_sector_ground_state): vectors of K doubles in the circuit's final order --------------
__global__ __launch_bounds__(256) void k_sec_randomize(double *__restrict__ v, uint32_t K, uint64_t seed, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        uint64_t z = seed + 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1u);   // splitmix64
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        const double x = (double)(int64_t)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;   // [-1, 1)
        v[k] = x;
        acc += x * x;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void k_sec_scale(double *__restrict__ v, uint32_t K, double a) {
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) v[k] *= a;
}
// y = a x (first) or y += a x
__global__ __launch_bounds__(256) void k_sec_axpy(double *__restrict__ y, const double *__restrict__ x, double a, uint32_t K, int first) {
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) y[k] = first ? a * x[k] : y[k] + a * x[k];
}
// w -= alpha v + beta vprev; partials = |w|^2
__global__ __launch_bounds__(256) void k_sec_lanczos_update(double *__restrict__ w, const double *__restrict__ v,
                                                            const double *__restrict__ vprev, double alpha, double beta, uint32_t K,
                                                            double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        double x = w[k] - alpha * v[k];
        if (vprev) x -= beta * vprev[k];
        w[k] = x;
        acc += x * x;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// the compact state back in canonical (ascending index) order, e.g. for ovqe_get_state-like consumers and tests
__global__ __launch_bounds__(256) void k_sec_scatter(const double *__restrict__ in, const uint32_t *__restrict__ cid, uint32_t K,
                                                     double *__restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) out[cid[e]] = in[e];
}

}  // namespace ovqe
