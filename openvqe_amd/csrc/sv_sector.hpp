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
//     tile cover of the x-groups and every tile, the matrix elements H_ij != 0 between its entries.  An evaluation streams
//     that list once against the tile's amplitudes in LDS: HBM-bound, no Pauli arithmetic left.  The matrix elements of the
//     x-groups with three or more mixing bits (double excitations: 96 % of the elements) take few distinct magnitudes per
//     sweep, so they are stored as ONE 32-bit word (slot_j, sign, index into the sweep's dictionary of magnitudes: 4 bytes
//     per element); the others (diagonal, single-excitation-like groups: occupation-dependent values) keep an explicit
//     double (12 bytes).  Elements are kept row by row in slices of 64 entries (lane = row), see "ROW format" below.
//
// All tables are built on the device (radix sort of the permuted indices; partners inside a tile from a dense LDS map
// local key -> slot, or a binary search in the tile's sorted keys when the tile's index bits exceed 16).
// The same tables serve ovqe_energy_gradient (lambda = H psi + backward sweeps) and ovqe_sector_ground_state (Lanczos).
#pragma once
#include "sv_kernels.hpp"
#include "sv_regular_host.hpp"

namespace ovqe {

// pair word, sb = slot bits of the engine (13, 14 or 15: tiles of up to 2^sb - 1 amplitudes; the all-ones slot means "partner
// outside the support"): slot_i | slot_j << sb | sign << 2 sb | pattern << (2 sb + 1), i.e. 32 / 8 / 2 active patterns per
// OP_TAB op;  u' = c u + s v, v' = c v - s u, s = sign ? -sin : sin
constexpr int SEC_MAX_PAT = 32;
constexpr uint32_t SEC_TILE_LDS_CAP = 14000;      // amplitudes of a circuit tile that fit LDS next to the staging buffers
constexpr uint32_t SEC_STAGE_WORDS = 4096;        // pair words per staging buffer
struct SecGroup {     // x-group of the Hamiltonian, global masks
    uint64_t x;
    int32_t t0, t1;
};
// ---- materialised <H>: ROW format ---------------------------------------------------------------------------------------
// Tiles of up to 8190 amplitudes (13-bit slots).  Every matrix element H_ij (i != j) is stored ONCE, under one of its two
// entries (a hash bit of the pair decides which: rows come out equally long); diagonal elements under their own entry.
// The rows of a tile are kept in SLICES of 64 rows — the tile's entries sorted by row length, so that the rows of a slice are
// equally long but for a few percent; rows are padded to the slice's longest (a multiple of 4) with null elements.  Coded
// element q of the slice's row l sits at base + 256 (q / 4) + 4 l + q % 4 — a lane reads four elements with one 16-byte
// load, a wave 1 KB contiguous; explicit element q at base + 64 q + l.  A wave owns a slice, lane = row: a_i stays in a register,
// loads are coalesced, per element one word + the dictionary + a_j from LDS.
//   coded element (x-groups with >= 3 mixing bits): word = slot_j | sign << 13 | dictionary index << 14, value 2 H_ij =
//     +- dict[index] (index == ndict: null, 0);
//   explicit element (diagonal, single-excitation-like groups): word = slot_j, value H_ii or 2 H_ij (null: 0).
constexpr int SEC_H_INFLIGHT = 4;   // 16-byte loads of coded words in flight per lane in the <H> kernels (8: slower, 4.0 against 4.4 TB/s)
constexpr int SEC_H_INFLIGHT_APPLY = 2;   // ... in k_sector_apply (lambda = H psi keeps every element's value for its LDS atomic: registers)
constexpr int SEC_HSLOT_BITS = 13;
constexpr uint32_t SEC_HSLOT_MASK = (1u << SEC_HSLOT_BITS) - 1u;
constexpr uint32_t SEC_HMAX_TILE = SEC_HSLOT_MASK - 1u;   // entries per <H> tile
constexpr int SEC_DICT_MAX = 4096;                   // magnitudes per sweep staged in LDS (the word has room for 2^18)
constexpr int SEC_CODED_MIN_WEIGHT = 3;              // x-groups with at least this many mixing bits go through the dictionary
constexpr int SEC_HSLICES = (SEC_HMAX_TILE + 64) / 64;   // slices per tile
constexpr uint32_t SEC_HTILE_LDS_CAP = 7600;             // entries per <H> tile: the lambda kernel holds two tiles + the dictionary in LDS
struct SecHSweep {    // one sweep of the materialised <H>: device pointers
    const uint32_t *src;    // [K] position of the entry in the circuit's final order
    const uint32_t *off;    // [ntiles + 1]
    const uint16_t *order;  // [K] entry (slot) at every slice position of its tile: rows sorted by length
    const uint32_t *cbase, *clen, *cwords;   // [ntiles * SEC_HSLICES] slice bases / lengths; coded words
    const double *cvals;    // the coded stream's values when the sweep has no dictionary (ndict == 0), else unused
    const double *dict;     // [ndict] magnitudes (ascending)
    const uint32_t *xbase, *xlen, *xwords;
    const double *xvals;
    int32_t ndict, ntiles;
    const uint32_t *torder;   // [ntiles] tiles by population, largest first: the order in which workgroups take them (nullptr: as numbered)
    int32_t packed;           // 1: cwords holds the coded words as 24-bit elements (k_sec_pack24: four elements of a lane in three dwords)
};
constexpr int SEC_DICT_PACKED_MAX = 1023;   // magnitudes up to which a coded word fits 24 bits (slot 13, sign 1, entry 10: the null element is entry ndict)
typedef uint32_t sec_u32x4 __attribute__((ext_vector_type(4)));
// (aligned(4): the 24-bit elements are read at a 12-byte stride from 4-byte aligned addresses — without it the type claims 16-byte
// alignment, which is undefined behaviour here and lets the compiler widen the load to a dwordx4)
typedef uint32_t sec_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
// four 24-bit elements of a lane from their three dwords (the layout of the 32-bit words with the empty top byte dropped)
__device__ __forceinline__ sec_u32x4 sec_unpack24(sec_u32x3 d) {
    sec_u32x4 w;
    w.x = d.x & 0xffffffu;
    w.y = __builtin_amdgcn_alignbit(d.y, d.x, 24) & 0xffffffu;
    w.z = __builtin_amdgcn_alignbit(d.z, d.y, 16) & 0xffffffu;
    w.w = d.z >> 8;
    return w;
}
// the coded stream of a sweep in 24-bit elements: group g = four consecutive words (one lane's) -> three dwords; a quarter fewer bytes
// for the kernels that stream the table from HBM (<H>, lambda = H psi, batches)
__global__ __launch_bounds__(256) void k_sec_pack24(const uint32_t *__restrict__ words, uint32_t ngroups, uint32_t *__restrict__ packed) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= ngroups) return;
    const sec_u32x4 w = *reinterpret_cast<const sec_u32x4 *>(words + 4u * (size_t)g);
    packed[3u * (size_t)g] = w.x | (w.y << 24);
    packed[3u * (size_t)g + 1u] = (w.y >> 8) | (w.z << 16);
    packed[3u * (size_t)g + 2u] = (w.z >> 16) | (w.w << 8);
}
// which entry of an off-diagonal pair keeps its matrix element: a hash bit of the pair (lower index, x mask) — per pair, not
// per entry, so that every row keeps about half of its elements
__device__ __forceinline__ bool sec_low_owns(uint64_t low, uint64_t x) {
    return (((low * 0x9e3779b97f4a7c15ull) ^ (x * 0xc2b2ae3d27d4eb4full)) >> 40) & 1ull;
}

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

// the coset of a program's Z2 symmetries through |hf>, ascending: member k has the kept bits pdep(k, kept) and every free bit f set to
// s_f ^ parity(member & G_f) (G_f: kept bits above f; sv_regular_host.hpp z2_symmetries)
struct SecCoset {
    uint32_t kept;
    int32_t nfree;
    uint32_t fbit[8], G[8], s[8];
};
__global__ __launch_bounds__(256) void k_sec_coset(SecCoset c, uint32_t K, uint32_t *__restrict__ sup) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= K) return;
    uint32_t i = 0, m = c.kept, q = k;
    while (m) {   // pdep (the mask is uniform)
        const uint32_t low = m & (0u - m);
        if (q & 1u) i |= low;
        q >>= 1;
        m &= m - 1u;
    }
    for (int f = c.nfree - 1; f >= 0; --f)   // (G_f holds kept bits only: the order does not matter)
        if ((c.s[f] ^ (uint32_t)__popc(i & c.G[f])) & 1u) i |= 1u << c.fbit[f];
    sup[k] = i;
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
// src in tile-padded form: srcpad[tile * cap + k] = src[off[tile] + k], 0xffffffff beyond the tile's entries — a sweep can
// then read its gather indices before it knows its tile's bounds
__global__ __launch_bounds__(256) void k_sec_pad_src(const uint32_t *__restrict__ src, const uint32_t *__restrict__ off, uint32_t cap,
                                                     uint32_t *__restrict__ srcpad) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    for (uint32_t k = threadIdx.x; k < cap; k += 256u) srcpad[(size_t)t * cap + k] = k < n ? src[e0 + k] : 0xffffffffu;
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
template <bool FILL, int NT, bool DENSE>
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
    uint16_t *slot_of = reinterpret_cast<uint16_t *>(sec_lk);   // DENSE: local key -> slot (see k_sec_hbuild)
    if (DENSE) {
        for (uint32_t k = threadIdx.x; k < (1u << M) / 2u; k += NT) sec_lk[k] = 0xffffffffu;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += NT) slot_of[keys[e0 + k] & lmask] = (uint16_t)k;
    } else {
        for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    }
    __syncthreads();
    auto find = [&](uint32_t key) -> int {
        if (DENSE) {
            const uint32_t v = slot_of[key];
            return v == 0xffffu ? -1 : (int)v;
        }
        return sec_find(sec_lk, n, key);
    };
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
                const uint32_t lk = keys[e0 + k] & lmask;
                for (int p = 0; p < op.npat; ++p) {
                    const SecPat pt = pats[op.pat0 + p];
                    const uint64_t bits = i & pt.pm;
                    if (bits == pt.pv) {
                        const int sj = find(lk ^ xl);
                        const uint32_t sign = (uint32_t)(parity64(i & op.zs) ^ op.flip);
                        word = (uint32_t)k | ((sj < 0 ? orphan : (uint32_t)sj) << sb) | (sign << (2 * sb)) | ((uint32_t)p << (2 * sb + 1));
                        emit = true;
                        break;
                    }
                    if (bits == (pt.pv ^ (op.x & pt.pm))) {   // second member: only its orphans are recorded
                        if (find(lk ^ xl) < 0) {
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

// Second form of the builder (round 4): no workgroup barrier inside the loop over the ops.  Every WAVE owns a contiguous range of the
// tile's entries (a multiple of 64), so the pairs of an op come out in ascending slot order — the layout of the first form, bit for
// bit — from a ballot inside the wave and, between the waves, from per-(op, wave) counts: the count pass leaves their exclusive
// prefix in wbase[(tile * nops + op) * NW + wave], the fill pass starts there.  The ops, their patterns and the x masks in tile-local
// bits are staged in LDS (as scalar loads from memory every op record cost a full memory latency, twice per 1024-entry chunk a
// barrier of 16 waves: N2 UCCSD 12.1 + 10.4 ms for the two passes of the 46 sweeps).
template <bool FILL, int NT, bool DENSE>
__global__ __launch_bounds__(NT) void k_sec_pairs2(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                   uint32_t smask, const SecBuildOp *__restrict__ ops, int nops,
                                                   const SecPat *__restrict__ pats, int npats, uint32_t map_bytes,
                                                   uint32_t *__restrict__ cnt, uint32_t *__restrict__ wbase,
                                                   const uint32_t *__restrict__ poff, uint32_t *__restrict__ pairs, int sb) {
    constexpr int NW = NT / 64;
    const uint32_t orphan = (1u << sb) - 1u;
    extern __shared__ uint32_t sec_lk[];
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) {
        if (!FILL)
            for (int o = threadIdx.x; o < nops; o += NT) cnt[(size_t)t * nops + o] = 0u;
        return;
    }
    unsigned char *p = reinterpret_cast<unsigned char *>(sec_lk) + map_bytes;
    SecBuildOp *lops = reinterpret_cast<SecBuildOp *>(p);
    p += (size_t)nops * sizeof(SecBuildOp);
    SecPat *lpats = reinterpret_cast<SecPat *>(p);
    p += (size_t)npats * sizeof(SecPat);
    uint32_t *lxl = reinterpret_cast<uint32_t *>(p);
    uint32_t *wcnt = lxl + nops;   // [nops][NW], count pass only
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(ops);
        uint32_t *dst = reinterpret_cast<uint32_t *>(lops);
        for (int k = threadIdx.x; k < nops * (int)(sizeof(SecBuildOp) / 4); k += NT) dst[k] = src[k];
        src = reinterpret_cast<const uint32_t *>(pats);
        dst = reinterpret_cast<uint32_t *>(lpats);
        for (int k = threadIdx.x; k < npats * (int)(sizeof(SecPat) / 4); k += NT) dst[k] = src[k];
        for (int o = threadIdx.x; o < nops; o += NT) lxl[o] = sec_pext((uint32_t)ops[o].x, smask);
    }
    const uint32_t lmask = (1u << M) - 1u;
    uint16_t *slot_of = reinterpret_cast<uint16_t *>(sec_lk);
    if (DENSE) {
        for (uint32_t k = threadIdx.x; k < (1u << M) / 2u; k += NT) sec_lk[k] = 0xffffffffu;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += NT) slot_of[keys[e0 + k] & lmask] = (uint16_t)k;
    } else {
        for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    }
    __syncthreads();
    auto find = [&](uint32_t key) -> int {
        if (DENSE) {
            const uint32_t v = slot_of[key];
            return v == 0xffffu ? -1 : (int)v;
        }
        return sec_find(sec_lk, n, key);
    };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = ((n + NW * 64 - 1) / (NW * 64)) * 64;   // entries per wave
    const int k_begin = wave * R, k_end = min(n, k_begin + R);
    const uint64_t below = (1ull << lane) - 1ull;
    for (int o = 0; o < nops; ++o) {
        const SecBuildOp op = lops[o];
        const uint32_t xl = lxl[o];
        uint32_t at = 0;
        if (FILL) at = poff[(size_t)t * (nops + 1) + o] + wbase[((size_t)t * nops + o) * NW + wave];
        uint32_t running = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += 64) {
            const int k = k0 + lane;
            bool emit = false;
            uint32_t word = 0;
            if (k < k_end) {
                const uint64_t i = sup[cid[e0 + k]];
                const uint32_t lk = keys[e0 + k] & lmask;
                for (int q = 0; q < op.npat; ++q) {
                    const SecPat pt = lpats[op.pat0 + q];
                    const uint64_t bits = i & pt.pm;
                    if (bits == pt.pv) {
                        const int sj = find(lk ^ xl);
                        const uint32_t sign = (uint32_t)(parity64(i & op.zs) ^ op.flip);
                        word = (uint32_t)k | ((sj < 0 ? orphan : (uint32_t)sj) << sb) | (sign << (2 * sb)) | ((uint32_t)q << (2 * sb + 1));
                        emit = true;
                        break;
                    }
                    if (bits == (pt.pv ^ (op.x & pt.pm))) {   // second member: only its orphans are recorded
                        if (find(lk ^ xl) < 0) {
                            word = (uint32_t)k | (orphan << sb) | ((uint32_t)q << (2 * sb + 1));
                            emit = true;
                        }
                        break;
                    }
                }
            }
            const uint64_t bal = __ballot(emit);
            if (FILL && emit) pairs[at + running + (uint32_t)__popcll(bal & below)] = word;
            running += (uint32_t)__popcll(bal);
        }
        if (!FILL && lane == 0) wcnt[o * NW + wave] = running;
    }
    if (!FILL) {
        __syncthreads();
        for (int o = threadIdx.x; o < nops; o += NT) {
            uint32_t acc = 0;
            for (int w = 0; w < NW; ++w) {
                wbase[((size_t)t * nops + o) * NW + w] = acc;
                acc += wcnt[o * NW + w];
            }
            cnt[(size_t)t * nops + o] = acc;
        }
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
// D2: two chunks of pair words ahead in registers instead of one (sweeps whose tiles are dense: the reference's QUCCSD templates at
// 24 qubits stream 3.5 G pair words per evaluation, two ops per chunk — a chunk's rotations take less time than a trip to HBM,
// and with one chunk ahead the sweep ran at the bytes-in-flight limit, 2.4 TB/s); requires that no op of a tile exceeds a
// buffer (host: SectorSeg::max_op_pairs).
template <int NT, bool D2 = false>
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
    // Order of the prologue: everything that depends only on the tile number first (op table, cos/sin, tile bounds), then
    // — one round trip later — the first chunk of pair words AND the gather indices together, then the gather itself; the
    // pair words used to wait for the gather (one more dependent trip to memory per sweep).
    const uint32_t t = blockIdx.x;
    const uint32_t *po = poff + (size_t)t * (nops + 1);
    for (int o = threadIdx.x; o <= nops; o += NT) lop[o] = SecOpLds{po[o], o < nops ? tab0[o] - rot0 : 0};
    for (int r = threadIdx.x; r < nrot; r += NT) {
        const RotParam rr = rp[rot0 + r];
        cs[r] = make_double2(rr.c, rr.s);
    }
    constexpr int PF = 4;   // gather indices per thread read before the tile's bounds are known (tile-padded array)
    uint32_t gidx[PF];
    const uint32_t *sp = src ? src + (size_t)t * tile_cap : nullptr;
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const uint32_t k = threadIdx.x + (uint32_t)r * NT;
        gidx[r] = (sp && k < tile_cap) ? sp[k] : 0xffffffffu;
    }
    const uint32_t e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) return;
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
    if (ob > oa) fetch(oa, ob);
    if (src) {
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int k = (int)threadIdx.x + r * NT;
            if (k < n) tile[k] = in[gidx[r]];
        }
        for (int k = (int)threadIdx.x + PF * NT; k < n; k += NT) tile[k] = in[sp[k]];
    } else {
        for (int k = threadIdx.x; k < n; k += NT) tile[k] = (e0 + (uint32_t)k == hf_pos) ? 1.0 : 0.0;
    }
    if (ob > oa) stash(wbuf);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if constexpr (D2) {
        uint32_t r0[SEC_WORDS_PER_THREAD], r1[SEC_WORDS_PER_THREAD];
        auto fetch2 = [&](uint32_t (&rr)[SEC_WORDS_PER_THREAD], int a, int b) {
            const uint32_t base = lop[a].p0, cnt = lop[b].p0 - base;
#pragma unroll
            for (int r = 0; r < SEC_WORDS_PER_THREAD; ++r) {
                const uint32_t idx = threadIdx.x + (uint32_t)r * NT;
                rr[r] = idx < cnt ? pairs[base + idx] : 0u;
            }
        };
        auto stash2 = [&](const uint32_t (&rr)[SEC_WORDS_PER_THREAD], uint32_t *buf) {
#pragma unroll
            for (int r = 0; r < SEC_WORDS_PER_THREAD; ++r) buf[threadIdx.x + (uint32_t)r * NT] = rr[r];
        };
        // chunk c = ops [q0, q1), c + 1 = [q1, q2), c + 2 = [q2, q3): r1 takes chunk 1, r0 chunk 2
        int q0 = oa, q1 = ob, q2 = q1 < nops ? chunk_end(q1) : q1, q3 = q2 < nops ? chunk_end(q2) : q2;
        if (q2 > q1) fetch2(r1, q1, q2);
        if (q3 > q2) fetch2(r0, q2, q3);
        auto iter = [&](uint32_t (&ra)[SEC_WORDS_PER_THREAD]) {   // ra holds chunk c + 1
            const uint32_t *wb = wbuf + (size_t)cb * W;
            const uint32_t base = lop[q0].p0;
            uint32_t p0 = 0;
            for (int o = q0; o < q1; ++o) {
                const uint32_t p1 = lop[o + 1].p0 - base;
                const double2 *tab = cs + lop[o].tab;
                for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) sec_rotate<NT>(tile, tab, wb[k], bad, sb);
                p0 = p1;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            int q4 = q3;
            if (q2 > q1) {
                stash2(ra, wbuf + (size_t)(cb ^ 1) * W);
                q4 = q3 < nops ? chunk_end(q3) : q3;
                if (q4 > q3) fetch2(ra, q3, q4);   // chunk c + 3, two chunks' rotations ahead of its use
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            cb ^= 1;
            q0 = q1;
            q1 = q2;
            q2 = q3;
            q3 = q4;
            return q0 < q1;
        };
        if (q0 < q1) {
            for (;;) {
                if (!iter(r1)) break;
                if (!iter(r0)) break;
            }
        }
        oa = nops;   // (the one-chunk-ahead loop below is skipped)
    }
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

// ---- one sweep of the circuit, second form (round 3) -------------------------------------------------------------------
// What the first form spends outside its rotations is a chain of dependent trips to memory and LDS: tile bounds / op table ->
// gather indices + first pair words -> gather -> [per op: op bounds -> staged pair word -> cos/sin + amplitudes] -> store
// (rocprofv3: 9 us of the 24 us per sweep without the ops, 0.4 us = 840 cycles per op).  Here
//  (a) the PREVIOUS sweep scatters its tile into this sweep's order (dst indices; tile-padded layout in[tile * cap + k]), so
//      tile, cos/sin table, first pair words and dst indices are addressed by the tile number alone and arrive in ONE trip,
//      loaded in unrolled batches (a load -> LDS-store loop is compiled to one memory round trip per iteration);
//  (b) pair words are 64 bits wide and carry everything a thread needs to act without looking anything up:
//        slot_i | slot_j << 16 | sign << 32 | cos/sin entry of the sweep << 33 (12 bits) | round << 45 (12 bits)
//      (slot_j = 0xffff: partner outside the support).  They never pass through LDS: chunk c of the tile's list is the words
//      [c CH, (c + 1) CH), thread t keeps words t, t + NT, ... of the current chunk in registers with their cos/sin values
//      (the next chunk's loads already in flight).  A ROUND is the part of one op inside one chunk — pairs of an op are
//      disjoint, so an op cut by a chunk boundary is applied in two rounds; rounds are numbered from 0 inside every chunk at
//      build time (k_sec_widen), ops without pairs in the tile have no round;
//  (c) per round that leaves: compare -> two LDS reads -> rotate -> two writes -> barrier.
// blockIdx.y = state of a batch (own in / out slices and angle tables).
constexpr uint32_t SEC_CHUNK = 4096;            // pair words per chunk (k_sector_sweep2: NT x WPT), option "sector_chunk"
constexpr uint32_t SEC_NO_ROUND = 0xfffu;
constexpr uint32_t SEC_WIDE_MAX_PAIRS = 16u << 20;   // pairs of a sweep up to which the 64-bit tables are built (128 MB of words)
__device__ __forceinline__ uint64_t sec_word64(uint32_t si, uint32_t sj, uint32_t sign, uint32_t csidx, uint32_t round) {
    return (uint64_t)si | ((uint64_t)sj << 16) | ((uint64_t)sign << 32) | ((uint64_t)csidx << 33) | ((uint64_t)round << 45);
}
// 32-bit pair words of one sweep -> 64-bit words + rounds per (tile, chunk); one workgroup per tile
__global__ __launch_bounds__(256) void k_sec_widen(const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ poff,
                                                   const int32_t *__restrict__ tab0, int nops, int rot0, int sb, uint32_t maxchunks,
                                                   uint32_t chunk, uint64_t *__restrict__ wide, uint16_t *__restrict__ rounds) {
    const uint32_t t = blockIdx.x;
    const uint32_t *po = poff + (size_t)t * (nops + 1);
    const uint32_t pbase = po[0];
    const uint32_t mask = (1u << sb) - 1u;
    uint32_t cur_chunk = 0, cur_round = 0;
    for (int o = 0; o < nops; ++o) {
        uint32_t a = po[o] - pbase;
        const uint32_t b = po[o + 1] - pbase;
        const uint32_t tab = (uint32_t)(tab0[o] - rot0);
        while (a < b) {   // the op's part inside chunk a / CH
            const uint32_t c = a / chunk, e = min(b, (c + 1u) * chunk);
            if (c != cur_chunk) {
                if (threadIdx.x == 0) rounds[(size_t)t * maxchunks + cur_chunk] = (uint16_t)cur_round;
                cur_chunk = c;
                cur_round = 0;
            }
            for (uint32_t k = a + threadIdx.x; k < e; k += 256u) {
                const uint32_t pw = pairs[(size_t)pbase + k];
                const uint32_t si = pw & mask, sj = (pw >> sb) & mask;
                wide[(size_t)pbase + k] = sec_word64(si, sj == mask ? 0xffffu : sj, (pw >> (2 * sb)) & 1u, tab + (pw >> (2 * sb + 1)), cur_round);
            }
            ++cur_round;
            a = e;
        }
    }
    if (threadIdx.x == 0) {
        rounds[(size_t)t * maxchunks + cur_chunk] = (uint16_t)cur_round;
        for (uint32_t c = cur_chunk + 1u; c < maxchunks; ++c) rounds[(size_t)t * maxchunks + c] = 0;
    }
}
template <int NT, int WPT>
__global__ __launch_bounds__(NT) void k_sector_sweep2(const double *__restrict__ in, double *__restrict__ out, size_t in_stride,
                                                      size_t out_stride, const uint32_t *__restrict__ dstpad,
                                                      const uint32_t *__restrict__ off, const uint32_t *__restrict__ poff, int nops,
                                                      const uint64_t *__restrict__ wide, const uint16_t *__restrict__ rounds,
                                                      uint32_t maxchunks, const RotParam *__restrict__ rp, size_t rp_stride, int rot0,
                                                      int nrot, uint32_t tile_cap, uint32_t hf_pos, int *__restrict__ flag, int dbg, int dst_lds, int bfast,
                                                      const uint32_t *__restrict__ torder) {
    constexpr uint32_t CH = (uint32_t)NT * WPT;   // = the chunk size the tables were built for (host checks)
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    const uint32_t spare = tile_cap;     // one slot behind the tile (see the orphan words below); tile_cap < 0xffff
    double2 *cs = reinterpret_cast<double2 *>(tile + ((tile_cap + 2u) & ~1u));
    uint32_t *dst = reinterpret_cast<uint32_t *>(cs + nrot);
    uint32_t *nround = dst + (dst_lds ? tile_cap : 0u);   // [maxchunks]; dst_lds = 0 (batches: three workgroups per CU instead of two): the
                                                            // scatter indices are read from memory when the tile is written
    // bfast: the grid of a BATCH has the state as its fastest index: the workgroups that
    // apply the same tile's pair words to different states run side by side and share the words through the caches
    const uint32_t tq = bfast ? blockIdx.y : blockIdx.x, b = bfast ? blockIdx.x : blockIdx.y;
    const uint32_t t = torder ? torder[tq] : tq;   // (many tiles per CU: largest first)
    if (dbg == 4) return;
    const uint32_t e0 = off[t];
    const uint32_t n = off[t + 1] - e0;
    if (n == 0) return;
    const uint32_t pbase = poff[(size_t)t * (nops + 1)], ptot = poff[(size_t)t * (nops + 1) + nops] - pbase;
    // dbg: measurements only (1: no ops, 2: scalar loads only, 3: loads only, 4: empty, 5: rounds = barriers only, 6: rounds without barriers)
    const uint32_t nchunks = (dbg >= 1 && dbg <= 4) ? 0u : (ptot + CH - 1u) / CH;
    if (dbg == 2) return;
    in += (size_t)b * in_stride;
    out += (size_t)b * out_stride;
    rp += (size_t)b * rp_stride;
    const size_t tbase = (size_t)t * tile_cap;
    const uint64_t *wp = wide + pbase;
    uint64_t wa[WPT], wb[WPT];
    // ---- one trip to memory: words of chunk 0, tile, dst indices, cos/sin (clamped addresses, no branches around the loads;
    // words past the tile's list are recognised by their index when the chunk is decoded — nothing here consumes a loaded value)
#pragma unroll
    for (int r = 0; r < WPT; ++r) wa[r] = wp[min(threadIdx.x + (uint32_t)r * NT, ptot ? ptot - 1u : 0u)];
    constexpr int TB = 8;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += TB * NT) {
        uint32_t d[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) d[j] = dst_lds ? dstpad[tbase + min(k0 + (uint32_t)j * NT, n - 1u)] : 0u;
        if (in) {
            double v[TB];
#pragma unroll
            for (int j = 0; j < TB; ++j) v[j] = in[tbase + min(k0 + (uint32_t)j * NT, n - 1u)];
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) tile[k0 + (uint32_t)j * NT] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) tile[k0 + (uint32_t)j * NT] = (e0 + k0 + (uint32_t)j * NT == hf_pos) ? 1.0 : 0.0;
        }
        if (dst_lds) {
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) dst[k0 + (uint32_t)j * NT] = d[j];
        }
    }
    for (int r0 = threadIdx.x; r0 < nrot; r0 += 2 * NT) {
        const RotParam ra = rp[rot0 + r0], rb = rp[rot0 + min(r0 + NT, nrot - 1)];
        cs[r0] = make_double2(ra.c, ra.s);
        if (r0 + NT < nrot) cs[r0 + NT] = make_double2(rb.c, rb.s);
    }
    if (threadIdx.x < maxchunks) nround[threadIdx.x] = rounds[(size_t)t * maxchunks + threadIdx.x];
    if (threadIdx.x == 0) tile[spare] = 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (dbg == 3) return;
    bool bad = false;
    const uint64_t tm0 = dbg == 7 ? __builtin_amdgcn_s_memtime() : 0, tr0 = dbg == 7 ? __builtin_amdgcn_s_memrealtime() : 0;
    uint32_t rtot = 0;
    for (uint32_t c = 0; c < nchunks; ++c) {
        // the next chunk's words: in flight while this chunk is applied
        const uint32_t nbeg = (c + 1u) * CH;
#pragma unroll
        for (int r = 0; r < WPT; ++r) wb[r] = wp[min(nbeg + threadIdx.x + (uint32_t)r * NT, ptot - 1u)];
        // the thread's words of this chunk, decoded once: slots, round, cos, signed sin.  A word whose partner is outside the
        // support rotates its amplitude against the spare slot behind the tile by the identity (branch-free) and reports a
        // non-zero amplitude.  (A queue with one compare per round and a register shift per rotation was measured: no faster.)
        uint32_t qs[WPT], qr[WPT];
        double qc[WPT], qn[WPT];
#pragma unroll
        for (int r = 0; r < WPT; ++r) {
            const double2 cr = cs[(uint32_t)(wa[r] >> 33) & 0xfffu];
            const bool live = c * CH + threadIdx.x + (uint32_t)r * NT < ptot;
            const bool orphan = ((uint32_t)(wa[r] >> 16) & 0xffffu) == 0xffffu;
            qs[r] = orphan ? (((uint32_t)wa[r] & 0xffffu) | (spare << 16)) : (uint32_t)wa[r];
            qr[r] = live ? (((uint32_t)(wa[r] >> 45) & 0xfffu) | (orphan ? 0x1000u : 0u)) : SEC_NO_ROUND;
            qc[r] = orphan ? 1.0 : cr.x;
            qn[r] = orphan ? 0.0 : (((uint32_t)(wa[r] >> 32) & 1u) ? -cr.y : cr.y);
        }
        const uint32_t R = nround[c];
        rtot += R;
        for (uint32_t q = 0; q < R; ++q) {
#pragma unroll
            for (int r = 0; r < WPT; ++r) {
                if ((qr[r] & 0xfffu) == q && dbg != 5) {
                    const uint32_t si = qs[r] & 0xffffu, sj = qs[r] >> 16;
                    const double u = tile[si], v = tile[sj];
                    tile[si] = qc[r] * u + qn[r] * v;
                    tile[sj] = qc[r] * v - qn[r] * u;
                    bad |= (qr[r] & 0x1000u) && u != 0.0;
                }
            }
            if (dbg != 6) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < WPT; ++r) wa[r] = wb[r];
    }
    if (dbg == 7 && threadIdx.x == 0 && (t % 37u) == 5u) {   // measurement: shader clock and time inside the rounds
        const uint64_t tm1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
        printf("tile %u n %u pairs %u chunks %u rounds %u: %.2f us in the chunk loop, clock %.0f MHz\n", t, n, ptot, nchunks, rtot,
               (double)(tr1 - tr0) / 100.0, (double)(tm1 - tm0) / (double)(tr1 - tr0) * 100.0);
    }
    if (dst_lds) {
        for (uint32_t k = threadIdx.x; k < n; k += NT) out[dst[k]] = tile[k];
    } else {
        for (uint32_t k = threadIdx.x; k < n; k += NT) out[dstpad[tbase + k]] = tile[k];
    }
    if (bad) atomicOr(flag, 1);
}
// ---- one sweep of the circuit, third form (round 5): pair words in PER-WAVE streams --------------------------------------
// What bounds the second form is the workgroup barrier behind every round (24-qubit UCCSD: 41 rounds of 0.3 us per sweep; rocprofv3:
// waves wait 61 % of their cycles, VALU 6.5 %, LDS 1.2 %).  Consecutive ops of a sweep mix few bits between them (UCCSD: neighbours in
// the list differ in one orbital), so the sweep's op list is cut into RUNS whose mixing masks together leave at least log2(waves) of the
// tile's index bits untouched: those bits split the tile's slots into classes no op of the run leaves, every class belongs to one
// wave for the length of the run (classes dealt to waves by pair count, largest first: k_sec_wave_plan), and a wave applies its part of
// every op of the run in op order with nothing but the in-order LDS pipe between them.  Workgroup barriers are left at the run
// boundaries only (24-qubit UCCSD: 140 for 1715 ops).
//   stream of (tile, wave): ROWS of 64 32-bit pair words (slot_i | slot_j << sb | pattern << 2 sb | sign << 31 — the first form's
//   fields with the sign moved to where it is xor-ed into the sine; 0xffffffff: empty lane), one op per row (an op with more than 64
//   pairs in the wave's classes takes several), rows in op order; rowhdr[row] = first cos/sin entry of the row's op (bit 15: the op
//   has a word in this wave whose partner is outside the support, bit 14: words of more than one pattern); rowinfo[(tile, wave)][r] = first row of run r (r = nruns: the end).
//   Rows are loaded SEC_STREAM_G at a time, the next batch in flight while this one is applied.
constexpr int SEC_STREAM_WAVES = 16;     // waves per workgroup the streams are planned for (k_sector_sweep3<1024>)
constexpr int SEC_STREAM_G = 8;          // rows per batch
constexpr int SEC_STREAM_CLASS_BITS = 6; // at most 64 slot classes per run
// (SecWaveRun and the host side of the plan: sv_regular_host.hpp)
// plan (FILL = false: rows per (tile, wave, op) and the class -> wave map) and fill (FILL = true: words and row headers) of the streams;
// one workgroup per tile
template <bool FILL>
__global__ __launch_bounds__(256) void k_sec_wave_plan(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ cid,
                                                       const uint32_t *__restrict__ off, const uint32_t *__restrict__ pairs,
                                                       const uint32_t *__restrict__ poff, int nops, const int32_t *__restrict__ tab0, int rot0,
                                                       const SecWaveRun *__restrict__ runs, int nruns, int sb, int nw, uint32_t cap,
                                                       uint32_t *__restrict__ oprows, const uint32_t *__restrict__ rowoff,
                                                       uint8_t *__restrict__ wof, uint32_t *__restrict__ stream, uint16_t *__restrict__ rowhdr) {
    constexpr int NC = 1 << SEC_STREAM_CLASS_BITS;
    static_assert(NC == 64, "the classes are dealt to the waves by the 64 lanes of one wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    const int NW = nw;           // waves that work on the rows (a power of two up to SEC_STREAM_WAVES)
    uint32_t *bas = reinterpret_cast<uint32_t *>(sec_smem);   // [cap] basis index of every slot
    uint32_t *lpo = bas + cap;                                 // [nops + 1] the tile's pair offsets
    uint32_t *wcnt = lpo + nops + 1;                           // [nops][NW] pair words of (op, wave)
    uint32_t *wflag = wcnt + (size_t)nops * NW;                // [nops][NW] bit 31: a word without partner; bits 0..30: patterns seen (30: that or beyond)
    uint8_t *own = reinterpret_cast<uint8_t *>(wflag + (size_t)nops * NW);   // [cap] class, then wave, of every slot
    __shared__ uint32_t ccount[NC];
    __shared__ uint8_t wofc[NC];
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    const uint32_t mask = (1u << sb) - 1u;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += 4u * 256u) {   // (one trip per four slots of a thread)
        uint32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = cid[e0 + min(k0 + (uint32_t)j * 256u, n - 1u)];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = sup[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + (uint32_t)j * 256u < n) bas[k0 + (uint32_t)j * 256u] = c[j];
    }
    for (int o = threadIdx.x; o <= nops; o += 256) lpo[o] = poff[(size_t)t * (nops + 1) + o];
    for (int i = threadIdx.x; i < nops * NW; i += 256) {
        wcnt[i] = 0;
        wflag[i] = 0;
    }
    __syncthreads();
    int o0 = 0;
    for (int r = 0; r < nruns; ++r) {
        const int o1 = runs[r].op_end;
        const uint32_t cmask = runs[r].cmask;
        if (threadIdx.x < NC) ccount[threadIdx.x] = 0;
        for (uint32_t k = threadIdx.x; k < n; k += 256u) own[k] = (uint8_t)sec_pext(bas[k], cmask);
        __syncthreads();
        uint8_t *wo = wof + ((size_t)t * nruns + r) * NC;
        if (!FILL) {
            for (uint32_t k = lpo[o0] + threadIdx.x; k < lpo[o1]; k += 256u) atomicAdd(&ccount[own[pairs[k] & mask]], 1u);
            __syncthreads();
            if (threadIdx.x < 64u) {   // classes to waves: largest first onto the wave with the fewest pairs so far (ties: lowest class, lowest wave)
                const uint32_t lane = threadIdx.x;
                unsigned long long key = (((unsigned long long)ccount[lane] + 1ull) << 6) | (63u - lane);   // 0 once the class has its wave
                uint32_t load = 0, myw = 0;
                for (int it = 0; it < NC; ++it) {
                    unsigned long long kmax = key;
                    for (int d = 32; d; d >>= 1) {
                        const unsigned long long other = __shfl_xor(kmax, d);
                        kmax = other > kmax ? other : kmax;
                    }
                    const uint32_t best = 63u - (uint32_t)(kmax & 63ull), bc = (uint32_t)(kmax >> 6) - 1u;
                    unsigned long long kmin = (int)lane < NW ? (((unsigned long long)load << 6) | lane) : ~0ull;
                    for (int d = 32; d; d >>= 1) {
                        const unsigned long long other = __shfl_xor(kmin, d);
                        kmin = other < kmin ? other : kmin;
                    }
                    const uint32_t w0 = (uint32_t)(kmin & 63ull);
                    if (lane == w0) load += bc;
                    if (lane == best) {
                        myw = w0;
                        key = 0ull;
                    }
                }
                wofc[lane] = (uint8_t)myw;
                wo[lane] = (uint8_t)myw;
            }
        } else if (threadIdx.x < NC) {
            wofc[threadIdx.x] = wo[threadIdx.x];
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += 256u) own[k] = wofc[own[k]];
        __syncthreads();
        // NOTE on reproducibility: a pair word's rank inside its (op, wave) row set comes from an LDS atomic, i.e. from thread timing:
        // WHICH lane (and, beyond 64 pairs, which row) of the op's rows a pair occupies can differ between two builds of the same
        // tables.  The forward sweep (k_sector_sweep3) does not care — every pair is rotated exactly once, energies are bit-identical.
        // The backward sweep (k_sector_adjoint3) sums its gradient terms per lane and per row, so the gradients of two builds of one
        // handle agree to rounding (1e-16 relative), not bit for bit; within one build every call repeats the same order.
        for (int o = o0; o < o1; ++o) {   // (every (op, wave) has its own counter: nothing to wait for between the ops)
            for (uint32_t k = lpo[o] + threadIdx.x; k < lpo[o + 1]; k += 256u) {
                const uint32_t pw = pairs[k], w = own[pw & mask];
                const uint32_t rank = atomicAdd(&wcnt[o * NW + w], 1u);
                if (FILL) {
                    stream[(size_t)rowoff[((size_t)t * NW + w) * (nops + 1) + o] * 64u + rank] =
                        (pw & ((1u << (2 * sb)) - 1u)) | ((pw >> (2 * sb + 1)) << (2 * sb)) | (((pw >> (2 * sb)) & 1u) << 31);
                    atomicOr(&wflag[o * NW + w], ((((pw >> sb) & mask) == mask) ? 0x80000000u : 0u) | (1u << min(pw >> (2 * sb + 1), 30u)));
                }
            }
        }
        __syncthreads();
        o0 = o1;
    }
    for (int i = threadIdx.x; i < nops * NW; i += 256) {
        const int o = i / NW, w = i - o * NW;
        const size_t q = (size_t)t * NW + w;
        if (!FILL) {
            oprows[q * nops + o] = (wcnt[i] + 63u) >> 6;
        } else {
            const uint32_t fl = wflag[i];
            const uint16_t hd = (uint16_t)((uint32_t)(tab0[o] - rot0) | ((fl >> 31) ? 0x8000u : 0u) |
                                           ((__popc(fl & 0x7fffffffu) > 1 || (fl & 0x40000000u)) ? 0x4000u : 0u));
            for (uint32_t rw = rowoff[q * (nops + 1) + o]; rw < rowoff[q * (nops + 1) + o + 1]; ++rw) rowhdr[rw] = hd;
        }
    }
}
// Lanes of a row chosen for the LDS banks (MI355X: a 64-bit read is served in two groups of 32 lanes over 64 dword banks, a 64-bit write in
// four groups of 16 lanes over 32: slots that agree mod 32 / mod 16 inside a group cost a cycle each; the second form's counters: 83 %
// of its LDS cycles are such conflicts).  One thread per row: every word goes to the 16-lane group where its two slots collide with
// the fewest already placed (first fit), groups filled from their first lane.
__global__ __launch_bounds__(256) void k_sec_row_arrange(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t rows, int sb) {
    const uint64_t row = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (row >= rows) return;
    const uint32_t mask = (1u << sb) - 1u;
    const uint32_t *src = in + row * 64u;
    uint32_t *dst = out + row * 64u;
    uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
    uint32_t wi0 = 0, wi1 = 0, wi2 = 0, wi3 = 0, wj0 = 0, wj1 = 0, wj2 = 0, wj3 = 0;   // residues mod 16 taken in a group: first / second slot
    uint32_t ri0 = 0, ri1 = 0, rj0 = 0, rj1 = 0;                                       // residues mod 32 taken in a half
    for (int k = 0; k < 64; ++k) {
        const uint32_t w = src[k];
        if (w == 0xffffffffu) break;       // (the fill leaves the words of a row contiguous from lane 0)
        const uint32_t si = w & mask, sj = (w >> sb) & mask;
        const bool plain = sj != mask;
        const uint32_t a16 = 1u << (si & 15u), a32 = 1u << (si & 31u), b16 = plain ? 1u << (sj & 15u) : 0u, b32 = plain ? 1u << (sj & 31u) : 0u;
        const uint32_t c0 = cnt0 < 16u ? ((wi0 & a16) != 0) + ((wj0 & b16) != 0) + ((ri0 & a32) != 0) + ((rj0 & b32) != 0) : 99u;
        const uint32_t c1 = cnt1 < 16u ? ((wi1 & a16) != 0) + ((wj1 & b16) != 0) + ((ri0 & a32) != 0) + ((rj0 & b32) != 0) : 99u;
        const uint32_t c2 = cnt2 < 16u ? ((wi2 & a16) != 0) + ((wj2 & b16) != 0) + ((ri1 & a32) != 0) + ((rj1 & b32) != 0) : 99u;
        const uint32_t c3 = cnt3 < 16u ? ((wi3 & a16) != 0) + ((wj3 & b16) != 0) + ((ri1 & a32) != 0) + ((rj1 & b32) != 0) : 99u;
        uint32_t best = 0, bc = c0;
        if (c1 < bc) { best = 1; bc = c1; }
        if (c2 < bc) { best = 2; bc = c2; }
        if (c3 < bc) { best = 3; bc = c3; }
        uint32_t pos;
        if (best == 0) { pos = cnt0++; wi0 |= a16; wj0 |= b16; ri0 |= a32; rj0 |= b32; }
        else if (best == 1) { pos = 16u + cnt1++; wi1 |= a16; wj1 |= b16; ri0 |= a32; rj0 |= b32; }
        else if (best == 2) { pos = 32u + cnt2++; wi2 |= a16; wj2 |= b16; ri1 |= a32; rj1 |= b32; }
        else { pos = 48u + cnt3++; wi3 |= a16; wj3 |= b16; ri1 |= a32; rj1 |= b32; }
        dst[pos] = w;
    }
}
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_sweep3(const double *__restrict__ in, double *__restrict__ out, size_t in_stride,
                                                      size_t out_stride, const uint32_t *__restrict__ dstpad,
                                                      const uint32_t *__restrict__ off, const uint32_t *__restrict__ rowinfo, int nruns, int nwave,
                                                      const uint32_t *__restrict__ stream, const uint16_t *__restrict__ rowhdr,
                                                      const RotParam *__restrict__ rp, size_t rp_stride, int rot0, int nrot, uint32_t tile_cap,
                                                      uint32_t hf_pos, int *__restrict__ flag, int sb, int dst_lds, int bfast,
                                                      const uint32_t *__restrict__ torder, int dbg) {
    constexpr int G = SEC_STREAM_G;
    static_assert(NT >= 64 * SEC_STREAM_WAVES, "the streams are planned for up to this many waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    const uint32_t spare = tile_cap;     // one slot behind the tile: partner of the words whose partner is outside the support
    double2 *cs = reinterpret_cast<double2 *>(tile + ((tile_cap + 2u) & ~1u));
    uint32_t *dst = reinterpret_cast<uint32_t *>(cs + nrot);
    const uint32_t tq = bfast ? blockIdx.y : blockIdx.x, b = bfast ? blockIdx.x : blockIdx.y;
    const uint32_t t = torder ? torder[tq] : tq;
    const uint32_t e0 = off[t];
    const uint32_t n = off[t + 1] - e0;
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the wave's run boundaries (lane r: first row of run r), then its first rows — ahead of the tile's loads
    // (waves past the nwave the streams were planned for have no rows: they load, store and meet the others at the barriers)
    const uint32_t myb = wv < (uint32_t)nwave ? rowinfo[((size_t)t * (uint32_t)nwave + wv) * (uint32_t)(nruns + 1) + min(lane, (uint32_t)nruns)] : 0u;
    const uint32_t R0 = __builtin_amdgcn_readlane(myb, 0), R1 = __builtin_amdgcn_readlane(myb, nruns);
    uint32_t cw[G], nw[G], ch, nh;
#pragma unroll
    for (int g = 0; g < G; ++g) cw[g] = stream[(size_t)(R0 + g) * 64u + lane];
    ch = rowhdr[R0 + (lane & (G - 1))];
    in += (size_t)b * in_stride;
    out += (size_t)b * out_stride;
    rp += (size_t)b * rp_stride;
    const size_t tbase = (size_t)t * tile_cap;
    constexpr int TB = 8;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += TB * NT) {
        uint32_t d[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) d[j] = dst_lds ? dstpad[tbase + min(k0 + (uint32_t)j * NT, n - 1u)] : 0u;
        if (in) {
            double v[TB];
#pragma unroll
            for (int j = 0; j < TB; ++j) v[j] = in[tbase + min(k0 + (uint32_t)j * NT, n - 1u)];
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) tile[k0 + (uint32_t)j * NT] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) tile[k0 + (uint32_t)j * NT] = (e0 + k0 + (uint32_t)j * NT == hf_pos) ? 1.0 : 0.0;
        }
        if (dst_lds) {
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) dst[k0 + (uint32_t)j * NT] = d[j];
        }
    }
    for (int r0 = threadIdx.x; r0 < nrot; r0 += 2 * NT) {
        const RotParam ra = rp[rot0 + r0], rb = rp[rot0 + min(r0 + NT, nrot - 1)];
        cs[r0] = make_double2(ra.c, ra.s);
        if (r0 + NT < nrot) cs[r0 + NT] = make_double2(rb.c, rb.s);
    }
    if (threadIdx.x == 0) tile[spare] = 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    bool bad = false;
    const uint32_t mask = (1u << sb) - 1u;
    int rb = 1;      // next run boundary of this wave
    uint32_t bnd = __builtin_amdgcn_readlane(myb, min(1, nruns));
    if (dbg == 1) rb = nruns;
    unsigned char *const lds = sec_smem;
    for (uint32_t base = R0; base < R1 && dbg != 1; base += G) {
#pragma unroll
        for (int g = 0; g < G; ++g) nw[g] = stream[(size_t)(base + G + g) * 64u + lane];   // (the tables end with 2 G rows of padding)
        nh = rowhdr[base + G + (lane & (G - 1))];
        // The batch decoded ahead of its rows: byte addresses of both slots, cos and signed sin — ten vector instructions a word (the
        // first version decoded inside the row: 45 instructions per row, and a wave's rows are one dependent chain).  What is left
        // per row is the chain itself: two LDS reads, four multiply-adds, two writes.  Empty lanes sit the row out (EXEC-masked lanes
        // take no part in the LDS banking).  A word whose partner is outside the support (its row's header says so: the batch
        // decodes with selects) rotates its amplitude against the spare slot by the identity, and the rows check that it is zero.
        uint32_t ai[G], aj[G];
        double qc[G], qn[G];
        const bool slow = __ballot((ch & 0x8000u) != 0u) != 0ull;
        const uint32_t pshift = 2u * (uint32_t)sb;
        if (!slow) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t w = base + g < R1 ? cw[g] : 0xffffffffu;
                const uint32_t csb = __builtin_amdgcn_readlane(ch, g) & 0x3fffu;
                const double2 cr = cs[csb + ((w & 0x7fffffffu) >> pshift)];   // (an empty lane reads past the table: nothing it gets is used)
                ai[g] = (w & mask) << 3;
                aj[g] = ((w >> sb) & mask) << 3;
                qc[g] = cr.x;
                qn[g] = __hiloint2double(__double2hiint(cr.y) ^ (int)(w & 0x80000000u), __double2loint(cr.y));
                asm volatile("" : "+v"(ai[g]), "+v"(aj[g]), "+v"(qc[g]), "+v"(qn[g]));
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t w = base + g < R1 ? cw[g] : 0xffffffffu;
                const uint32_t csb = __builtin_amdgcn_readlane(ch, g) & 0x3fffu;
                const bool empty = w == 0xffffffffu;
                const uint32_t sj = (w >> sb) & mask;
                const bool plain = sj != mask;      // (an empty lane reads as a word without partner)
                const double2 cr = cs[empty ? 0u : csb + ((w & 0x7fffffffu) >> pshift)];
                ai[g] = (w & mask) << 3;
                aj[g] = (plain ? sj : spare) << 3;
                qc[g] = plain ? cr.x : 1.0;
                qn[g] = plain ? ((w >> 31) ? -cr.y : cr.y) : 0.0;
                asm volatile("" : "+v"(ai[g]), "+v"(aj[g]), "+v"(qc[g]), "+v"(qn[g]));
            }
        }
        auto rows = [&](auto check) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (base + g < R1) {
                    while (rb < nruns && base + g >= bnd) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        ++rb;
                        bnd = __builtin_amdgcn_readlane(myb, min(rb, nruns));
                    }
                    if (ai[g] != (mask << 3)) {
                        double *pu = reinterpret_cast<double *>(lds + ai[g]), *pv = reinterpret_cast<double *>(lds + aj[g]);
                        const double u = *pu, v = *pv;
                        *pu = qc[g] * u + qn[g] * v;
                        *pv = qc[g] * v - qn[g] * u;
                        if (decltype(check)::value) bad |= aj[g] == (spare << 3) && u != 0.0;
                    }
                }
            }
        };
        if (slow) rows(std::true_type{});
        else rows(std::false_type{});
#pragma unroll
        for (int g = 0; g < G; ++g) cw[g] = nw[g];
        ch = nh;
    }
    for (; rb < nruns; ++rb) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (dst_lds) {
        for (uint32_t k = threadIdx.x; k < n; k += NT) out[dst[k]] = tile[k];
    } else {
        for (uint32_t k = threadIdx.x; k < n; k += NT) out[dstpad[tbase + k]] = tile[k];
    }
    if (bad) atomicOr(flag, 1);
}
// scatter indices of a sweep: where every entry of its order sits in the NEXT sweep's tile-padded order
// (dstc[p], p = position in this sweep's order; src_next / off_next / cap_next describe the next sweep) ...
__global__ __launch_bounds__(256) void k_sec_dst_compact(const uint32_t *__restrict__ src_next, const uint32_t *__restrict__ off_next,
                                                         uint32_t cap_next, uint32_t *__restrict__ dstc) {
    const uint32_t t = blockIdx.x, e0 = off_next[t], n = off_next[t + 1] - e0;
    for (uint32_t k = threadIdx.x; k < n; k += 256u) dstc[src_next[e0 + k]] = t * cap_next + k;
}
// ... in this sweep's own tile-padded form (dstc == nullptr: the last sweep writes its order contiguously)
__global__ __launch_bounds__(256) void k_sec_dst_pad(const uint32_t *__restrict__ dstc, const uint32_t *__restrict__ off, uint32_t cap,
                                                     uint32_t *__restrict__ dstpad) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    for (uint32_t k = threadIdx.x; k < cap; k += 256u) dstpad[(size_t)t * cap + k] = k < n ? (dstc ? dstc[e0 + k] : e0 + k) : 0xffffffffu;
}

// ---- materialised <H>: construction ------------------------------------------------------------------------------------
// One thread per entry of the tile; for every x-group of the sweep D_g = sum_t c_t (-1)^{|hi & z_t|} with hi = the pair's
// member whose pivot bit is set, kept by the pair's owner.  FILL = false: ccnt[e] / xcnt[e] = elements of entry e in the
// two streams; FILL = true: element q of the row goes to base[tile, slice] + 64 q + lane, the row is padded to the
// slice's length (the lanes past the tile's last row only pad).  The coded stream is written with its values;
// k_sec_encode replaces them.
// DENSE: the partner's slot comes from a dense LDS map local key -> slot (2 bytes x 2^M: M <= 16) instead of a binary
// search in the tile's sorted keys (13 dependent LDS reads per pair: the construction was bound by them).
// The x-groups of the sweep and their terms (those with an even number of Y: the others vanish on a real state) come as 16-byte
// records, everything that does not depend on the entry precomputed (the x mask in tile-local bits), and are STAGED IN LDS when they
// fit behind the slot map (`staged`): every thread walks all of them for each of its entries, and as scalar loads from memory each
// record cost a full memory latency with four waves per SIMD to hide it — N2 UCCSD, 7 sweeps: 13.8 + 13.1 ms for the two passes.
struct SecGroupL {
    uint32_t x, lx;     // global x mask; pext(x, tile bit set)
    uint32_t t0, t1;    // its terms in the sweep's list
};
struct SecTermL {
    uint32_t z, pad;
    double c;
};
template <bool FILL, int NT, bool DENSE>
__global__ __launch_bounds__(NT) void k_sec_hbuild(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                   uint32_t map_bytes, const SecGroupL *__restrict__ groups_g, int ngroups,
                                                   const SecTermL *__restrict__ terms_g, int nterms, int staged,
                                                   uint32_t *__restrict__ ccnt,
                                                   uint32_t *__restrict__ xcnt, const uint32_t *__restrict__ cbase,
                                                   const uint32_t *__restrict__ xbase, const uint32_t *__restrict__ clen,
                                                   const uint32_t *__restrict__ xlen, const uint16_t *__restrict__ rank,
                                                   uint32_t *__restrict__ cwords, double *__restrict__ cvals,
                                                   uint32_t *__restrict__ xwords, double *__restrict__ xvals) {
    extern __shared__ uint32_t sec_lk[];
    uint16_t *slot_of = reinterpret_cast<uint16_t *>(sec_lk);
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) return;
    const uint32_t lmask = (1u << M) - 1u;
    const SecGroupL *groups = groups_g;
    const SecTermL *terms = terms_g;
    if (staged) {
        uint4 *lg = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(sec_lk) + map_bytes);
        uint4 *lt = lg + ngroups;
        for (int k = threadIdx.x; k < ngroups; k += NT) lg[k] = reinterpret_cast<const uint4 *>(groups_g)[k];
        for (int k = threadIdx.x; k < nterms; k += NT) lt[k] = reinterpret_cast<const uint4 *>(terms_g)[k];
        groups = reinterpret_cast<const SecGroupL *>(lg);
        terms = reinterpret_cast<const SecTermL *>(lt);
    }
    if (DENSE) {
        for (uint32_t k = threadIdx.x; k < (1u << M) / 2u; k += NT) sec_lk[k] = 0xffffffffu;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += NT) slot_of[keys[e0 + k] & lmask] = (uint16_t)k;
    } else {
        for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    }
    __syncthreads();
    const int nrows = FILL ? ((n + 63) & ~63) : n;
    for (int k = threadIdx.x; k < nrows; k += NT) {
        uint32_t cc = 0, xc = 0;
        const uint32_t pos = (FILL && k < n) ? rank[e0 + k] : (uint32_t)k;   // the row's place in its tile's slices
        const size_t sl = (size_t)t * SEC_HSLICES + (size_t)(pos >> 6);
        const uint32_t cb = FILL ? cbase[sl] + 4u * (pos & 63u) : 0u, xb = FILL ? xbase[sl] + (pos & 63u) : 0u;
        if (k < n) {
            const uint64_t i = sup[cid[e0 + k]];
            const uint32_t li = keys[e0 + k] & lmask;
            for (int g = 0; g < ngroups; ++g) {
                const SecGroupL gr = groups[g];
                int sj = k;
                uint32_t hi = (uint32_t)i;
                if (gr.x) {
                    const uint32_t pbit = 1u << (31 - __clz((int)gr.x));
                    const bool is_low = !((uint32_t)i & pbit);
                    const uint32_t low = is_low ? (uint32_t)i : (uint32_t)i ^ gr.x;
                    if (sec_low_owns((uint64_t)low, (uint64_t)gr.x) != is_low) continue;   // the other member keeps this element
                    hi = low ^ gr.x;
                    const uint32_t lj = li ^ gr.lx;
                    if (DENSE) {
                        const uint32_t v = slot_of[lj];
                        sj = v == 0xffffu ? -1 : (int)v;
                    } else {
                        sj = sec_find(sec_lk, n, lj);
                    }
                    if (sj < 0) continue;
                }
                double d = 0.0;
                for (uint32_t tt = gr.t0; tt < gr.t1; ++tt) {
                    const SecTermL ht = terms[tt];
                    d += (__popc(hi & ht.z) & 1) ? -ht.c : ht.c;
                }
                if (d == 0.0) continue;
                if (__popc(gr.x) >= SEC_CODED_MIN_WEIGHT) {
                    if (FILL) {
                        cwords[cb + 256u * (cc >> 2) + (cc & 3u)] = (uint32_t)sj;
                        cvals[cb + 256u * (cc >> 2) + (cc & 3u)] = 2.0 * d;
                    }
                    ++cc;
                } else {
                    if (FILL) {
                        xwords[xb + 64u * xc] = (uint32_t)sj;
                        xvals[xb + 64u * xc] = gr.x ? 2.0 * d : d;
                    }
                    ++xc;
                }
            }
        }
        if (!FILL) {
            ccnt[e0 + k] = cc;
            xcnt[e0 + k] = xc;
        } else {   // pad the row to the slice's length
            for (uint32_t q = cc; q < clen[sl]; ++q) {
                cwords[cb + 256u * (q >> 2) + (q & 3u)] = 0u;
                cvals[cb + 256u * (q >> 2) + (q & 3u)] = 0.0;
            }
            for (uint32_t q = xc; q < xlen[sl]; ++q) {
                xwords[xb + 64u * q] = 0u;
                xvals[xb + 64u * q] = 0.0;
            }
        }
    }
}
// order[e0 + p] = the entry at slice position p of its tile, rank = the inverse: entries sorted by row length (descending;
// ties by slot, so the layout is reproducible) with a bitonic sort of (65535 - length) << 16 | slot in LDS
template <int NT>
__global__ __launch_bounds__(NT) void k_sec_row_order(const uint32_t *__restrict__ off, const uint32_t *__restrict__ ccnt,
                                                      const uint32_t *__restrict__ xcnt, uint16_t *__restrict__ order,
                                                      uint16_t *__restrict__ rank) {
    extern __shared__ uint32_t sec_lk[];
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    if (n == 0) return;
    uint32_t P = 64;
    while (P < n) P <<= 1;
    for (uint32_t k = threadIdx.x; k < P; k += NT)
        sec_lk[k] = k < n ? ((65535u - min(ccnt[e0 + k] + xcnt[e0 + k], 65535u)) << 16) | k : 0xffffffffu;
    __syncthreads();
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t k = threadIdx.x; k < P / 2; k += NT) {
                const uint32_t i = 2u * k - (k & (stride - 1u)), j = i + stride;
                const bool up = !(i & size);
                const uint32_t a = sec_lk[i], b = sec_lk[j];
                if ((a > b) == up) {
                    sec_lk[i] = b;
                    sec_lk[j] = a;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t p = threadIdx.x; p < n; p += NT) {
        const uint32_t row = sec_lk[p] & 0xffffu;
        order[e0 + p] = (uint16_t)row;
        rank[e0 + row] = (uint16_t)p;
    }
}
// slice lengths: len[tile * SEC_HSLICES + s] = longest row of the slice (one wave per slice); words64 = 64 len, the input of
// the scan that lays the slices out
__global__ __launch_bounds__(256) void k_sec_slice_len(const uint32_t *__restrict__ off, const uint16_t *__restrict__ order,
                                                       const uint32_t *__restrict__ ccnt,
                                                       const uint32_t *__restrict__ xcnt, uint32_t *__restrict__ clen,
                                                       uint32_t *__restrict__ xlen, uint32_t *__restrict__ cwords64,
                                                       uint32_t *__restrict__ xwords64) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t s = wave; s < (uint32_t)SEC_HSLICES; s += 4u) {
        const uint32_t k = 64u * s + lane;
        const uint32_t row = k < n ? order[e0 + k] : 0u;
        uint32_t c = k < n ? ccnt[e0 + row] : 0u, x = k < n ? xcnt[e0 + row] : 0u;
        for (int o = 32; o > 0; o >>= 1) {
            c = max(c, (uint32_t)__shfl_xor(c, o, 64));
            x = max(x, (uint32_t)__shfl_xor(x, o, 64));
        }
        if (lane == 0) {
            const size_t sl = (size_t)t * SEC_HSLICES + s;
            c = (c + 3u) & ~3u;   // a lane reads its coded elements four at a time
            clen[sl] = c;
            xlen[sl] = x;
            cwords64[sl] = 64u * c;
            xwords64[sl] = 64u * x;
        }
    }
}
// dictionary of a sweep: sort keys = |value| (the bit pattern of a non-negative double orders like the number)
// keys of every `stride`-th value (m = ceil(n / stride) of them): the dictionary is first built from a SAMPLE of the coded stream —
// sorting the 60 M values of a sweep of the 2^22-amplitude QUCCSD support to find their few hundred distinct magnitudes was 55 of
// the 214 ms of that table build — and k_sec_encode reports a value the sample missed (then the dictionary is rebuilt from all values)
__global__ __launch_bounds__(256) void k_sec_abs_keys(const double *__restrict__ vals, uint32_t n, uint32_t stride, uint64_t *__restrict__ keys) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if ((uint64_t)e * stride < n) keys[e] = (uint64_t)__double_as_longlong(fabs(vals[(size_t)e * stride]));
}
// (an encoding attempt that is given up: the words back to their slots, the values stay explicit)
__global__ __launch_bounds__(256) void k_sec_words_slots(uint32_t *__restrict__ words, uint32_t n) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < n) words[e] &= SEC_HSLOT_MASK;
}
// idempotent: slot bits kept, sign and dictionary index rewritten; *miss = 1 when a magnitude is not in the dictionary
__global__ __launch_bounds__(256) void k_sec_encode(uint32_t *__restrict__ words, const double *__restrict__ vals, uint32_t n,
                                                    const uint64_t *__restrict__ dict, int ndict, int *__restrict__ miss) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    const double v = vals[e];
    if (v == 0.0) {   // padding: the null element
        words[e] = (uint32_t)ndict << 14;
        return;
    }
    const uint64_t key = (uint64_t)__double_as_longlong(fabs(v));
    int lo = 0, hi = ndict - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (dict[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (dict[lo] != key) *miss = 1;
    words[e] = (words[e] & SEC_HSLOT_MASK) | (v < 0.0 ? 1u << SEC_HSLOT_BITS : 0u) | ((uint32_t)lo << 14);
}

// ---- materialised <H>: evaluation --------------------------------------------------------------------------------------
// The sum over the elements of the lane's row: sum_e value_e a[slot_j]  (APPLY: also lambda[slot_j] += H_ij a_i for the
// off-diagonal elements, f64 LDS atomics on scattered slots)
struct SecSliceMeta {   // what a wave needs to know about a slice before it can fetch its elements
    uint32_t cbase, clen, xbase, xlen, row;
};
__device__ __forceinline__ SecSliceMeta sec_slice_meta(const SecHSweep &sw, size_t sl, uint32_t e0, uint32_t p, uint32_t n) {
    SecSliceMeta m;
    m.cbase = sw.cbase[sl];
    m.clen = sw.clen[sl];
    m.xbase = sw.xbase[sl];
    m.xlen = sw.xlen[sl];
    m.row = p < n ? sw.order[e0 + p] : SEC_HSLOT_MASK;
    return m;
}
// APPLY: the atomics of a block of elements are issued AFTER all of its reads, and unconditionally — padding elements (value
// zero) add to a per-lane dummy slot behind the dictionary.  (A masked atomic between the reads of consecutive elements made
// every element wait out an LDS round trip: 1.43 ms per pass at 24 qubits against 0.58 ms for the sum alone;
// tools/micro/lds_atomic.hip: the atomics themselves cost 9-20 cycles per wave instruction.)
template <bool APPLY>
__device__ __forceinline__ double sec_row_sum(const SecHSweep &sw, const SecSliceMeta &mt, uint32_t lane, uint32_t row, double ai,
                                              const double *tile, const double *dict, double *lam) {
    constexpr int NFL = APPLY ? SEC_H_INFLIGHT_APPLY : SEC_H_INFLIGHT;   // 16-byte (packed: 12-byte) loads in flight per lane
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double *dummy = const_cast<double *>(dict) + sw.ndict + 2 + lane;   // APPLY only: 64 doubles behind the dictionary (sector_h_smem)
    const double hai = 0.5 * ai;
    {
        const uint32_t L = mt.clen;   // a multiple of 4
        const uint32_t *wp = sw.cwords + mt.cbase + 4u * lane;
        if (sw.ndict) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            auto value = [&](uint32_t w) {
                double v = dict[w >> 14];
                if (w & (1u << SEC_HSLOT_BITS)) v = -v;
                return v;
            };
            auto scatter = [&](uint32_t w, double v) {
                double *to = (w >> 14) != (uint32_t)sw.ndict ? lam + (w & SEC_HSLOT_MASK) : dummy;
                __hip_atomic_fetch_add(to, v * hai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            uint32_t q = 0;
            const uint32_t *wp3 = sw.cwords + (mt.cbase >> 2) * 3u + 3u * lane;   // packed: the lane's four elements of step q / 4 are three dwords at 192 (q / 4)
            for (; q + 4u * NFL - 1u < L; q += 4u * NFL) {   // NFL 16-byte (12-byte) loads in flight per lane
                u32x4 w[NFL];
                if (sw.packed) {
                    sec_u32x3 d[NFL];
#pragma unroll
                    for (int u = 0; u < NFL; ++u) d[u] = __builtin_nontemporal_load(reinterpret_cast<const sec_u32x3 *>(wp3 + 48u * (q + 4u * (uint32_t)u)));
#pragma unroll
                    for (int u = 0; u < NFL; ++u) w[u] = sec_unpack24(d[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < NFL; ++u) w[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wp + 64u * (q + 4u * (uint32_t)u)));
                }
                double v[NFL][4];
#pragma unroll
                for (int u = 0; u < NFL; ++u) {
                    v[u][0] = value(w[u].x);
                    v[u][1] = value(w[u].y);
                    v[u][2] = value(w[u].z);
                    v[u][3] = value(w[u].w);
                    a0 += v[u][0] * tile[w[u].x & SEC_HSLOT_MASK];
                    a1 += v[u][1] * tile[w[u].y & SEC_HSLOT_MASK];
                    a2 += v[u][2] * tile[w[u].z & SEC_HSLOT_MASK];
                    a3 += v[u][3] * tile[w[u].w & SEC_HSLOT_MASK];
                }
                if (APPLY) {
#pragma unroll
                    for (int u = 0; u < NFL; ++u) {
                        scatter(w[u].x, v[u][0]);
                        scatter(w[u].y, v[u][1]);
                        scatter(w[u].z, v[u][2]);
                        scatter(w[u].w, v[u][3]);
                    }
                }
            }
            for (; q < L; q += 4u) {
                const u32x4 w = sw.packed ? sec_unpack24(*reinterpret_cast<const sec_u32x3 *>(wp3 + 48u * q)) : *reinterpret_cast<const u32x4 *>(wp + 64u * q);
                const double v0 = value(w.x), v1 = value(w.y), v2 = value(w.z), v3 = value(w.w);
                a0 += v0 * tile[w.x & SEC_HSLOT_MASK];
                a1 += v1 * tile[w.y & SEC_HSLOT_MASK];
                a2 += v2 * tile[w.z & SEC_HSLOT_MASK];
                a3 += v3 * tile[w.w & SEC_HSLOT_MASK];
                if (APPLY) {
                    scatter(w.x, v0);
                    scatter(w.y, v1);
                    scatter(w.z, v2);
                    scatter(w.w, v3);
                }
            }
        } else {
            const double *vp = sw.cvals + mt.cbase + 4u * lane;
            for (uint32_t q = 0; q < L; ++q) {
                const uint32_t at = 256u * (q >> 2) + (q & 3u);
                const uint32_t sj = wp[at] & SEC_HSLOT_MASK;
                const double v = vp[at];
                a0 += v * tile[sj];
                if (APPLY) __hip_atomic_fetch_add(v != 0.0 ? lam + sj : dummy, v * hai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    double diag = 0.0;   // APPLY: H_ii a_i enters lambda_i once, the off-diagonal parts with H_ij = value / 2
    {
        const uint32_t L = mt.xlen;
        const uint32_t *wp = sw.xwords + mt.xbase + lane;
        const double *vp = sw.xvals + mt.xbase + lane;
        for (uint32_t q = 0; q < L; ++q) {
            const uint32_t sj = wp[64u * q] & SEC_HSLOT_MASK;
            const double v = vp[64u * q];
            const bool on_diag = sj == row;
            diag += on_diag ? v * ai : 0.0;
            a1 += on_diag ? 0.0 : v * tile[sj];
            if (APPLY) __hip_atomic_fetch_add((on_diag || v == 0.0) ? dummy : lam + sj, on_diag ? 0.0 : v * hai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    const double offd = (a0 + a1) + (a2 + a3);
    return APPLY ? 0.5 * offd + diag : offd + diag;
}

// tile <- state[src[e0 + k]], k < n, in batches of eight gathers per thread: a plain `load index -> load amplitude -> LDS store`
// loop is compiled to two dependent trips to memory per iteration (15 iterations per tile at 512 threads), and a workgroup
// loads half a dozen tiles per launch
template <int NT, int NB>
__device__ __forceinline__ void sec_load_tile(double *__restrict__ tile, const double *__restrict__ state, size_t stride,
                                              const uint32_t *__restrict__ src, uint32_t e0, uint32_t n) {
    constexpr int TB = 8;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += TB * NT) {
        uint32_t ix[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) ix[j] = src[e0 + min(k0 + (uint32_t)j * NT, n - 1u)];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            double v[TB];
#pragma unroll
            for (int j = 0; j < TB; ++j) v[j] = state[(size_t)q * stride + ix[j]];
#pragma unroll
            for (int j = 0; j < TB; ++j)
                if (k0 + (uint32_t)j * NT < n) tile[(size_t)(k0 + (uint32_t)j * NT) * NB + q] = v[j];
        }
    }
}
// E = sum over the sweeps and tiles of sum_rows a_i (sum_e value_e a_j): blockIdx.y = sweep; the workgroups of a sweep
// share its tiles round robin; one wave per slice of a tile
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_expect(const double *__restrict__ state, const SecHSweep *__restrict__ sweeps,
                                                      double2 *__restrict__ partials, uint32_t tile_cap, int dbg = 0) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    double *dict = tile + ((tile_cap + 1u) & ~1u);
    const SecHSweep sw = sweeps[blockIdx.y];
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = sw.dict[k];
    if (threadIdx.x == 0) dict[sw.ndict] = 0.0;   // the null element
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    double acc = 0.0;
    for (uint32_t tt = blockIdx.x; tt < (uint32_t)sw.ntiles; tt += gridDim.x) {
        const uint32_t t = sw.torder ? sw.torder[tt] : tt;
        const uint32_t e0 = sw.off[t], n = sw.off[t + 1] - e0;
        if (n == 0) continue;
        __syncthreads();
        if (dbg != 2) sec_load_tile<NT, 1>(tile, state, 0, sw.src, e0, n);   // dbg: measurements only (1: tile loads only, 2: no tile loads, 3: metadata only)
        __syncthreads();
        if (dbg == 1) continue;
        // A slice is only a few iterations of the element loop long (24 qubits: 71 elements per row and sweep), so a wave walking
        // one slice after the other spends a good part of its time in the dependent trip base / length -> first elements: the
        // metadata of the NEXT slice of this wave is fetched while the current one is summed (0.62 -> 0.56 ms with 256 workgroups
        // per sweep).  Measured on top and dropped: the next slice's first elements fetched ahead as well (0.61 ms), two slices
        // summed in one loop (0.59 ms), eight instead of four 16-byte loads in flight (0.66 ms).
        SecSliceMeta mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + wave, e0, 64u * wave + lane, n);
        for (uint32_t s = wave; 64u * s < n; s += NW) {
            const uint32_t p = 64u * s + lane;
            const SecSliceMeta cur = mt;
            const uint32_t sn = s + NW;
            if (64u * sn < n) mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + sn, e0, 64u * sn + lane, n);
            const uint32_t row = cur.row;
            const double ai = p < n ? tile[row] : 0.0;
            if (dbg == 3) {
                acc += (double)(cur.clen + cur.xlen + cur.cbase + cur.xbase) * ai;
                continue;
            }
            acc += ai * sec_row_sum<false>(sw, cur, lane, row, ai, tile, dict, nullptr);
        }
    }
    __syncthreads();
    double2 *red = reinterpret_cast<double2 *>(sec_smem);   // (the tile is done with: no static LDS next to the dynamic block)
    const double2 tsum = block_sum<NT>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[slot] = tsum;
}

// the partial sums of k_sector_expect -> energy and orphan flag, written straight into mapped host memory: a lone evaluation ends
// with this launch instead of a reduction and two 16-byte copies behind it (an ADAPT-sized evaluation is a dozen dispatches of
// ~7 us each).  Same summation order as k_reduce.  (Measured and dropped: the LAST workgroup of k_sector_expect doing this sum
// behind an arrival counter — the device-scope release / acquire of 1800 workgroups, an L2 write-back each, cost 60-150 us.)
__global__ __launch_bounds__(256) void k_sector_finish(const double2 *__restrict__ partials, int64_t count, const int *__restrict__ flag,
                                                       double *__restrict__ host_out) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t i = threadIdx.x; i < count; i += 256) {
        acc.x += partials[i].x;
        acc.y += partials[i].y;
    }
    const double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) {
        host_out[0] = t.x;
        host_out[1] = t.y;
        host_out[2] = (double)*flag;
        __threadfence_system();
        host_out[3] = 1.0;   // "result there": what a polling host watches (set to a sentinel before the launch)
    }
}

// <H> of NB states per pass over the table (batched evaluations, ovqe_energy_batch on the sector tables): the tile holds the NB
// states interleaved (tile[slot * NB + s]), so an element's word, its dictionary value and its row serve NB states and the
// table — the bytes that bound the single-state kernel — streams once for all of them.  blockIdx.z = group of NB states
// (slices of `stride` doubles); partials[((z * sweeps + y) * groups + x) * NB + s].
template <int NT, int NB>
__global__ __launch_bounds__(NT) void k_sector_expect_batch(const double *__restrict__ states, size_t stride,
                                                            const SecHSweep *__restrict__ sweeps, double *__restrict__ partials,
                                                            uint32_t tile_cap, int zfast) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    double *dict = tile + (size_t)NB * ((tile_cap + 1u) & ~1u);
    __shared__ double2 red[NW];
    // zfast: the state group is the FASTEST grid index — the workgroups that walk the same tile's elements for different states run
    // side by side and share them through L2 / the Infinity Cache (cached loads) instead of streaming the table once per group
    const uint32_t bx = zfast ? blockIdx.y : blockIdx.x, by = zfast ? blockIdx.z : blockIdx.y, bz = zfast ? blockIdx.x : blockIdx.z;
    const uint32_t gx = zfast ? gridDim.y : gridDim.x, gy = zfast ? gridDim.z : gridDim.y;
    const SecHSweep sw = sweeps[by];
    states += (size_t)bz * NB * stride;
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = sw.dict[k];
    if (threadIdx.x == 0) dict[sw.ndict] = 0.0;   // the null element
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    double acc[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) acc[s] = 0.0;
    for (uint32_t tt = bx; tt < (uint32_t)sw.ntiles; tt += gx) {
        const uint32_t t = sw.torder ? sw.torder[tt] : tt;
        const uint32_t e0 = sw.off[t], n = sw.off[t + 1] - e0;
        if (n == 0) continue;
        __syncthreads();
        sec_load_tile<NT, NB>(tile, states, stride, sw.src, e0, n);
        __syncthreads();
        SecSliceMeta mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + wave, e0, 64u * wave + lane, n);
        for (uint32_t sl0 = wave; 64u * sl0 < n; sl0 += NW) {
            const uint32_t p = 64u * sl0 + lane;
            const SecSliceMeta cur = mt;
            const uint32_t sn = sl0 + NW;
            if (64u * sn < n) mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + sn, e0, 64u * sn + lane, n);
            const uint32_t row = p < n ? cur.row : 0u;
            double r[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) r[s] = 0.0;
            {
                const uint32_t L = cur.clen;   // a multiple of 4
                const uint32_t *wp = sw.cwords + cur.cbase + 4u * lane;
                if (sw.ndict) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    auto term = [&](uint32_t w) {
                        const double *a = tile + (size_t)(w & SEC_HSLOT_MASK) * NB;
                        double v = dict[w >> 14];
                        if (w & (1u << SEC_HSLOT_BITS)) v = -v;
#pragma unroll
                        for (int s = 0; s < NB; ++s) r[s] += v * a[s];
                    };
                    uint32_t q = 0;
                    const uint32_t *wp3 = sw.cwords + (cur.cbase >> 2) * 3u + 3u * lane;   // packed 24-bit elements: see sec_row_sum
                    for (; q + 4u * SEC_H_INFLIGHT - 1u < L; q += 4u * SEC_H_INFLIGHT) {
                        u32x4 w[SEC_H_INFLIGHT];
                        if (sw.packed) {
                            sec_u32x3 d[SEC_H_INFLIGHT];
#pragma unroll
                            for (int u = 0; u < SEC_H_INFLIGHT; ++u) d[u] = zfast ? *reinterpret_cast<const sec_u32x3 *>(wp3 + 48u * (q + 4u * (uint32_t)u))
                                                               : __builtin_nontemporal_load(reinterpret_cast<const sec_u32x3 *>(wp3 + 48u * (q + 4u * (uint32_t)u)));
#pragma unroll
                            for (int u = 0; u < SEC_H_INFLIGHT; ++u) w[u] = sec_unpack24(d[u]);
                        } else {
#pragma unroll
                        for (int u = 0; u < SEC_H_INFLIGHT; ++u) w[u] = zfast ? *reinterpret_cast<const u32x4 *>(wp + 64u * (q + 4u * (uint32_t)u))
                                                           : __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(wp + 64u * (q + 4u * (uint32_t)u)));
                        }
#pragma unroll
                        for (int u = 0; u < SEC_H_INFLIGHT; ++u) {
                            term(w[u].x);
                            term(w[u].y);
                            term(w[u].z);
                            term(w[u].w);
                        }
                    }
                    for (; q < L; q += 4u) {
                        const u32x4 w = sw.packed ? sec_unpack24(*reinterpret_cast<const sec_u32x3 *>(wp3 + 48u * q)) : *reinterpret_cast<const u32x4 *>(wp + 64u * q);
                        term(w.x);
                        term(w.y);
                        term(w.z);
                        term(w.w);
                    }
                } else {
                    const double *vp = sw.cvals + cur.cbase + 4u * lane;
                    for (uint32_t q = 0; q < L; ++q) {
                        const uint32_t at = 256u * (q >> 2) + (q & 3u);
                        const double *a = tile + (size_t)(wp[at] & SEC_HSLOT_MASK) * NB;
                        const double v = vp[at];
#pragma unroll
                        for (int s = 0; s < NB; ++s) r[s] += v * a[s];
                    }
                }
            }
            {
                const uint32_t L = cur.xlen;
                const uint32_t *wp = sw.xwords + cur.xbase + lane;
                const double *vp = sw.xvals + cur.xbase + lane;
                for (uint32_t q = 0; q < L; ++q) {   // (a diagonal element reads the row's own amplitude: value * a_i)
                    const double *a = tile + (size_t)(wp[64u * q] & SEC_HSLOT_MASK) * NB;
                    const double v = vp[64u * q];
#pragma unroll
                    for (int s = 0; s < NB; ++s) r[s] += v * a[s];
                }
            }
            if (p < n) {
#pragma unroll
                for (int s = 0; s < NB; ++s) acc[s] += tile[(size_t)row * NB + s] * r[s];
            }
        }
    }
    const size_t slot = (((size_t)bz * gy + by) * gx + bx) * NB;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        __syncthreads();
        const double2 tsum = block_sum<NT>(make_double2(acc[s], 0.0), red);
        if (threadIdx.x == 0) partials[slot + s] = tsum.x;
    }
}
// energies of a batch from the partial sums of k_sector_expect_batch: one workgroup per state, fixed order
__global__ __launch_bounds__(256) void k_sec_reduce_batch(const double *__restrict__ partials, int nparts, int NB, double constant,
                                                          double *__restrict__ energies) {
    __shared__ double2 red[4];
    const int b = blockIdx.x, z = b / NB, s = b % NB;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[((size_t)z * nparts + i) * NB + s];
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) energies[b] = t.x + constant;
}

// ---- exact gradient on the sector tables (adjoint method) ----------------------------------------------------------------
// lambda = H psi restricted to the support from the same tables: the lane's row sum is lambda_i's part from the elements
// its entry owns (a register sum), the partner's part lambda_j += H_ij a_i is an f64 LDS atomic on an LDS copy of the tile;
// the tile's result is added to lambda in the circuit's final order with f64 atomics.  The order of these additions is not
// fixed: the gradient reproduces to rounding, not bit for bit.
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_apply(const double *__restrict__ state, const SecHSweep *__restrict__ sweeps,
                                                     double *__restrict__ lam_out, uint32_t tile_cap, int sweep0 = 0, int mode = 0) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    double *tile = reinterpret_cast<double *>(sec_smem);
    double *lam = tile + ((tile_cap + 1u) & ~1u);
    double *dict = lam + ((tile_cap + 1u) & ~1u);
    const SecHSweep sw = sweeps[sweep0 + blockIdx.y];
    for (int k = threadIdx.x; k < sw.ndict; k += NT) dict[k] = sw.dict[k];
    if (threadIdx.x == 0) dict[sw.ndict] = 0.0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t tt = blockIdx.x; tt < (uint32_t)sw.ntiles; tt += gridDim.x) {
        const uint32_t t = sw.torder ? sw.torder[tt] : tt;
        const uint32_t e0 = sw.off[t], n = sw.off[t + 1] - e0;
        if (n == 0) continue;
        __syncthreads();
        sec_load_tile<NT, 1>(tile, state, 0, sw.src, e0, n);
        for (uint32_t k = threadIdx.x; k < n; k += NT) lam[k] = 0.0;
        __syncthreads();
        SecSliceMeta mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + wave, e0, 64u * wave + lane, n);
        for (uint32_t s = wave; 64u * s < n; s += NW) {
            const uint32_t p = 64u * s + lane;
            const SecSliceMeta cur = mt;
            const uint32_t sn = s + NW;
            if (64u * sn < n) mt = sec_slice_meta(sw, (size_t)t * SEC_HSLICES + sn, e0, 64u * sn + lane, n);
            const uint32_t row = cur.row;
            const double ai = p < n ? tile[row] : 0.0;
            const double ci = sec_row_sum<true>(sw, cur, lane, row, ai, tile, dict, lam);
            if (p < n) __hip_atomic_fetch_add(&lam[row], ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        // mode 0: every sweep of the launch (blockIdx.y) adds its tiles' results with f64 atomics — 71 M scattered global atomics for the
        // N2 QUCCSD tables: 1.1 of the pass's 3.5 ms.  Modes 1 / 2 (one sweep per launch, launches in sequence): the sweep's `src` is a
        // permutation of the support, so its tiles never meet — plain read-add-write, or plain stores for the first sweep; the order of
        // the additions is then fixed.
        if (mode == 2) {
            for (uint32_t k = threadIdx.x; k < n; k += NT) lam_out[sw.src[e0 + k]] = lam[k];
        } else if (mode == 1) {
            for (uint32_t k = threadIdx.x; k < n; k += NT) {
                const uint32_t d = sw.src[e0 + k];
                lam_out[d] += lam[k];
            }
        } else {
            for (uint32_t k = threadIdx.x; k < n; k += NT) unsafeAtomicAdd(&lam_out[sw.src[e0 + k]], lam[k]);
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
    __syncthreads();   // (the tile loads follow the first chunk's pair-word loads below: one round trip for both)
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
    if (oa < ob) fetch(oa, ob);
    for (int k = threadIdx.x; k < n; k += NT) {
        tp[k] = psi_in[e0 + k];
        tl[k] = lam_in[e0 + k];
    }
    if (oa < ob) stash(wbuf);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
// Second form of the backward sweep, on the tables of k_sector_sweep2 (64-bit words in registers, rounds numbered at build
// time, walked from the last chunk / round to the first): a word is applied exactly once, so its gradient term
// sigma (lambda_i psi_j - lambda_j psi_i) stays in a register until its chunk is done and the chunk's terms are then summed
// by table entry inside every wave (DPP sums over the lanes that share an entry, usually all 64) into the wave's own row
// of partial sums — the first form pays a select chain over eight patterns per pair and a wave sum per op and pattern in
// every wave.  psi / lambda come in this sweep's tile-padded order (in_compact: contiguous, the last sweep — what the forward
// pass and lambda = H psi leave) and go out into the PREVIOUS sweep's tile-padded order (bdst; null: first sweep, no output).
template <int NT, int WPT>
__global__ __launch_bounds__(NT) void k_sector_adjoint2(const double *__restrict__ psi_in, const double *__restrict__ lam_in,
                                                        double *__restrict__ psi_out, double *__restrict__ lam_out, int in_compact,
                                                        const uint32_t *__restrict__ bdst, const uint32_t *__restrict__ off,
                                                        const uint32_t *__restrict__ poff, int nops, const uint64_t *__restrict__ wide,
                                                        const uint16_t *__restrict__ rounds, uint32_t maxchunks,
                                                        const RotParam *__restrict__ rp, int rot0, int nrot, uint32_t tile_cap,
                                                        double *__restrict__ wpart, int wstride) {
    constexpr uint32_t CH = (uint32_t)NT * WPT;
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    const uint32_t capp = (tile_cap + 2u) & ~1u;
    double *tp = reinterpret_cast<double *>(sec_smem);
    double *tl = tp + capp;
    double2 *cs = reinterpret_cast<double2 *>(tl + capp);
    double *wacc = reinterpret_cast<double *>(cs + nrot);   // [NW][nrot]
    uint32_t *dst = reinterpret_cast<uint32_t *>(wacc + (size_t)NW * nrot);
    uint32_t *nround = dst + tile_cap;   // [maxchunks]
    const uint32_t t = blockIdx.x;
    const uint32_t e0 = off[t];
    const uint32_t n = off[t + 1] - e0;
    if (n == 0) return;
    const uint32_t pbase = poff[(size_t)t * (nops + 1)], ptot = poff[(size_t)t * (nops + 1) + nops] - pbase;
    const uint32_t nchunks = (ptot + CH - 1u) / CH;
    const size_t tbase = (size_t)t * tile_cap, ibase = in_compact ? (size_t)e0 : tbase;
    const uint64_t *wp = wide + pbase;
    const uint32_t lane = threadIdx.x & 63u;
    double *mine = wacc + (size_t)(threadIdx.x >> 6) * nrot;
    uint64_t wa[WPT], wb[WPT];
    // ---- one trip to memory: the last chunk's words, both tiles, the scatter indices, cos/sin, rounds per chunk
    const uint32_t lastc = nchunks ? nchunks - 1u : 0u;
#pragma unroll
    for (int r = 0; r < WPT; ++r) wa[r] = wp[min(lastc * CH + threadIdx.x + (uint32_t)r * NT, ptot ? ptot - 1u : 0u)];
    constexpr int TB = 4;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += TB * NT) {
        uint32_t d[TB];
        double u[TB], v[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const uint32_t k = min(k0 + (uint32_t)j * NT, n - 1u);
            d[j] = bdst ? bdst[tbase + k] : 0u;
            u[j] = psi_in[ibase + k];
            v[j] = lam_in[ibase + k];
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const uint32_t k = k0 + (uint32_t)j * NT;
            if (k < n) {
                tp[k] = u[j];
                tl[k] = v[j];
                dst[k] = d[j];
            }
        }
    }
    for (int r0 = threadIdx.x; r0 < nrot; r0 += NT) {
        const RotParam ra = rp[rot0 + r0];
        cs[r0] = make_double2(ra.c, ra.s);
    }
    for (int r0 = threadIdx.x; r0 < NW * nrot; r0 += NT) wacc[r0] = 0.0;
    if (threadIdx.x < maxchunks) nround[threadIdx.x] = rounds[(size_t)t * maxchunks + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (uint32_t c = nchunks; c-- > 0u;) {
        // the chunk before this one: in flight while this chunk is applied
        const uint32_t pbeg = c ? (c - 1u) * CH : 0u;
#pragma unroll
        for (int r = 0; r < WPT; ++r) wb[r] = wp[min(pbeg + threadIdx.x + (uint32_t)r * NT, ptot - 1u)];
        uint32_t qs[WPT], qr[WPT], qe[WPT];
        double qc[WPT], qn[WPT], g[WPT];
#pragma unroll
        for (int r = 0; r < WPT; ++r) {
            qe[r] = (uint32_t)(wa[r] >> 33) & 0xfffu;
            const double2 cr = cs[qe[r]];
            const bool live = c * CH + threadIdx.x + (uint32_t)r * NT < ptot;
            const bool orphan = ((uint32_t)(wa[r] >> 16) & 0xffffu) == 0xffffu;   // partner outside the support: amplitude zero, no term
            qs[r] = (uint32_t)wa[r];
            qr[r] = (live && !orphan) ? ((uint32_t)(wa[r] >> 45) & 0xfffu) : SEC_NO_ROUND;
            qc[r] = cr.x;
            qn[r] = ((uint32_t)(wa[r] >> 32) & 1u) ? -cr.y : cr.y;
            g[r] = 0.0;
        }
        const uint32_t R = nround[c];
        for (uint32_t q = R; q-- > 0u;) {
#pragma unroll
            for (int r = 0; r < WPT; ++r) {
                if (qr[r] == q) {
                    const uint32_t si = qs[r] & 0xffffu, sj = qs[r] >> 16;
                    const double u1 = tp[si], v1 = tp[sj], lu = tl[si], lv = tl[sj];
                    const double gg = lu * v1 - lv * u1;
                    g[r] = ((uint32_t)(wa[r] >> 32) & 1u) ? -gg : gg;
                    tp[si] = qc[r] * u1 - qn[r] * v1;
                    tp[sj] = qc[r] * v1 + qn[r] * u1;
                    tl[si] = qc[r] * lu - qn[r] * lv;
                    tl[sj] = qc[r] * lv + qn[r] * lu;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // the chunk's gradient terms, summed by table entry inside the wave (its lanes hold consecutive words: mostly one entry)
#pragma unroll
        for (int r = 0; r < WPT; ++r) {
            uint64_t todo = __ballot(qr[r] != SEC_NO_ROUND);
            while (todo) {
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)qe[r], __ffsll((long long)todo) - 1);
                const bool m = qr[r] != SEC_NO_ROUND && qe[r] == e;
                const double tsum = sec_wave_sum63(m ? g[r] : 0.0);
                if (lane == 63u) mine[e] += tsum;
                todo &= ~__ballot(m);
            }
        }
#pragma unroll
        for (int r = 0; r < WPT; ++r) wa[r] = wb[r];
    }
    __syncthreads();
    double *wout = wpart + (size_t)t * wstride + rot0;   // [tile][table entry], zeroed by the host
    for (int r = threadIdx.x; r < nrot; r += NT) {
        double tsum = 0.0;
        for (int w = 0; w < NW; ++w) tsum += wacc[(size_t)w * nrot + r];
        wout[r] = tsum;
    }
    if (bdst) {
        for (uint32_t k = threadIdx.x; k < n; k += NT) {
            const uint32_t d = dst[k];
            psi_out[d] = tp[k];
            lam_out[d] = tl[k];
        }
    }
}
// bdst of k_sector_adjoint2: padpos[p] = where position p of a sweep's order sits in its tile-padded form ...
__global__ __launch_bounds__(256) void k_sec_padpos(const uint32_t *__restrict__ off, uint32_t cap, uint32_t *__restrict__ padpos) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    for (uint32_t k = threadIdx.x; k < n; k += 256u) padpos[e0 + k] = t * cap + k;
}
// ... and, tile-padded for this sweep, where its entries sit in the PREVIOUS sweep's tile-padded form
__global__ __launch_bounds__(256) void k_sec_bdst(const uint32_t *__restrict__ src, const uint32_t *__restrict__ off, uint32_t cap,
                                                  const uint32_t *__restrict__ padpos_prev, uint32_t *__restrict__ bdst) {
    const uint32_t t = blockIdx.x, e0 = off[t], n = off[t + 1] - e0;
    for (uint32_t k = threadIdx.x; k < cap; k += 256u) bdst[(size_t)t * cap + k] = k < n ? padpos_prev[src[e0 + k]] : 0xffffffffu;
}
// w[r] = sum over the tiles of wpart[tile][r] (fixed order), r over the whole angle table: 64 entries per workgroup, the
// tiles dealt to its four waves, the four partial sums added in wave order
__global__ __launch_bounds__(256) void k_sec_reduce_w(const double *__restrict__ wpart, uint32_t ntiles, int nrot,
                                                      double *__restrict__ w) {
    __shared__ double part[4][64];
    const int r = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double tsum = 0.0;
    if (r < nrot)
        for (uint32_t t = g; t < ntiles; t += 4u) tsum += wpart[(size_t)t * nrot + r];
    part[g][threadIdx.x & 63] = tsum;
    __syncthreads();
    if (g == 0 && r < nrot) w[r] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---- sigma = H psi of the ADAPT screens on the materialised Hamiltonian of a symmetry sector (sector_host.inc build_screen_sector)
__global__ __launch_bounds__(256) void k_scr_narrow(const uint64_t *__restrict__ idx, uint32_t K, uint32_t *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < K) out[k] = (uint32_t)idx[k];
}
__global__ __launch_bounds__(256) void k_scr_iota(uint32_t *__restrict__ out, uint32_t K) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < K) out[k] = k;
}
// the listed amplitudes of psi into the sector's compact vector (zeroed by the caller); flag |= 1: an index outside the
// sector, |= 2: an amplitude with an imaginary part — the caller then takes the register path
__global__ __launch_bounds__(256) void k_scr_compact(const uint64_t *__restrict__ idx, const double2 *__restrict__ val, uint64_t count,
                                                     const uint32_t *__restrict__ sup, uint32_t K, double *__restrict__ psic,
                                                     int *__restrict__ flag) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= count) return;
    const uint32_t want = (uint32_t)idx[e];
    uint32_t lo = 0, hi = K;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sup[mid] < want) lo = mid + 1u;
        else hi = mid;
    }
    const double2 a = val[e];
    if ((idx[e] >> 32) != 0 || lo >= K || sup[lo] != want) {
        atomicOr(flag, 1);
        return;
    }
    if (a.y != 0.0) atomicOr(flag, 2);
    psic[lo] = a.x;
}
// sigma on the register (zeroed by the caller) from the compact product: sig[sup[k]] = sigma_c[k] + ident * psi_c[k]
__global__ __launch_bounds__(256) void k_scr_scatter(double2 *__restrict__ sig, const uint32_t *__restrict__ sup, uint32_t K,
                                                     const double *__restrict__ sigc, const double *__restrict__ psic, double ident) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k < K) sig[sup[k]] = make_double2(sigc[k] + ident * psic[k], 0.0);
}

// ---- Lanczos on the support (ovqe_sector_ground_state): vectors of K doubles in the circuit's final order --------------
__device__ __forceinline__ double sec_unit_pm1(uint64_t seed, uint32_t k) {
    uint64_t z = seed + 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1u);   // splitmix64
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(int64_t)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;   // [-1, 1)
}
// start vector: seeded noise on the entries whose `reach` value is non-zero (the block of H that holds the reference
// determinant, k_sec_reach_step), exact zeros elsewhere — H never leaves that block, so neither do the Lanczos vectors
__global__ __launch_bounds__(256) void k_sec_randomize(double *__restrict__ v, const double *__restrict__ reach, uint32_t K, uint64_t seed,
                                                       double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        const double x = reach[k] != 0.0 ? sec_unit_pm1(seed, k) : 0.0;
        v[k] = x;
        acc += x * x;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// reachability from the reference determinant through the non-zero elements of the materialised Hamiltonian: r = e_hf, then
// r <- positive noise on {r != 0 or (H r) != 0} until the set stops growing (positive, irrational-ish values: a sum of elements
// of mixed sign vanishes on a null set only).  partials.x = members.
__global__ __launch_bounds__(256) void k_sec_reach_init(double *__restrict__ r, uint32_t K, uint32_t hf_pos) {
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) r[k] = k == hf_pos ? 1.5 : 0.0;
}
__global__ __launch_bounds__(256) void k_sec_reach_step(double *__restrict__ r, const double *__restrict__ w, uint32_t K, uint64_t salt,
                                                        double thresh, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        // thresh: rounding residues of cancelling strings (the XX and YY coefficients of a hopping term agree to an ulp, not
        // exactly: <N + 2| H |N> ~ 1e-17) are not connections
        const bool in = r[k] != 0.0 || fabs(w[k]) > thresh;
        r[k] = in ? 1.5 + 0.5 * sec_unit_pm1(salt, k) : 0.0;   // [1, 2)
        acc += in ? 1.0 : 0.0;
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
// w -= alpha v + beta vprev; partials = |w|^2.  mask (optional): entries with mask[k] == 0 are projected out (the operator
// is P H P, P = the block of the reference determinant)
__global__ __launch_bounds__(256) void k_sec_lanczos_update(double *__restrict__ w, const double *__restrict__ v,
                                                            const double *__restrict__ vprev, double alpha, double beta, uint32_t K,
                                                            double2 *__restrict__ partials, const double *__restrict__ mask = nullptr) {
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        double x = w[k] - alpha * v[k];
        if (vprev) x -= beta * vprev[k];
        if (mask && mask[k] == 0.0) x = 0.0;
        w[k] = x;
        acc += x * x;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// a compact vector (final circuit order) as a dense complex state: state[sup[cid[e]]] = (v[e], 0)  (the eigenvector of
// ovqe_sector_ground_state for ovqe_get_state / ovqe_get_support)
__global__ __launch_bounds__(256) void k_sec_scatter_dense(const double *__restrict__ v, const uint32_t *__restrict__ cid,
                                                           const uint32_t *__restrict__ sup, uint32_t K, double2 *__restrict__ state) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) state[sup[cid ? cid[e] : e]] = make_double2(v[e], 0.0);   // cid == nullptr: v is in ascending-index order
}
// the compact state back in canonical (ascending index) order, e.g. for ovqe_get_state-like consumers and tests
__global__ __launch_bounds__(256) void k_sec_scatter(const double *__restrict__ in, const uint32_t *__restrict__ cid, uint32_t K,
                                                     double *__restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) out[cid[e]] = in[e];
}


// ---- circuit sweeps on a REGULAR support (round 4) ----------------------------------------------------------------------
// When the support of a program is a full coset of its Z2 symmetries — every rotation string commutes with t independent
// Z-type operators Z^{g_f} and the states fill the whole common eigenspace: 2^(n-t) amplitudes, e.g. the spin-parity quarter
// of the register that the reference's QUCCSD templates populate (ref:openvqe/common_files/circuit.py:13-106 do not conserve
// the particle number) — the pair lists describe a pattern that bit arithmetic gives for free.  g_f has its LOWEST bit f (the
// "free" bit) to itself, so a member is fixed by its other bits, i_f = s_f ^ parity(i & G_f), G_f = g_f \ {f}; with the free
// bits inside every sweep's tile bit set S, the sorted tile IS the m = |S| - t bit register of the kept inside bits:
// slot = pext(i, S \ free), every tile full, partner slot = slot ^ xs.  A thread takes a GROUP — the 2^w slots that differ on
// the op's w kept mixing bits — into registers, rotates its 2^(w-1) pairs and writes them back: two LDS accesses per amplitude
// and op, no pair words (the reference's templates at 24 qubits: 14 GB per evaluation), no staging buffers.
//   sign of a pair  = parity(slot & zin) ^ parity(tile & zt) ^ (bit 31 of its table entry)
//   selector sigma_f (free bits INSIDE the op's x mask: which member of the pair matches which pattern depends on the value
//   of the free bit, i.e. on a parity over kept bits outside x) = parity(slot & sel_in[f]) ^ parity(tile & sel_t[f]);
//   table entry (c, s) of pair q under selector sigma: cs[tab + sigma * 2^(w-1) + q]; (1, 0) where no pattern is active.
__device__ __forceinline__ double sec_flip(double v, uint32_t signbit) {   // v with its sign bit xor-ed (signbit = 0 or 0x80000000)
    return __hiloint2double(__double2hiint(v) ^ (int)signbit, __double2loint(v));
}

// Per-evaluation (c, s) table of all sweeps in global memory: entry e = (cos, +-sin) of its angle-table entry, or (1, 0).  The
// sweeps read an op's entries with SCALAR loads when no selector is involved (they are wave-uniform), so the LDS carries
// amplitudes only.
__global__ __launch_bounds__(256) void k_sec_reg_angles(const uint32_t *__restrict__ emap, uint32_t n, const RotParam *__restrict__ rp,
                                                        double2 *__restrict__ tab) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= n) return;
    const uint32_t m = emap[e];
    double2 r = make_double2(1.0, 0.0);
    if (m != 0xffffffffu) {
        const RotParam rr = rp[m & 0x7fffffffu];
        r = make_double2(rr.c, (m >> 31) ? -rr.s : rr.s);
    }
    tab[e] = r;
}

// group words (host-built, SEC_REG_GSTRIDE per op): bits 0..15 = BYTE address of the group's swizzled base slot, bit 16 = its
// parity on zin, bits 17 / 18 = its parities on sel_in[0] / sel_in[1] — the whole per-group index arithmetic is one load.
// (c, s) table: SEC_REG_TSTRIDE = 8 entries per op — 2^nsel variants of 2^(w-1) pairs, and w + nsel <= 4 — read with scalar loads.
template <int NT, int W, int NSEL>
__device__ __forceinline__ void sec_reg_apply(double *__restrict__ tile, const double2 *__restrict__ T, const SecRegHead &op, uint32_t nslots,
                                              uint32_t tnum, int dbg, const uint32_t *__restrict__ gwo, uint32_t wd0, uint32_t wd1) {
    constexpr int NA = 1 << W, NP = NA / 2;
    static_assert((NP << NSEL) <= 8, "an op has at most 8 table entries");
    const uint32_t tz = __popc(tnum & op.zt) & 1u;
    const uint32_t ts0 = NSEL > 0 ? (__popc(tnum & op.sel_t[0]) & 1u) : 0u, ts1 = NSEL > 1 ? (__popc(tnum & op.sel_t[1]) & 1u) : 0u;
    uint32_t dep[NA];   // byte offsets of the members (wave-uniform)
#pragma unroll
    for (int e = 0; e < NA; ++e) dep[e] = (op.dep[e >> 1] >> (16 * (e & 1))) & 0xffffu;
    char *tb = reinterpret_cast<char *>(tile);
    for (uint32_t g = threadIdx.x, it = 0; g < (nslots >> W); g += NT, ++it) {
        const uint32_t wd = it == 0 ? wd0 : (it == 1 ? wd1 : gwo[g]);   // (the first two arrived under the previous op's barrier)
        const uint32_t sb = wd & 0xffffu;
        const uint32_t neg = (((wd >> 16) & 1u) ^ tz) << 31;
        double a[NA];
#pragma unroll
        for (int e = 0; e < NA; ++e) a[e] = *reinterpret_cast<const double *>(tb + (sb ^ dep[e]));
        // this group's table entries: the op's 2^(w-1) pairs, of the variant its selector picks (a uniform address without selector)
        uint32_t sel = 0;
        if constexpr (NSEL > 0) sel = ((wd >> 17) & 1u) ^ ts0;
        if constexpr (NSEL > 1) sel |= (((wd >> 18) & 1u) ^ ts1) << 1;
        const double2 *Tl = T + sel * NP;
        double2 r[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) r[q] = Tl[q];
        // the group's sign on the second members: (u, sigma v) rotates by the plain (c, s)
        // the group's sign sigma on the second members: (u, sigma v) rotated by (c, s) and signed back = (u, v) rotated by (c, sigma s)
        if (dbg != 2)   // (measurement: 2 = no arithmetic, 3 = no stores)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double u = a[q], v = a[NA - 1 - q], sn = sec_flip(r[q].y, neg);
            a[q] = r[q].x * u + sn * v;
            a[NA - 1 - q] = r[q].x * v - sn * u;
        }
        if (dbg != 3)
#pragma unroll
        for (int e = 0; e < NA; ++e) *reinterpret_cast<double *>(tb + (sb ^ dep[e])) = a[e];
    }
}

// A BLOCK of two consecutive ops A, B with three kept mixing bits each, two of them shared (the reference's doubles are listed with
// three of their four indices running slowest): the 16 slots over (shared 0, shared 1, own A, own B) in registers, A on the two
// halves own B = 0 / 1, then B on the two halves own A = 0 / 1 — one read and one write of the tile, one barrier, for two ops.
// Group word: bit 16 / 17 = A's sign / selector parity of the base slot, 18 / 19 = B's.
template <int NT, int NSEL, bool MOVE_A = true, bool MOVE_B = true>
__device__ __forceinline__ void sec_reg_apply_pair(double *__restrict__ tile, const double2 *__restrict__ TA, const double2 *__restrict__ TB,
                                                   const SecRegHead &op, uint32_t nslots, uint32_t tnum, int dbg,
                                                   const uint32_t *__restrict__ gwo, uint32_t wd0, uint32_t wd1) {
    const uint32_t fl = op.pad[0];
    const uint32_t tzA = __popc(tnum & op.zt) & 1u, tzB = __popc(tnum & op.pad[1]) & 1u;
    const uint32_t tsA = NSEL ? (__popc(tnum & op.sel_t[0]) & 1u) : 0u, tsB = NSEL ? (__popc(tnum & op.pad[2]) & 1u) : 0u;
    uint32_t dep[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) dep[e] = (op.dep[e >> 1] >> (16 * (e & 1))) & 0xffffu;
    char *tb = reinterpret_cast<char *>(tile);
    for (uint32_t g = threadIdx.x, it = 0; g < (nslots >> 4); g += NT, ++it) {
        const uint32_t wd = it == 0 ? wd0 : (it == 1 ? wd1 : gwo[g]);
        const uint32_t sb = wd & 0xffffu;
        double a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = *reinterpret_cast<const double *>(tb + (sb ^ dep[e]));
        if (dbg != 2) {
            // the second half of an op takes other table entries than the first only when the other op's own bit sits on this op's
            // selector mask (MOVE_*: compile-time, so that the four LDS reads of the entries are not repeated where they are the same)
            {
                const uint32_t sel0 = NSEL ? (((wd >> 17) & 1u) ^ tsA) & 1u : 0u;
                double2 r0[4], r1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r0[q] = TA[sel0 * 4 + q];
                if constexpr (NSEL && MOVE_A) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r1[q] = TA[(sel0 ^ 1u) * 4 + q];
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {   // op A on the half own B = v: members 8 v + (shared, own A)
                    const uint32_t neg = (((wd >> 16) & 1u) ^ tzA ^ ((uint32_t)v & fl)) << 31;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double2 r = (NSEL && MOVE_A && v) ? r1[q] : r0[q];
                        const double u = a[8 * v + q], w = a[8 * v + 7 - q], sn = sec_flip(r.y, neg);
                        a[8 * v + q] = r.x * u + sn * w;
                        a[8 * v + 7 - q] = r.x * w - sn * u;
                    }
                }
            }
            {
                const uint32_t sel0 = NSEL ? (((wd >> 19) & 1u) ^ tsB) & 1u : 0u;
                double2 r0[4], r1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r0[q] = TB[sel0 * 4 + q];
                if constexpr (NSEL && MOVE_B) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r1[q] = TB[(sel0 ^ 1u) * 4 + q];
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {   // op B on the half own A = v: pairs (shared = q, own B = 0) <-> (shared = ~q, own B = 1)
                    const uint32_t neg = (((wd >> 18) & 1u) ^ tzB ^ ((uint32_t)v & (fl >> 2))) << 31;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double2 r = (NSEL && MOVE_B && v) ? r1[q] : r0[q];
                        const double u = a[4 * v + q], w = a[4 * v + 11 - q], sn = sec_flip(r.y, neg);
                        a[4 * v + q] = r.x * u + sn * w;
                        a[4 * v + 11 - q] = r.x * w - sn * u;
                    }
                }
            }
        }
        if (dbg != 3)
#pragma unroll
        for (int e = 0; e < 16; ++e) *reinterpret_cast<double *>(tb + (sb ^ dep[e])) = a[e];
    }
}

// one sweep: gather the tile from the previous sweep's order (srcpad: tile-padded gather indices; nullptr: |hf> at hf_pos),
// apply the sweep's ops group by group, write the tile back contiguously (tile t = positions [t 2^m, (t + 1) 2^m) of this
// sweep's order).  tg = the sweep's part of the (c, s) table (k_sec_reg_angles: 8 entries per op), staged in LDS behind the
// tile — an op's entries are read next to its amplitudes, by every lane from the address its selector gives —; gw = the sweep's
// group words.  What an op waits for is its LDS traffic only: the record of op o + 1 and this thread's first two group words of it
// are requested while op o runs.
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_sweep_reg(const double *__restrict__ in, double *__restrict__ out,
                                                         const uint32_t *__restrict__ srcpad, const SecRegOp *__restrict__ ops, int nops,
                                                         const uint32_t *__restrict__ gw, const double2 *__restrict__ tg, int mbits,
                                                         uint32_t hf_pos, int dbg, const uint16_t *__restrict__ gslot,
                                                         const uint16_t *__restrict__ oslot, int sync_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    const uint32_t nslots = 1u << mbits, t = blockIdx.x;
    double *tile = reinterpret_cast<double *>(sec_smem);
    double2 *cs = reinterpret_cast<double2 *>(tile + nslots);   // [nops][8]
    const uint32_t flagmask = sync_all ? 0u : (1u << 27);        // w_nsel bit 27: the next unit follows without a barrier (sv_regular_host.hpp)
    const size_t e0 = (size_t)t * nslots;
    if (srcpad) {
        // gather step j takes slot gslot[j] from position srcpad[j] of the previous sweep's buffer (gslot == nullptr: slot j).  The
        // host orders the steps — and the previous sweep its stores (oslot) — by the kept bits the two sweeps SHARE, lowest first:
        // consecutive steps read consecutive positions in runs of 2^|shared| (10 or 11 of the 12 slot bits for the reference's
        // QUCCSD list) instead of 8 bytes per 64-byte line
        const uint32_t *sp = srcpad + e0;
        for (uint32_t k0 = threadIdx.x; k0 < nslots; k0 += 4 * NT) {   // four gathers in flight per thread
            uint32_t gi[4], ks[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t j = k0 + r * NT < nslots ? k0 + r * NT : 0u;
                gi[r] = sp[j];
                ks[r] = gslot ? (uint32_t)gslot[j] : j;
            }
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = in[gi[r]];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + r * NT < nslots) tile[sec_reg_swz(ks[r])] = v[r];
        }
    } else {
        for (uint32_t k = threadIdx.x; k < nslots; k += NT) tile[sec_reg_swz(k)] = (e0 + k == hf_pos) ? 1.0 : 0.0;
    }
    for (uint32_t e = threadIdx.x; e < (uint32_t)nops * SEC_REG_TSTRIDE; e += NT) cs[e] = tg[e];
    __syncthreads();
    if (nops > 0) {
        SecRegHead cur = *reinterpret_cast<const SecRegHead *>(ops);
        uint32_t wd0 = gw[threadIdx.x], wd1 = gw[threadIdx.x + NT];
        for (int o = 0; o < nops;) {
            const int step = ((cur.w_nsel >> 24) & 1u) ? 2 : 1;   // a block of two ops
            const int on = o + step < nops ? o + step : o;
            const SecRegHead nxt = *reinterpret_cast<const SecRegHead *>(ops + on);
            const uint32_t nw0 = gw[(size_t)on * SEC_REG_GSTRIDE + threadIdx.x], nw1 = gw[(size_t)on * SEC_REG_GSTRIDE + threadIdx.x + NT];
            const uint32_t *gwo = gw + (size_t)o * SEC_REG_GSTRIDE;
            const double2 *T = cs + (size_t)o * SEC_REG_TSTRIDE;
            switch (cur.w_nsel & ~(1u << 27)) {
            case 3 | (1 << 24): sec_reg_apply_pair<NT, 0>(tile, T, T + SEC_REG_TSTRIDE, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3 | (1 << 16) | (1 << 24): sec_reg_apply_pair<NT, 1, false, false>(tile, T, T + SEC_REG_TSTRIDE, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3 | (1 << 16) | (1 << 24) | (1 << 25): sec_reg_apply_pair<NT, 1, true, false>(tile, T, T + SEC_REG_TSTRIDE, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3 | (1 << 16) | (1 << 24) | (1 << 26): sec_reg_apply_pair<NT, 1, false, true>(tile, T, T + SEC_REG_TSTRIDE, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3 | (1 << 16) | (1 << 24) | (3 << 25): sec_reg_apply_pair<NT, 1, true, true>(tile, T, T + SEC_REG_TSTRIDE, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 4: sec_reg_apply<NT, 4, 0>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3: sec_reg_apply<NT, 3, 0>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 3 | (1 << 16): sec_reg_apply<NT, 3, 1>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 2: sec_reg_apply<NT, 2, 0>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 2 | (1 << 16): sec_reg_apply<NT, 2, 1>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 2 | (2 << 16): sec_reg_apply<NT, 2, 2>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 1: sec_reg_apply<NT, 1, 0>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            case 1 | (1 << 16): sec_reg_apply<NT, 1, 1>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            default: sec_reg_apply<NT, 1, 2>(tile, T, cur, nslots, t, dbg, gwo, wd0, wd1); break;
            }
            // inside a run the waves own disjoint slot sets and the LDS pipeline keeps a wave's accesses in order: no barrier, no wait
            if (!(cur.w_nsel & flagmask) && dbg != 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (measurement: 4 = no barriers, results wrong)
            cur = nxt;
            wd0 = nw0;
            wd1 = nw1;
            o += step;
        }
    }
    if (oslot) {
        for (uint32_t j = threadIdx.x; j < nslots; j += NT) out[e0 + j] = tile[sec_reg_swz(oslot[j])];
    } else {
        for (uint32_t k = threadIdx.x; k < nslots; k += NT) out[e0 + k] = tile[sec_reg_swz(k)];
    }
}

// ---- backward sweeps on a regular support (ovqe_energy_gradient): psi and lambda = H psi walk the program backwards together.
// Per op, on the states AFTER it: w[entry] += sigma (lambda_A psi_B - lambda_B psi_A) over its pairs (dE/dphi of the entry's angle
// is twice that), then both states are rotated back.  A lane's eight per-entry sums are reduced over the wave by a reduce-scatter
// (three exchange-and-halve steps, then three butterfly steps: ten additions instead of forty-eight), the four waves' totals meet in
// LDS and lanes 0..7 of wave 0 store the tile's partial sums of the op: wpart[tile][op][8].
// lane l <- lane l ^ 1, ^ 2, ^ 4 by DPP (quad permutations; a row shift by four lanes each way under complementary bank masks): no
// LDS crossbar — the __shfl_xor these replace were 20 ds_bpermute per reduction on the pipe the kernel is bound by (round 5)
template <int CTRL>
__device__ __forceinline__ double sec_dpp_quad(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sec_dpp_xor4(double v) {
    // row_shr:4 (lane i <- i - 4) into the lanes with bit 2 set (banks 1 and 3), then row_shl:4 (lane i <- i + 4) into the others
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x114, 0xf, 0xa, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x114, 0xf, 0xa, false);
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x104, 0xf, 0x5, false);
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x104, 0xf, 0x5, false);
    return __hiloint2double(hi, lo);
}
// -> in the lanes with bit 3 clear: the total over the lane's ROW of 16 of entry 4 b0 + 2 b1 + b2 (b = lane bits): a reduce-scatter
// (three exchange-and-halve steps: ten additions instead of forty-eight) and one row shift by eight lanes
__device__ __forceinline__ double sec_reduce8(const double (&c)[8]) {
    const int lane = threadIdx.x & 63;
    double t4[4], t2[2], t1;
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] = (hi ? c[4 + j] : c[j]) + sec_dpp_quad<0xB1>(hi ? c[j] : c[4 + j]);   // quad_perm [1,0,3,2]
    }
    {
        const bool hi = lane & 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) t2[j] = (hi ? t4[2 + j] : t4[j]) + sec_dpp_quad<0x4E>(hi ? t4[j] : t4[2 + j]);   // quad_perm [2,3,0,1]
    }
    {
        const bool hi = lane & 4;
        t1 = (hi ? t2[1] : t2[0]) + sec_dpp_xor4(hi ? t2[0] : t2[1]);
    }
    {   // row_shl:8: lane i <- i + 8 (the upper half of a row reads nothing: zero)
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(t1), 0x108, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(t1), 0x108, 0xf, 0xf, false);
        t1 += __hiloint2double(hi, lo);
    }
    return t1;
}
// Backward sweep on the streams of the third form (k_sector_sweep3): a wave walks its rows from the last to the first, workgroup
// barriers at the run boundaries only.  A row is one op (usually one table entry: its header says when not), so the gradient terms
// sigma (lambda_i psi_j - lambda_j psi_i) of a BATCH of eight rows are summed over the wave together — a reduce-scatter (ten
// additions for eight sums, sec_reduce8) whose four partial totals per entry are added to the wave's row of partial sums by LDS
// atomics (rows of the same op share an entry).  psi / lambda come and go as in k_sector_adjoint2.  (Order of summation: fixed for
// a given build of the streams, not across builds — the lanes of a row are filled by LDS atomics, see k_sec_wave_plan.)
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_adjoint3(const double *__restrict__ psi_in, const double *__restrict__ lam_in,
                                                        double *__restrict__ psi_out, double *__restrict__ lam_out, int in_compact,
                                                        const uint32_t *__restrict__ bdst, const uint32_t *__restrict__ off,
                                                        const uint32_t *__restrict__ rowinfo, int nruns, int nwave,
                                                        const uint32_t *__restrict__ stream, const uint16_t *__restrict__ rowhdr,
                                                        const RotParam *__restrict__ rp, int rot0, int nrot, uint32_t tile_cap,
                                                        double *__restrict__ wpart, int wstride, int sb) {
    constexpr int G = SEC_STREAM_G;
    static_assert(G == 8, "the batch's gradient terms are reduced eight rows at a time");
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    const uint32_t capp = (tile_cap + 2u) & ~1u;
    double *tp = reinterpret_cast<double *>(sec_smem);
    double *tl = tp + capp;
    double2 *cs = reinterpret_cast<double2 *>(tl + capp);
    double *wacc = reinterpret_cast<double *>(cs + nrot);   // [nwave][nrot]
    uint32_t *dst = reinterpret_cast<uint32_t *>(wacc + (size_t)nwave * nrot);
    const uint32_t t = blockIdx.x;
    const uint32_t e0 = off[t];
    const uint32_t n = off[t + 1] - e0;
    if (n == 0) return;
    const uint32_t lane = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t myb = wv < (uint32_t)nwave ? rowinfo[((size_t)t * (uint32_t)nwave + wv) * (uint32_t)(nruns + 1) + min(lane, (uint32_t)nruns)] : 0u;
    const uint32_t R0 = __builtin_amdgcn_readlane(myb, 0), R1 = __builtin_amdgcn_readlane(myb, nruns);
    const int nb = (int)((R1 - R0 + (uint32_t)G - 1u) / (uint32_t)G);   // batches, aligned from the wave's first row
    uint32_t cw[G], nw[G], ch, nh;
    {
        const uint32_t bs = R0 + (uint32_t)(nb ? nb - 1 : 0) * G;
#pragma unroll
        for (int g = 0; g < G; ++g) cw[g] = stream[(size_t)(bs + g) * 64u + lane];
        ch = rowhdr[bs + (lane & (G - 1))];
    }
    const size_t tbase = (size_t)t * tile_cap, ibase = in_compact ? (size_t)e0 : tbase;
    constexpr int TB = 4;
    for (uint32_t k0 = threadIdx.x; k0 < n; k0 += TB * NT) {
        uint32_t d[TB];
        double u[TB], v[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const uint32_t k = min(k0 + (uint32_t)j * NT, n - 1u);
            d[j] = bdst ? bdst[tbase + k] : 0u;
            u[j] = psi_in[ibase + k];
            v[j] = lam_in[ibase + k];
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const uint32_t k = k0 + (uint32_t)j * NT;
            if (k < n) {
                tp[k] = u[j];
                tl[k] = v[j];
                dst[k] = d[j];
            }
        }
    }
    for (int r0 = threadIdx.x; r0 < nrot; r0 += NT) {
        const RotParam ra = rp[rot0 + r0];
        cs[r0] = make_double2(ra.c, ra.s);
    }
    for (int r0 = threadIdx.x; r0 < nwave * nrot; r0 += NT) wacc[r0] = 0.0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    double *mine = wacc + (size_t)min(wv, (uint32_t)nwave - 1u) * nrot;
    const uint32_t mask = (1u << sb) - 1u, pshift = 2u * (uint32_t)sb;
    unsigned char *const pb = reinterpret_cast<unsigned char *>(tp);
    const uint32_t loff = capp * (uint32_t)sizeof(double);   // lambda's tile behind psi's
    int rb = nruns - 1;      // next run boundary below this wave's current row
    uint32_t bnd = __builtin_amdgcn_readlane(myb, max(rb, 0));
    for (int b = nb - 1; b >= 0; --b) {
        const uint32_t bs = R0 + (uint32_t)b * G, ps = b ? bs - G : bs;
#pragma unroll
        for (int g = 0; g < G; ++g) nw[g] = stream[(size_t)(ps + g) * 64u + lane];   // the batch below: in flight while this one is applied
        nh = rowhdr[ps + (lane & (G - 1))];
        uint32_t ai[G], aj[G];
        double qc[G], qn[G], gq[G];
        const bool slow = __ballot((ch & 0xc000u) != 0u) != 0ull;   // a row with words of several entries, or without partner: sums per row
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t w = bs + g < R1 ? cw[g] : 0xffffffffu;
            const uint32_t csb = __builtin_amdgcn_readlane(ch, g) & 0x3fffu;
            const bool live = ((w >> sb) & mask) != mask;      // (an empty lane reads as a word without partner; neither has a term)
            const double2 cr = cs[live ? csb + ((w & 0x7fffffffu) >> pshift) : 0u];
            ai[g] = (w & mask) << 3;
            aj[g] = ((w >> sb) & mask) << 3;
            qc[g] = cr.x;
            qn[g] = __hiloint2double(__double2hiint(cr.y) ^ (int)(w & 0x80000000u), __double2loint(cr.y));
            gq[g] = 0.0;
            asm volatile("" : "+v"(ai[g]), "+v"(aj[g]), "+v"(qc[g]), "+v"(qn[g]));
        }
        auto rows = [&](auto per_row) {
#pragma unroll
            for (int g = G - 1; g >= 0; --g) {
                if (bs + g < R1) {
                    while (rb >= 1 && bs + g < bnd) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        --rb;
                        bnd = __builtin_amdgcn_readlane(myb, max(rb, 0));
                    }
                    const bool live = aj[g] != (mask << 3);
                    double gg = 0.0;
                    if (live) {
                        double *pu = reinterpret_cast<double *>(pb + ai[g]), *pv = reinterpret_cast<double *>(pb + aj[g]);
                        double *lu_ = reinterpret_cast<double *>(pb + loff + ai[g]), *lv_ = reinterpret_cast<double *>(pb + loff + aj[g]);
                        const double u1 = *pu, v1 = *pv, lu = *lu_, lv = *lv_;
                        const double t0 = lu * v1 - lv * u1;
                        gg = __hiloint2double(__double2hiint(t0) ^ (int)(cw[g] & 0x80000000u), __double2loint(t0));
                        *pu = qc[g] * u1 - qn[g] * v1;
                        *pv = qc[g] * v1 + qn[g] * u1;
                        *lu_ = qc[g] * lu - qn[g] * lv;
                        *lv_ = qc[g] * lv + qn[g] * lu;
                    }
                    if (decltype(per_row)::value) {   // the row's terms by table entry
                        const uint32_t ent = (__builtin_amdgcn_readlane(ch, g) & 0x3fffu) + ((cw[g] & 0x7fffffffu) >> pshift);
                        uint64_t todo = __ballot(live);
                        while (todo) {
                            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)ent, __ffsll((long long)todo) - 1);
                            const bool m = live && ent == e;
                            const double tsum = sec_wave_sum63(m ? gg : 0.0);
                            if (lane == 63u) mine[e] += tsum;
                            todo &= ~__ballot(m);
                        }
                    } else {
                        gq[g] = gg;
                    }
                }
            }
        };
        if (slow) {
            rows(std::true_type{});
        } else {
            rows(std::false_type{});
            // eight rows' sums at once; lane l with bit 3 clear holds, for its 16 lanes, the total of row 4 b0 + 2 b1 + b2 (b = bits of l)
            const double tot = sec_reduce8(gq);
            uint32_t ent = 0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t eg = bs + g < R1 ? (__builtin_amdgcn_readlane(ch, g) & 0x3fffu) + ((__builtin_amdgcn_readlane(cw[g], 0) & 0x7fffffffu) >> pshift) : 0u;
                const int L = ((g & 1) << 2) | (g & 2) | ((g & 4) >> 2);   // the lanes (mod 8) that hold row g's total
                ent = (lane & 7u) == (uint32_t)L ? eg : ent;
            }
            if (!(lane & 8u) && tot != 0.0) __hip_atomic_fetch_add(&mine[ent], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) cw[g] = nw[g];
        ch = nh;
    }
    for (; rb >= 1; --rb) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __syncthreads();
    double *wout = wpart + (size_t)t * wstride + rot0;   // [tile][table entry], zeroed by the host
    for (int r = threadIdx.x; r < nrot; r += NT) {
        double tsum = 0.0;
        for (int w = 0; w < nwave; ++w) tsum += wacc[(size_t)w * nrot + r];
        wout[r] = tsum;
    }
    if (bdst) {
        for (uint32_t k = threadIdx.x; k < n; k += NT) {
            const uint32_t d = dst[k];
            psi_out[d] = tp[k];
            lam_out[d] = tl[k];
        }
    }
}
constexpr int SEC_REG_WROWS = 4;   // rows of 16 lanes per wave: each stores its own eight totals (summed over rows and waves at the flush)
__device__ __forceinline__ void sec_reg_wstore(double *__restrict__ wslot, const double (&acc)[8]) {   // this wave's 4 x 8 totals
    const double tot = sec_reduce8(acc);
    const int lane = threadIdx.x & 63;
    if (!(lane & 8)) wslot[(lane >> 4) * 8 + (((lane & 1) << 2) | (lane & 2) | ((lane & 4) >> 2))] = tot;
}

template <int NT, int W, int NSEL>
__device__ __forceinline__ void sec_reg_unapply(double *__restrict__ psi, double *__restrict__ lam, const double2 *__restrict__ T,
                                                const SecRegHead &op, uint32_t nslots, uint32_t tnum, const uint32_t *__restrict__ gwo,
                                                uint32_t wd0, uint32_t wd1, double *__restrict__ wslot) {
    constexpr int NA = 1 << W, NP = NA / 2;
    const uint32_t tz = __popc(tnum & op.zt) & 1u;
    const uint32_t ts0 = NSEL > 0 ? (__popc(tnum & op.sel_t[0]) & 1u) : 0u, ts1 = NSEL > 1 ? (__popc(tnum & op.sel_t[1]) & 1u) : 0u;
    uint32_t dep[NA];
#pragma unroll
    for (int e = 0; e < NA; ++e) dep[e] = (op.dep[e >> 1] >> (16 * (e & 1))) & 0xffffu;
    char *pb = reinterpret_cast<char *>(psi), *lb = reinterpret_cast<char *>(lam);
    double acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0;
    for (uint32_t g = threadIdx.x, it = 0; g < (nslots >> W); g += NT, ++it) {
        const uint32_t wd = it == 0 ? wd0 : (it == 1 ? wd1 : gwo[g]);
        const uint32_t sb = wd & 0xffffu;
        const uint32_t neg = (((wd >> 16) & 1u) ^ tz) << 31;
        double a[NA], l[NA];
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            a[e] = *reinterpret_cast<const double *>(pb + (sb ^ dep[e]));
            l[e] = *reinterpret_cast<const double *>(lb + (sb ^ dep[e]));
        }
        uint32_t sel = 0;
        if constexpr (NSEL > 0) sel = ((wd >> 17) & 1u) ^ ts0;
        if constexpr (NSEL > 1) sel |= (((wd >> 18) & 1u) ^ ts1) << 1;
        const double2 *Tl = T + sel * NP;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double2 r = Tl[q];
            const double u = a[q], v = a[NA - 1 - q], lu = l[q], lv = l[NA - 1 - q], sn = sec_flip(r.y, neg);
            const double cq = sec_flip(lu * v - lv * u, neg);   // (sigma on the second members: folded into s and into the sum)
#pragma unroll
            for (int vs = 0; vs < (1 << NSEL); ++vs) acc[vs * NP + q] += (NSEL == 0 || sel == (uint32_t)vs) ? cq : 0.0;
            a[q] = r.x * u - sn * v;
            a[NA - 1 - q] = r.x * v + sn * u;
            l[q] = r.x * lu - sn * lv;
            l[NA - 1 - q] = r.x * lv + sn * lu;
        }
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            *reinterpret_cast<double *>(pb + (sb ^ dep[e])) = a[e];
            *reinterpret_cast<double *>(lb + (sb ^ dep[e])) = l[e];
        }
    }
    sec_reg_wstore(wslot, acc);
}

// a block of two ops backwards: B first, then A (see sec_reg_apply_pair); wslotA / wslotB = this wave's rows of the two ops
template <int NT, int NSEL>
__device__ __forceinline__ void sec_reg_unapply_pair(double *__restrict__ psi, double *__restrict__ lam, const double2 *__restrict__ TA,
                                                     const double2 *__restrict__ TB, const SecRegHead &op, uint32_t nslots, uint32_t tnum,
                                                     const uint32_t *__restrict__ gwo, uint32_t wd0, uint32_t wd1,
                                                     double *__restrict__ wslotA, double *__restrict__ wslotB) {
    const uint32_t fl = op.pad[0];
    const uint32_t tzA = __popc(tnum & op.zt) & 1u, tzB = __popc(tnum & op.pad[1]) & 1u;
    const uint32_t tsA = NSEL ? (__popc(tnum & op.sel_t[0]) & 1u) : 0u, tsB = NSEL ? (__popc(tnum & op.pad[2]) & 1u) : 0u;
    uint32_t dep[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) dep[e] = (op.dep[e >> 1] >> (16 * (e & 1))) & 0xffffu;
    char *pb = reinterpret_cast<char *>(psi), *lb = reinterpret_cast<char *>(lam);
    double accA[8], accB[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) accA[e] = accB[e] = 0.0;
    for (uint32_t g = threadIdx.x, it = 0; g < (nslots >> 4); g += NT, ++it) {
        const uint32_t wd = it == 0 ? wd0 : (it == 1 ? wd1 : gwo[g]);
        const uint32_t sb = wd & 0xffffu;
        double a[16], l[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            a[e] = *reinterpret_cast<const double *>(pb + (sb ^ dep[e]));
            l[e] = *reinterpret_cast<const double *>(lb + (sb ^ dep[e]));
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {   // op B on the half own A = v
            const uint32_t neg = (((wd >> 18) & 1u) ^ tzB ^ ((uint32_t)v & (fl >> 2))) << 31;
            const uint32_t sel = NSEL ? (((wd >> 19) & 1u) ^ tsB ^ ((uint32_t)v & (fl >> 3))) & 1u : 0u;
            const double2 *Tl = TB + sel * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 r = Tl[q];
                const int i = 4 * v + q, j = 4 * v + 11 - q;
                const double u = a[i], w = a[j], lu = l[i], lw = l[j], sn = sec_flip(r.y, neg);
                const double cq = sec_flip(lu * w - lw * u, neg);
                accB[q] += (NSEL == 0 || sel == 0u) ? cq : 0.0;
                if (NSEL) accB[4 + q] += sel ? cq : 0.0;
                a[i] = r.x * u - sn * w;
                a[j] = r.x * w + sn * u;
                l[i] = r.x * lu - sn * lw;
                l[j] = r.x * lw + sn * lu;
            }
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {   // op A on the half own B = v
            const uint32_t neg = (((wd >> 16) & 1u) ^ tzA ^ ((uint32_t)v & fl)) << 31;
            const uint32_t sel = NSEL ? (((wd >> 17) & 1u) ^ tsA ^ ((uint32_t)v & (fl >> 1))) & 1u : 0u;
            const double2 *Tl = TA + sel * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 r = Tl[q];
                const int i = 8 * v + q, j = 8 * v + 7 - q;
                const double u = a[i], w = a[j], lu = l[i], lw = l[j], sn = sec_flip(r.y, neg);
                const double cq = sec_flip(lu * w - lw * u, neg);
                accA[q] += (NSEL == 0 || sel == 0u) ? cq : 0.0;
                if (NSEL) accA[4 + q] += sel ? cq : 0.0;
                a[i] = r.x * u - sn * w;
                a[j] = r.x * w + sn * u;
                l[i] = r.x * lu - sn * lw;
                l[j] = r.x * lw + sn * lu;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            *reinterpret_cast<double *>(pb + (sb ^ dep[e])) = a[e];
            *reinterpret_cast<double *>(lb + (sb ^ dep[e])) = l[e];
        }
    }
    sec_reg_wstore(wslotA, accA);
    sec_reg_wstore(wslotB, accB);
}

// one backward sweep: the tiles of psi and lambda arrive in this sweep's output order (oslot: slot stored at position j; nullptr:
// slot j), the sweep's ops are undone last to first, the tiles leave for the positions this sweep's forward pass gathered them from
// (srcpad / gslot of k_sector_sweep_reg; nullptr: the first sweep, nothing leaves).  wpart: [tile][nops][8] partial sums.
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_adjoint_reg(const double *__restrict__ psi_in, const double *__restrict__ lam_in,
                                                           double *__restrict__ psi_out, double *__restrict__ lam_out,
                                                           const uint32_t *__restrict__ srcpad, const uint16_t *__restrict__ gslot,
                                                           const uint16_t *__restrict__ oslot, const SecRegOp *__restrict__ ops, int nops,
                                                           const uint32_t *__restrict__ gw, const double2 *__restrict__ tg, int mbits,
                                                           double *__restrict__ wpart, int runcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sec_smem[];
    constexpr int NW = NT / 64;
    const uint32_t nslots = 1u << mbits, t = blockIdx.x;
    double *psi = reinterpret_cast<double *>(sec_smem), *lam = psi + nslots;
    double2 *cs = reinterpret_cast<double2 *>(lam + nslots);                       // [nops][8]
    double *wrow = reinterpret_cast<double *>(cs + (size_t)nops * SEC_REG_TSTRIDE);   // [2 banks][runcap units][2 ops of a block][NW][4 rows of lanes][8]
    const size_t e0 = (size_t)t * nslots;
    for (uint32_t j = threadIdx.x; j < nslots; j += NT) {
        const uint32_t k = sec_reg_swz(oslot ? (uint32_t)oslot[j] : j);
        psi[k] = psi_in[e0 + j];
        lam[k] = lam_in[e0 + j];
    }
    for (uint32_t e = threadIdx.x; e < (uint32_t)nops * SEC_REG_TSTRIDE; e += NT) cs[e] = tg[e];
    __syncthreads();
    // blocks of two ops start at their first op; the list is walked backwards (wave-uniform).  Units of a run (sv_regular_host.hpp:
    // each wave stays inside its own slots) follow each other without a barrier; their rows of partial sums wait in LDS — one bank
    // per run, at most `runcap` units — and are added up over the waves behind the barrier that ends the run.
    const int wave = threadIdx.x >> 6;
    double *wp = wpart + (size_t)t * nops * 8;
    int o = nops - 1, bank = 0, slot = 0;
    while (o >= 0) {
        // the start of the block that ends at op o: op o - 1 carries the pair flag iff (o - 1, o) is a block
        const bool pair = o > 0 && ((reinterpret_cast<const SecRegHead *>(ops + (o - 1))->w_nsel >> 24) & 1u);
        const int ob = pair ? o - 1 : o;
        const SecRegHead cur = *reinterpret_cast<const SecRegHead *>(ops + ob);
        const uint32_t *gwo = gw + (size_t)ob * SEC_REG_GSTRIDE;
        const uint32_t wd0 = gwo[threadIdx.x], wd1 = gwo[threadIdx.x + NT];
        const double2 *T = cs + (size_t)ob * SEC_REG_TSTRIDE;
        double *rowA = wrow + (size_t)((((bank * runcap + slot) * 2 + 0) * NW + wave) * (8 * SEC_REG_WROWS));
        double *rowB = wrow + (size_t)((((bank * runcap + slot) * 2 + 1) * NW + wave) * (8 * SEC_REG_WROWS));
        switch (cur.w_nsel & ~((3u << 25) | (1u << 27))) {
        case 3 | (1 << 24): sec_reg_unapply_pair<NT, 0>(psi, lam, T, T + SEC_REG_TSTRIDE, cur, nslots, t, gwo, wd0, wd1, rowA, rowB); break;
        case 3 | (1 << 16) | (1 << 24): sec_reg_unapply_pair<NT, 1>(psi, lam, T, T + SEC_REG_TSTRIDE, cur, nslots, t, gwo, wd0, wd1, rowA, rowB); break;
        case 4: sec_reg_unapply<NT, 4, 0>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 3: sec_reg_unapply<NT, 3, 0>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 3 | (1 << 16): sec_reg_unapply<NT, 3, 1>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 2: sec_reg_unapply<NT, 2, 0>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 2 | (1 << 16): sec_reg_unapply<NT, 2, 1>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 2 | (2 << 16): sec_reg_unapply<NT, 2, 2>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 1: sec_reg_unapply<NT, 1, 0>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        case 1 | (1 << 16): sec_reg_unapply<NT, 1, 1>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        default: sec_reg_unapply<NT, 1, 2>(psi, lam, T, cur, nslots, t, gwo, wd0, wd1, rowA); break;
        }
        // does the unit before this one (the next to be undone) run on without a barrier?  (its first op carries the flag)
        bool sync = true;
        if (ob > 0 && slot + 1 < runcap) {
            const int pb = (ob > 1 && ((reinterpret_cast<const SecRegHead *>(ops + (ob - 2))->w_nsel >> 24) & 1u)) ? ob - 2 : ob - 1;
            sync = !((reinterpret_cast<const SecRegHead *>(ops + pb)->w_nsel >> 27) & 1u);
        }
        if (sync) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // the run's units, earliest op first: unit u sits in slot (slot - u); thread -> (unit, op of the block, entry)
            const int nu = slot + 1;
            const int mine = (int)(threadIdx.x >> 4);
            int idx = ob, my_op = -1;
            bool my_pair = false;
            for (int u = 0; u < nu; ++u) {
                const bool pr = (reinterpret_cast<const SecRegHead *>(ops + idx)->w_nsel >> 24) & 1u;
                if (u == mine) {
                    my_op = idx;
                    my_pair = pr;
                }
                idx += pr ? 2 : 1;
            }
            const uint32_t which = (threadIdx.x >> 3) & 1u, e = threadIdx.x & 7u;
            if (my_op >= 0 && (which == 0u || my_pair)) {
                const int sl = slot - mine;
                double sum = 0.0;
#pragma unroll
                for (int w = 0; w < NW * SEC_REG_WROWS; ++w)   // (waves x rows of lanes, in a fixed order)
                    sum += wrow[(size_t)(((bank * runcap + sl) * 2 + (int)which) * NW * SEC_REG_WROWS + w) * 8 + e];
                wp[(size_t)(my_op + (int)which) * 8 + e] = sum;
            }
            bank ^= 1;   // (the next run's rows go to the other bank: no second barrier)
            slot = 0;
        } else {
            ++slot;
        }
        o = ob - 1;
    }
    if (srcpad) {
        const uint32_t *sp = srcpad + e0;
        for (uint32_t j = threadIdx.x; j < nslots; j += NT) {
            const uint32_t k = sec_reg_swz(gslot ? (uint32_t)gslot[j] : j);
            const uint32_t d = sp[j];
            psi_out[d] = psi[k];
            lam_out[d] = lam[k];
        }
    }
}
// w[angle-table entry] += sign * (sum over the tiles of wpart[.][e]) for the entries e of one sweep (one block per entry)
__global__ __launch_bounds__(256) void k_sec_reg_wreduce(const double *__restrict__ wpart, uint32_t ntiles, uint32_t nent,
                                                         const uint32_t *__restrict__ emap, double *__restrict__ w) {
    __shared__ double part[4];
    const uint32_t e = blockIdx.x, m = emap[e];
    if (m == 0xffffffffu) return;
    double s = 0.0;
    for (uint32_t t = threadIdx.x; t < ntiles; t += 256u) s += wpart[(size_t)t * nent + e];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = part[0] + part[1] + part[2] + part[3];
        atomicAdd(&w[m & 0x7fffffffu], (m >> 31) ? -tot : tot);
    }
}

// gather table of a sweep whose predecessor stored its tiles in permuted slot order: step j of tile t reads the element that the
// canonical table (srcpad: position in the predecessor's canonical order, per slot) names for slot gslot[j], at the predecessor's
// actual position of it (tile * cap + pos_prev[slot])
__global__ __launch_bounds__(256) void k_sec_reg_src(const uint32_t *__restrict__ srcpad, const uint16_t *__restrict__ gslot,
                                                     const uint16_t *__restrict__ pos_prev, uint32_t cap, uint32_t *__restrict__ rsrc) {
    const size_t e0 = (size_t)blockIdx.x * cap;
    for (uint32_t j = threadIdx.x; j < cap; j += 256u) {
        const uint32_t p = srcpad[e0 + gslot[j]];
        rsrc[e0 + j] = (p & ~(cap - 1u)) | (uint32_t)pos_prev[p & (cap - 1u)];
    }
}

}  // namespace ovqe
