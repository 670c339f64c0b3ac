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
//     that list once (12 bytes per matrix element) against the tile's amplitudes in LDS: HBM-bound, no Pauli arithmetic left.
//
// All tables are built on the device (radix sort of the permuted indices, binary search of partners inside a tile).
#pragma once
#include "sv_kernels.hpp"

namespace ovqe {

constexpr int SEC_SLOT_BITS = 13;
constexpr uint32_t SEC_SLOT_MASK = (1u << SEC_SLOT_BITS) - 1u;
constexpr uint32_t SEC_ORPHAN = SEC_SLOT_MASK;    // "partner outside the support" (slots are 0 .. 8190)
constexpr uint32_t SEC_MAX_TILE = SEC_SLOT_MASK;  // entries per compact tile
constexpr int SEC_MAX_PAT = 32;                   // active patterns of an OP_TAB op (5 bits of the pair word)

// pair word: slot_i | slot_j << 13 | sign << 26 | pattern << 27;  u' = c u + s v, v' = c v - s u, s = sign ? -sin : sin
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
struct SecHSweep {    // one sweep of the materialised <H>: device pointers
    const uint32_t *src;    // [K] position of the entry in the circuit's final order
    const uint32_t *off;    // [ntiles + 1]
    const uint32_t *ebase;  // [K + 1] first matrix element of every entry
    const uint32_t *words;  // slot_i | slot_j << 13
    const double *vals;     // H_ii, or 2 H_ij (pair counted once)
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
// pair whose partner is not in the support is recorded with SEC_ORPHAN: its amplitude is structurally zero at that point of
// the circuit (otherwise the partner would have been populated at the probe parameters) and the sweep checks just that.
template <bool FILL, int NT>
__global__ __launch_bounds__(NT) void k_sec_pairs(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                  const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                  uint32_t smask, const SecBuildOp *__restrict__ ops, int nops,
                                                  const SecPat *__restrict__ pats, uint32_t *__restrict__ cnt,
                                                  const uint32_t *__restrict__ poff, uint32_t *__restrict__ pairs) {
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
                        word = (uint32_t)k | ((sj < 0 ? SEC_ORPHAN : (uint32_t)sj) << SEC_SLOT_BITS) | (sign << 26) | ((uint32_t)p << 27);
                        emit = true;
                        break;
                    }
                    if (bits == (pt.pv ^ (op.x & pt.pm))) {   // second member: only its orphans are recorded
                        if (sec_find(sec_lk, n, sec_lk[k] ^ xl) < 0) {
                            word = (uint32_t)k | (SEC_ORPHAN << SEC_SLOT_BITS) | ((uint32_t)p << 27);
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
constexpr int SEC_WORDS_PER_THREAD = 16;   // pair words per thread and staging buffer
template <int NT>
__device__ __forceinline__ void sec_rotate(double *tile, const double2 *tab, uint32_t pw, bool &bad) {
    const uint32_t si = pw & SEC_SLOT_MASK, sj = (pw >> SEC_SLOT_BITS) & SEC_SLOT_MASK;
    if (sj == SEC_ORPHAN) {
        bad |= tile[si] != 0.0;
        return;
    }
    const double2 r = tab[pw >> 27];
    const double s = (pw & (1u << 26)) ? -r.y : r.y;
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
                                                     uint32_t hf_pos, int *__restrict__ flag) {
    // The pair words of the tile are consecutive in memory (op after op): they are staged in LDS in op-aligned chunks of
    // at most W words, the loads of chunk c + 1 in flight (registers) while chunk c is processed — an op never waits for
    // global memory.  Barriers wait for LDS only (s_waitcnt lgkmcnt), not for those loads.
    constexpr uint32_t W = SEC_WORDS_PER_THREAD * NT;
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
    if (ob > oa) {
        fetch(oa, ob);
        stash(wbuf);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    while (oa < nops) {
        if (ob == oa) {   // one op with more pairs in this tile than a buffer holds: straight from memory
            const uint32_t p0 = lop[oa].p0, p1 = lop[oa + 1].p0;
            const double2 *tab = cs + lop[oa].tab;
            for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) sec_rotate<NT>(tile, tab, pairs[k], bad);
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
            for (uint32_t k = p0 + threadIdx.x; k < p1; k += NT) sec_rotate<NT>(tile, tab, wb[k], bad);
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
// taken from its member with a clear pivot bit.  FILL = false: ecnt[e] = non-zero elements of entry e; FILL = true: they
// are written at ebase[e] (entry-major: a wave of the evaluation kernel reads one amplitude for a run of elements).
template <bool FILL, int NT>
__global__ __launch_bounds__(NT) void k_sec_hbuild(const uint32_t *__restrict__ sup, const uint32_t *__restrict__ keys,
                                                   const uint32_t *__restrict__ cid, const uint32_t *__restrict__ off, int M,
                                                   uint32_t smask, const SecGroup *__restrict__ groups, int ngroups,
                                                   const HTerm *__restrict__ terms, uint32_t *__restrict__ ecnt,
                                                   const uint32_t *__restrict__ ebase, uint32_t *__restrict__ words,
                                                   double *__restrict__ vals) {
    extern __shared__ uint32_t sec_lk[];
    const uint32_t t = blockIdx.x, e0 = off[t];
    const int n = (int)(off[t + 1] - e0);
    if (n == 0) return;
    const uint32_t lmask = (1u << M) - 1u;
    for (int k = threadIdx.x; k < n; k += NT) sec_lk[k] = keys[e0 + k] & lmask;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += NT) {
        const uint64_t i = sup[cid[e0 + k]];
        const uint32_t li = sec_lk[k];
        uint32_t c = 0;
        const uint32_t wb = FILL ? ebase[e0 + k] : 0u;
        for (int g = 0; g < ngroups; ++g) {
            const SecGroup gr = groups[g];
            int sj = k;
            if (gr.x) {
                const uint64_t pbit = 1ull << (63 - __clzll((long long)gr.x));
                if (i & pbit) continue;
                sj = sec_find(sec_lk, n, li ^ sec_pext((uint32_t)gr.x, smask));
                if (sj < 0) continue;
            }
            const uint64_t j = i ^ gr.x;
            double d = 0.0;
            for (int tt = gr.t0; tt < gr.t1; ++tt) {
                const HTerm ht = terms[tt];
                if (__popcll(gr.x & ht.z) & 1) continue;   // odd number of Y: <P> = 0 on a real state
                d += parity64(j & ht.z) ? -ht.cr : ht.cr;
            }
            if (d != 0.0) {
                if (FILL) {
                    words[wb + c] = (uint32_t)k | ((uint32_t)sj << SEC_SLOT_BITS);
                    vals[wb + c] = gr.x ? 2.0 * d : d;
                }
                ++c;
            }
        }
        if (!FILL) ecnt[e0 + k] = c;
    }
}

// E = sum over the sweeps and tiles of sum_e vals[e] a[slot_i] a[slot_j]: blockIdx.y = sweep, blockIdx.x = tile
template <int NT>
__global__ __launch_bounds__(NT) void k_sector_expect(const double *__restrict__ state, const SecHSweep *__restrict__ sweeps,
                                                      double2 *__restrict__ partials) {
    extern __shared__ double sec_tile[];
    __shared__ double2 red[NT / 64];
    const SecHSweep sw = sweeps[blockIdx.y];
    const uint32_t t = blockIdx.x, e0 = sw.off[t];
    const int n = (int)(sw.off[t + 1] - e0);
    const size_t slot = (size_t)blockIdx.y * gridDim.x + t;
    if (n == 0) {
        if (threadIdx.x == 0) partials[slot] = make_double2(0.0, 0.0);
        return;
    }
    for (int k = threadIdx.x; k < n; k += NT) sec_tile[k] = state[sw.src[e0 + k]];
    __syncthreads();
    const uint32_t b0 = sw.ebase[e0], b1 = sw.ebase[e0 + n];
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    uint32_t e = b0 + threadIdx.x;
    for (; e + 3u * NT < b1; e += 4u * NT) {
        const uint32_t w0 = sw.words[e], w1 = sw.words[e + NT], w2 = sw.words[e + 2 * NT], w3 = sw.words[e + 3 * NT];
        const double v0 = sw.vals[e], v1 = sw.vals[e + NT], v2 = sw.vals[e + 2 * NT], v3 = sw.vals[e + 3 * NT];
        acc0 += v0 * sec_tile[w0 & SEC_SLOT_MASK] * sec_tile[w0 >> SEC_SLOT_BITS];
        acc1 += v1 * sec_tile[w1 & SEC_SLOT_MASK] * sec_tile[w1 >> SEC_SLOT_BITS];
        acc2 += v2 * sec_tile[w2 & SEC_SLOT_MASK] * sec_tile[w2 >> SEC_SLOT_BITS];
        acc3 += v3 * sec_tile[w3 & SEC_SLOT_MASK] * sec_tile[w3 >> SEC_SLOT_BITS];
    }
    for (; e < b1; e += NT) {
        const uint32_t w0 = sw.words[e];
        acc0 += sw.vals[e] * sec_tile[w0 & SEC_SLOT_MASK] * sec_tile[w0 >> SEC_SLOT_BITS];
    }
    const double2 tsum = block_sum<NT>(make_double2((acc0 + acc1) + (acc2 + acc3), 0.0), red);
    if (threadIdx.x == 0) partials[slot] = tsum;
}

// the compact state back in canonical (ascending index) order, e.g. for ovqe_get_state-like consumers and tests
__global__ __launch_bounds__(256) void k_sec_scatter(const double *__restrict__ in, const uint32_t *__restrict__ cid, uint32_t K,
                                                     double *__restrict__ out) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e < K) out[cid[e]] = in[e];
}

}  // namespace ovqe
