// sv_cross.hpp — Pauli sums ACROSS two shards of the index-bit-partitioned register (openvqe_amd/distributed.py).
//
// A term whose x mask has a part d on the rank bits pairs amplitude i of this rank's shard with amplitude j = i ^ x_l of the
// shard of rank ^ d ("group terms by x_g ... exchange (read-only) + local partial sums").  The partner's
// shard arrives in chunks of 2^m amplitudes; the kernels below contract one chunk (the KET side) with this rank's resident
// buffer (the BRA of an expectation value, or the sigma = H psi being accumulated):
//
//   k_tile_cross  the tile-cover machinery of sv_tile.hpp with TWO base pointers.  A pass has a set S of M index bits inside the
//                 chunk and a displacement d_out outside S; the workgroup of ket tile t stages that tile in LDS, every thread
//                 holds the amplitudes of the OTHER buffer at tile t ^ d_out in registers (its own outputs), and all x-groups of
//                 the pass — every group whose local x mask is (something inside S) | d_out — are evaluated from the LDS copy:
//                 s_i = sum_g D_g(j) ket_j, j = i ^ x.  DOT: Re sum_i conj(bra_i) s_i into a per-workgroup partial (the
//                 product with the bra once per pass, not once per group); APPLY: out_i += s_i.  A pass moves 32 B per
//                 amplitude whatever the number of its groups, where the streaming kernel k_bilinear re-reads the ket once per
//                 group.  Because the two tiles are fetched separately, x bits OUTSIDE the tile set cost nothing but a pass key:
//                 the x bits above the chunk (which pair ket chunk c with bra chunk c ^ h) are part of d_out.
//   k_cross_small the same contraction for registers too small to tile (chunks below 2^10 amplitudes: the CPU-sized tests):
//                 one thread per output amplitude, groups inside.
#pragma once
#include "sv_tile.hpp"

namespace ovqe {

struct CrossPass {
    uint64_t smask, mask_lo, mask_hi;   // tile bits (inside the chunk), thread bits, trip bits (sv_tile.hpp ExSweep)
    uint64_t d_out;                     // x bits of the pass's groups outside the tile: other tile = ket tile ^ d_out (local index space)
    int32_t a0, a1;                     // chunk range of the pass's apply-form tables (ExChunkT / ExAGroupT / ExTermT)
};

template <int M, int NT, bool NTL, bool DOT>
__global__ __launch_bounds__(NT) void k_tile_cross(const amp_t *__restrict__ ket, amp_t *__restrict__ other, uint64_t ket_gbase,
                                                   uint64_t chunk_off, CrossPass ps, const ExChunkT *__restrict__ chunks,
                                                   const ExAGroupT *__restrict__ groups, const ExTermT *__restrict__ terms,
                                                   double2 *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NEL = 1u << M;
    constexpr int TRIPS = NEL / NT;
    double2 *tile = reinterpret_cast<double2 *>(smem);
    ExTermLds *lt = reinterpret_cast<ExTermLds *>(smem + (size_t)NEL * sizeof(double2));
    ExAGroupT *lg = reinterpret_cast<ExAGroupT *>(lt + TILE_TERM_CAP);
    double2 *red = reinterpret_cast<double2 *>(lg + TILE_APPLY_GROUPS);
    const v2d *p = reinterpret_cast<const v2d *>(ket);
    v2d *q = reinterpret_cast<v2d *>(other);

    uint64_t tb = blockIdx.x;   // the ket tile: the block index spread over the chunk's index bits outside S
    for (uint64_t mk = ps.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, ps.mask_lo);
    const uint64_t gbase = ket_gbase | tb;                 // global index of the ket tile (z bits outside the tile: a sign per tile)
    const uint64_t ob = (chunk_off | tb) ^ ps.d_out;       // the other buffer's tile, local index space of the shard
    double2 acc[TRIPS];
    bool any = false;
    {
        v2d reg[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            const uint64_t hi = spread_bits((uint32_t)j, ps.mask_hi);
            reg[j] = NTL ? __builtin_nontemporal_load(&p[tb | glow | hi]) : p[tb | glow | hi];
            acc[j] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            tile[tile_swz_v(threadIdx.x + j * NT)] = make_double2(reg[j].x, reg[j].y);
            any |= reg[j].x != 0.0 || reg[j].y != 0.0;
        }
    }
    // a ket tile of zeros contributes nothing (a UCC / ADAPT state lives on a particle-number sector: most tiles of the register)
    if (!__syncthreads_or(any)) return;
    for (int ch = ps.a0; ch < ps.a1; ++ch) {
        const ExChunkT ck = chunks[ch];
        __syncthreads();
        for (int t = ck.t0 + (int)threadIdx.x; t < ck.t1; t += NT) {
            const ExTermT et = terms[t];
            const bool neg = parity64(gbase & et.zout);
            ExTermLds l;
            l.cr = neg ? -et.cr : et.cr;
            l.ci = neg ? -et.ci : et.ci;
            l.zin = et.zin;
            l.pad = 0;
            lt[t - ck.t0] = l;
        }
        for (int g = ck.g0 + (int)threadIdx.x; g < ck.g1; g += NT) lg[g - ck.g0] = groups[g];
        __syncthreads();
        for (int g = ck.g0; g < ck.g1; ++g) {
            const ExAGroupT gr = lg[g - ck.g0];
            const uint32_t xl = __builtin_amdgcn_readfirstlane(gr.x);
            const int t0 = __builtin_amdgcn_readfirstlane(gr.t0) - ck.t0, t1 = __builtin_amdgcn_readfirstlane(gr.t1) - ck.t0;
            uint32_t je[TRIPS];
            double2 k[TRIPS];
            double dr[TRIPS], di[TRIPS];
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                je[j] = (threadIdx.x + j * NT) ^ xl;   // the ket's tile-local index: the sign of a term is read off IT
                k[j] = tile[tile_swz_v(je[j])];
                dr[j] = 0.0;
                di[j] = 0.0;
            }
            if (__builtin_amdgcn_readfirstlane(gr.pad) & 1) {   // real folded coefficients only (every group of a real-symmetric H)
                for (int t = t0; t < t1; ++t) {
                    const ExTermLds l = lt[t];
#pragma unroll
                    for (int j = 0; j < TRIPS; ++j) dr[j] = fma(l.cr, parity_sign(je[j] & l.zin), dr[j]);
                }
#pragma unroll
                for (int j = 0; j < TRIPS; ++j) {
                    acc[j].x += dr[j] * k[j].x;
                    acc[j].y += dr[j] * k[j].y;
                }
                continue;
            }
            for (int t = t0; t < t1; ++t) {
                const ExTermLds l = lt[t];
#pragma unroll
                for (int j = 0; j < TRIPS; ++j) {
                    const double sg = parity_sign(je[j] & l.zin);
                    dr[j] = fma(l.cr, sg, dr[j]);
                    di[j] = fma(l.ci, sg, di[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                acc[j].x += dr[j] * k[j].x - di[j] * k[j].y;
                acc[j].y += dr[j] * k[j].y + di[j] * k[j].x;
            }
        }
    }
    if constexpr (DOT) {
        // (the bra tile is fetched HERE, behind the groups: 32 registers less while they run — two workgroups per CU instead of one —
        // and the other workgroup's arithmetic hides the fetch)
        v2d oreg[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) oreg[j] = NTL ? __builtin_nontemporal_load(&q[ob | glow | spread_bits((uint32_t)j, ps.mask_hi)])
                                                      : q[ob | glow | spread_bits((uint32_t)j, ps.mask_hi)];
        double2 part = make_double2(0.0, 0.0);   // conj(bra_i) s_i
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            part.x += oreg[j].x * acc[j].x + oreg[j].y * acc[j].y;
            part.y += oreg[j].x * acc[j].y - oreg[j].y * acc[j].x;
        }
        __syncthreads();
        const double2 t = block_sum<NT>(part, red);
        if (threadIdx.x == 0) {   // launches on one stream are ordered: the slot of this workgroup accumulates over passes, chunks, partners
            const double2 o = partials[blockIdx.x];
            partials[blockIdx.x] = make_double2(o.x + t.x, o.y + t.y);
        }
    } else {
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            const uint64_t g = ob | glow | spread_bits((uint32_t)j, ps.mask_hi);
            const v2d o = q[g];
            v2d r;
            r.x = o.x + acc[j].x;
            r.y = o.y + acc[j].y;
            q[g] = r;
        }
    }
}

// REAL amplitudes (2^n doubles: a basis state under rotations whose strings all carry an odd number of Y — every UCC / ADAPT
// generator): the expectation value on 8-byte amplitudes.  As in k_tile_sweep<REAL> the 16-byte element is a PAIR of amplitudes and
// the pass's masks live in the index space of the pairs (index bit 0 is always inside the tile, d_out never has it); the tile holds
// 2^M doubles (M <= 13: the same 64 KB).  Terms with an imaginary folded coefficient (odd number of Y) vanish between real vectors
// and are left out by the host; partials take the real part only.
template <int M, int NT, bool NTL>
__global__ __launch_bounds__(NT) void k_tile_cross_real(const double *__restrict__ ket, const double *__restrict__ bra, uint64_t ket_gbase,
                                                        uint64_t chunk_off, CrossPass ps, const ExChunkT *__restrict__ chunks,
                                                        const ExAGroupT *__restrict__ groups, const ExTermT *__restrict__ terms,
                                                        double2 *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NEL = 1u << M;          // amplitudes per tile
    constexpr uint32_t NELV = NEL / 2;         // 16-byte elements
    constexpr int TRIPS = NELV / NT;
    double *tile = reinterpret_cast<double *>(smem);
    double2 *tilev = reinterpret_cast<double2 *>(smem);
    ExTermLds *lt = reinterpret_cast<ExTermLds *>(smem + (size_t)NEL * sizeof(double));
    ExAGroupT *lg = reinterpret_cast<ExAGroupT *>(lt + TILE_TERM_CAP);
    double2 *red = reinterpret_cast<double2 *>(lg + TILE_APPLY_GROUPS);
    const v2d *p = reinterpret_cast<const v2d *>(ket);
    const v2d *q = reinterpret_cast<const v2d *>(bra);

    uint64_t tb = blockIdx.x;   // pair-index space
    for (uint64_t mk = ps.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, ps.mask_lo);
    const uint64_t gbase = ket_gbase | (tb << 1);
    const uint64_t ob = ((chunk_off >> 1) | tb) ^ (ps.d_out >> 1);
    double acc0[TRIPS], acc1[TRIPS];
    bool any = false;
    {
        v2d reg[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            reg[j] = NTL ? __builtin_nontemporal_load(&p[tb | glow | spread_bits((uint32_t)j, ps.mask_hi)]) : p[tb | glow | spread_bits((uint32_t)j, ps.mask_hi)];
            acc0[j] = 0.0;
            acc1[j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            tilev[tile_swz_v(threadIdx.x + j * NT)] = make_double2(reg[j].x, reg[j].y);
            any |= reg[j].x != 0.0 || reg[j].y != 0.0;
        }
    }
    if (!__syncthreads_or(any)) return;
    for (int ch = ps.a0; ch < ps.a1; ++ch) {
        const ExChunkT ck = chunks[ch];
        __syncthreads();
        for (int t = ck.t0 + (int)threadIdx.x; t < ck.t1; t += NT) {
            const ExTermT et = terms[t];
            ExTermLds l;
            l.cr = parity64(gbase & et.zout) ? -et.cr : et.cr;
            l.ci = 0.0;
            l.zin = et.zin;
            l.pad = 0;
            lt[t - ck.t0] = l;
        }
        for (int g = ck.g0 + (int)threadIdx.x; g < ck.g1; g += NT) lg[g - ck.g0] = groups[g];
        __syncthreads();
        for (int g = ck.g0; g < ck.g1; ++g) {
            const ExAGroupT gr = lg[g - ck.g0];
            const uint32_t xl = __builtin_amdgcn_readfirstlane(gr.x);
            const int t0 = __builtin_amdgcn_readfirstlane(gr.t0) - ck.t0, t1 = __builtin_amdgcn_readfirstlane(gr.t1) - ck.t0;
            uint32_t je0[TRIPS], je1[TRIPS];
            double k0[TRIPS], k1[TRIPS], d0[TRIPS], d1[TRIPS];
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                const uint32_t e = (threadIdx.x + j * NT) << 1;
                je0[j] = e ^ xl;
                je1[j] = (e | 1u) ^ xl;
                k0[j] = tile[tile_swz<true>(je0[j])];
                k1[j] = tile[tile_swz<true>(je1[j])];
                d0[j] = 0.0;
                d1[j] = 0.0;
            }
            for (int t = t0; t < t1; ++t) {
                const ExTermLds l = lt[t];
#pragma unroll
                for (int j = 0; j < TRIPS; ++j) {
                    d0[j] = fma(l.cr, parity_sign(je0[j] & l.zin), d0[j]);
                    d1[j] = fma(l.cr, parity_sign(je1[j] & l.zin), d1[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < TRIPS; ++j) {
                acc0[j] = fma(d0[j], k0[j], acc0[j]);
                acc1[j] = fma(d1[j], k1[j], acc1[j]);
            }
        }
    }
    v2d breg[TRIPS];     // (the bra tile behind the groups: fewer registers while they run)
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) breg[j] = NTL ? __builtin_nontemporal_load(&q[ob | glow | spread_bits((uint32_t)j, ps.mask_hi)])
                                                  : q[ob | glow | spread_bits((uint32_t)j, ps.mask_hi)];
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) part += breg[j].x * acc0[j] + breg[j].y * acc1[j];
    __syncthreads();
    const double2 t = block_sum<NT>(make_double2(part, 0.0), red);
    if (threadIdx.x == 0) {
        const double2 o = partials[blockIdx.x];
        partials[blockIdx.x] = make_double2(o.x + t.x, o.y);
    }
}

// Registers below the tile sizes.  Groups [g0, g1) share the part of their x mask above the chunk bits (the host launches one
// class at a time), so output amplitude i of the class's output chunk `other` takes ket_{i ^ x_low} of the received chunk from
// every group; a thread owns its outputs.  HGroup::x = the x mask on the chunk bits, HTerm::z = the full z mask (the sign is read
// off the ket's GLOBAL index ket_gbase | j).
__global__ __launch_bounds__(256) void k_cross_small_real(const double *__restrict__ ket, const double *__restrict__ bra, uint64_t csize,
                                                          uint64_t ket_gbase, const HGroup *__restrict__ groups, int g0, int g1,
                                                          const HTerm *__restrict__ terms, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double part = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < csize; i += (uint64_t)gridDim.x * 256u) {
        double s = 0.0;
        for (int g = g0; g < g1; ++g) {
            const HGroup gr = groups[g];
            const uint64_t j = i ^ gr.x;
            const uint64_t gj = ket_gbase | j;
            double dr = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                dr = fma(ht.cr, parity_sign64(gj & ht.z), dr);
            }
            s = fma(dr, ket[j], s);
        }
        part = fma(bra[i], s, part);
    }
    const double2 t = block_sum<256>(make_double2(part, 0.0), red);
    if (threadIdx.x == 0) {
        const double2 o = partials[blockIdx.x];
        partials[blockIdx.x] = make_double2(o.x + t.x, o.y);
    }
}

template <bool DOT>
__global__ __launch_bounds__(256) void k_cross_small(const amp_t *__restrict__ ket, amp_t *__restrict__ other, uint64_t csize,
                                                     uint64_t ket_gbase, const HGroup *__restrict__ groups, int g0, int g1,
                                                     const HTerm *__restrict__ terms, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double2 part = make_double2(0.0, 0.0);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < csize; i += (uint64_t)gridDim.x * 256u) {
        double sx = 0.0, sy = 0.0;
        for (int g = g0; g < g1; ++g) {
            const HGroup gr = groups[g];
            const uint64_t j = i ^ gr.x;
            const amp_t k = ket[j];
            const uint64_t gj = ket_gbase | j;
            double dr = 0.0, di = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                const double sg = parity_sign64(gj & ht.z);
                dr = fma(ht.cr, sg, dr);
                di = fma(ht.ci, sg, di);
            }
            sx += dr * k.x - di * k.y;
            sy += dr * k.y + di * k.x;
        }
        if constexpr (DOT) {
            const amp_t b = other[i];
            part.x += b.x * sx + b.y * sy;
            part.y += b.x * sy - b.y * sx;
        } else {
            amp_t o = other[i];
            o.x += sx;
            o.y += sy;
            other[i] = o;
        }
    }
    if constexpr (DOT) {
        const double2 t = block_sum<256>(part, red);
        if (threadIdx.x == 0) {
            const double2 o = partials[blockIdx.x];
            partials[blockIdx.x] = make_double2(o.x + t.x, o.y + t.y);
        }
    }
}

}  // namespace ovqe
