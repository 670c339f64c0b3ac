// sv_kernels.hpp — gfx950 device code of the statevector backend (HBM-streaming path).
//
// Every sweep is bandwidth-bound (≈0.25–0.5 flop/B), so the design rules are: each amplitude is read
// once and written once per sweep with 16-byte accesses, consecutive lanes touch consecutive
// amplitudes (or an XOR-permutation inside the same aligned 64-B segments), several independent
// 16-B loads are in flight per lane, and no MFMA/LDS tiling is attempted — there is no reuse to
// capture.  Reductions are wave-shuffle + LDS trees with a fixed order (deterministic, so ADAPT
// rankings are reproducible).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ovqe {

typedef double2 amp_t;  // .x = re, .y = im

// one Pauli rotation of a fused same-x run, angles already resolved
struct RotParam {
    uint64_t z;   // z mask (global index space)
    double c;     // cos(phi)
    double s;     // sin(phi) * (ny&2 ? -1 : +1)
    int32_t odd;  // ny & 1  (ny = popcount(x&z)): 1 -> real mixing, 0 -> multiply partner by -i
    int32_t pad;
};

// Hamiltonian / pool term with i^{ny} folded into the coefficient
struct HTerm {
    uint64_t z;
    double cr, ci;
};

struct HGroup {
    uint64_t x;      // local part of the x mask
    uint64_t jbase;  // high (global) bits of the partner's global index
    int32_t t0, t1;  // term range
    double tiny;     // 64 eps sum_t |c_t|: a D_g(j) at or below it is a rounding residue of a sum that cancels (see group_coeff_snap)
};
// D_g(j) = sum_t +-c_t of an operator application.  The strings of a number-conserving operator cancel EXACTLY on the basis states
// they must not connect, but their coefficients come out of the caller's algebra equal to an ulp, not bit for bit: the residue
// (1e-17 c) would plant amplitudes outside the particle-number sector, which stay harmless in value (1e-17, 1e-34, ...) and
// ruinous in cost — the support of an ADAPT state is what the screens and the exponentials walk (24 qubits, 30 operators:
// 3.8 M "non-zero" amplitudes, 0.6 M of them above 1e-14).  A sum at or below 64 eps sum |c_t| is that residue: zero.
__device__ __forceinline__ void group_coeff_snap(double &dr, double &di, double tiny) {
    dr = fabs(dr) <= tiny ? 0.0 : dr;
    di = fabs(di) <= tiny ? 0.0 : di;
}

__device__ __forceinline__ uint64_t insert_zero(uint64_t k, int p) {
    const uint64_t low = (1ull << p) - 1ull;
    return ((k & ~low) << 1) | (k & low);
}
__device__ __forceinline__ int parity64(uint64_t v) { return __popcll(v) & 1; }
// (-1)^popcount as a double, from the parity bit by integer arithmetic (bit 0 of the count shifted into the sign of 1.0):
// accumulating fma(c, sign, acc) is the same exactly-rounded sum as acc += parity ? -c : c at half the VALU instructions
// (no compare / 64-bit select)
__device__ __forceinline__ double parity_sign(uint32_t v) {
    return __hiloint2double((int)(((uint32_t)__popc(v) << 31) | 0x3FF00000u), 0);
}
__device__ __forceinline__ double parity_sign64(uint64_t v) {
    return __hiloint2double((int)(((uint32_t)__popcll(v) << 31) | 0x3FF00000u), 0);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
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

// block-wide deterministic sum of a complex value; result valid in thread 0
template <int NT>
__device__ __forceinline__ double2 block_sum(double2 v, double2 *lds /* NT/64 entries */) {
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double2 t = make_double2(0.0, 0.0);
    if (threadIdx.x == 0) {
        for (int i = 0; i < NT / 64; ++i) {
            t.x += lds[i].x;
            t.y += lds[i].y;
        }
    }
    __syncthreads();
    return t;
}

// apply one rotation of a fused run to the pair (u = a_i, v = a_j), j = i ^ x
__device__ __forceinline__ void rot_pair(amp_t &u, amp_t &v, const RotParam &r, uint64_t gi) {
    const int pi = parity64(gi & r.z);  // sign for <j|P|i>
    const int pj = pi ^ r.odd;          // parity(gj & z) = pi ^ parity(x & z)
    const double si = pj ? -r.s : r.s;
    const double sj = pi ? -r.s : r.s;
    amp_t nu, nv;
    if (r.odd) {
        nu.x = r.c * u.x + si * v.x;
        nu.y = r.c * u.y + si * v.y;
        nv.x = r.c * v.x + sj * u.x;
        nv.y = r.c * v.y + sj * u.y;
    } else {
        nu.x = r.c * u.x + si * v.y;
        nu.y = r.c * u.y - si * v.x;
        nv.x = r.c * v.x + sj * u.y;
        nv.y = r.c * v.y - sj * u.x;
    }
    u = nu;
    v = nv;
}

// ------------------------------------------------------------------------------------------------
// exp(-i phi_r P_r) for a run of rotations sharing the x mask (x != 0): one thread per pair (i, i^x),
// i has the pivot bit (highest x bit) clear.  In place; each amplitude is read once, written once.
template <int U>
__global__ __launch_bounds__(256) void k_rot_pairs(amp_t *__restrict__ st, uint64_t npairs, int pivot, uint64_t x,
                                                   uint64_t base, const RotParam *__restrict__ rp, int nrot) {
    const uint64_t k0 = (uint64_t)blockIdx.x * (256u * U) + threadIdx.x;
    amp_t u[U], v[U];
    uint64_t ii[U];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t k = k0 + (uint64_t)m * 256u;
        ii[m] = insert_zero(k, pivot);
        if (k < npairs) {
            u[m] = st[ii[m]];
            v[m] = st[ii[m] ^ x];
        }
    }
    for (int r = 0; r < nrot; ++r) {
        const RotParam rr = rp[r];
#pragma unroll
        for (int m = 0; m < U; ++m) rot_pair(u[m], v[m], rr, base | ii[m]);
    }
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t k = k0 + (uint64_t)m * 256u;
        if (k < npairs) {
            st[ii[m]] = u[m];
            st[ii[m] ^ x] = v[m];
        }
    }
}

// tuning variants of the pair sweep (selected with ovqe_set_option("rot_variant", v)):
//   NT threads per block, U pairs per thread per trip, NTL non-temporal loads/stores, PERSIST grid-stride loop
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NT, int U, bool NTL, bool PERSIST>
__global__ __launch_bounds__(NT) void k_rot_pairs_v(amp_t *__restrict__ st, uint64_t npairs, int pivot, uint64_t x,
                                                    uint64_t base, const RotParam *__restrict__ rp, int nrot) {
    v2d *p = reinterpret_cast<v2d *>(st);
    const uint64_t tile = (uint64_t)NT * U;
    const uint64_t ntiles = (npairs + tile - 1) / tile;
    for (uint64_t t = blockIdx.x; t < ntiles; t += PERSIST ? gridDim.x : ntiles) {
        const uint64_t k0 = t * tile + threadIdx.x;
        amp_t u[U], v[U];
        uint64_t ii[U];
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint64_t k = k0 + (uint64_t)m * NT;
            ii[m] = insert_zero(k, pivot);
        }
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint64_t k = k0 + (uint64_t)m * NT;
            if (k < npairs) {
                v2d a, b;
                if (NTL) {
                    a = __builtin_nontemporal_load(&p[ii[m]]);
                    b = __builtin_nontemporal_load(&p[ii[m] ^ x]);
                } else {
                    a = p[ii[m]];
                    b = p[ii[m] ^ x];
                }
                u[m] = make_double2(a.x, a.y);
                v[m] = make_double2(b.x, b.y);
            }
        }
        for (int r = 0; r < nrot; ++r) {
            const RotParam rr = rp[r];
#pragma unroll
            for (int m = 0; m < U; ++m) rot_pair(u[m], v[m], rr, base | ii[m]);
        }
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint64_t k = k0 + (uint64_t)m * NT;
            if (k < npairs) {
                v2d a = {u[m].x, u[m].y}, b = {v[m].x, v[m].y};
                if (NTL) {
                    __builtin_nontemporal_store(a, &p[ii[m]]);
                    __builtin_nontemporal_store(b, &p[ii[m] ^ x]);
                } else {
                    p[ii[m]] = a;
                    p[ii[m] ^ x] = b;
                }
            }
        }
    }
}

// diagonal run (x == 0): a_i <- prod_r (c_r - i s_r (-1)^{parity(i & z_r)}) a_i
template <int U>
__global__ __launch_bounds__(256) void k_rot_diag(amp_t *__restrict__ st, uint64_t namps, uint64_t base,
                                                  const RotParam *__restrict__ rp, int nrot) {
    const uint64_t i0 = (uint64_t)blockIdx.x * (256u * U) + threadIdx.x;
    amp_t a[U];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t i = i0 + (uint64_t)m * 256u;
        if (i < namps) a[m] = st[i];
    }
    for (int r = 0; r < nrot; ++r) {
        const RotParam rr = rp[r];
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint64_t i = i0 + (uint64_t)m * 256u;
            const double s = parity64((base | i) & rr.z) ? -rr.s : rr.s;
            amp_t t;
            t.x = rr.c * a[m].x + s * a[m].y;
            t.y = rr.c * a[m].y - s * a[m].x;
            a[m] = t;
        }
    }
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t i = i0 + (uint64_t)m * 256u;
        if (i < namps) st[i] = a[m];
    }
}

template <int NT, int U, bool NTL>
__global__ __launch_bounds__(NT) void k_rot_diag_v(amp_t *__restrict__ st, uint64_t namps, uint64_t base,
                                                   const RotParam *__restrict__ rp, int nrot) {
    v2d *p = reinterpret_cast<v2d *>(st);
    const uint64_t i0 = (uint64_t)blockIdx.x * ((uint64_t)NT * U) + threadIdx.x;
    amp_t a[U];
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t i = i0 + (uint64_t)m * NT;
        if (i < namps) {
            const v2d t = NTL ? __builtin_nontemporal_load(&p[i]) : p[i];
            a[m] = make_double2(t.x, t.y);
        }
    }
    for (int r = 0; r < nrot; ++r) {
        const RotParam rr = rp[r];
#pragma unroll
        for (int m = 0; m < U; ++m) {
            const uint64_t i = i0 + (uint64_t)m * NT;
            const double s = parity64((base | i) & rr.z) ? -rr.s : rr.s;
            a[m] = make_double2(rr.c * a[m].x + s * a[m].y, rr.c * a[m].y - s * a[m].x);
        }
    }
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t i = i0 + (uint64_t)m * NT;
        if (i < namps) {
            const v2d t = {a[m].x, a[m].y};
            if (NTL) __builtin_nontemporal_store(t, &p[i]); else p[i] = t;
        }
    }
}

// literal non-rotation gates: kind 0 = X(bit b0) swap, 1 = H(bit b0), 2 = CNOT(control b0, target b1)
template <int U>
__global__ __launch_bounds__(256) void k_gate(amp_t *__restrict__ st, uint64_t nwork, int kind, int b0, int b1) {
    const uint64_t k0 = (uint64_t)blockIdx.x * (256u * U) + threadIdx.x;
#pragma unroll
    for (int m = 0; m < U; ++m) {
        const uint64_t k = k0 + (uint64_t)m * 256u;
        if (k >= nwork) continue;
        uint64_t i, j;
        if (kind == 2) {
            const int lo = b0 < b1 ? b0 : b1, hi = b0 < b1 ? b1 : b0;
            i = insert_zero(insert_zero(k, lo), hi) | (1ull << b0);
            j = i | (1ull << b1);
        } else {
            i = insert_zero(k, b0);
            j = i | (1ull << b0);
        }
        amp_t a = st[i], b = st[j];
        if (kind == 1) {
            const double r = 0.70710678118654752440;
            amp_t s, d;
            s.x = (a.x + b.x) * r;
            s.y = (a.y + b.y) * r;
            d.x = (a.x - b.x) * r;
            d.y = (a.y - b.y) * r;
            st[i] = s;
            st[j] = d;
        } else {
            st[i] = b;
            st[j] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sum_g sum_i conj(bra_i) D_g(i) ket_{i^x_g},  D_g(i) = sum_{t in g} (-1)^{parity(gj & z_t)} (cr_t + i ci_t),
// gj = jbase_g | (i ^ x_g) the partner's global index.  Grid-stride over amplitudes, loop over the
// groups [g0,g1) inside; one complex partial per block (fixed reduction order).
__global__ __launch_bounds__(256) void k_bilinear(const amp_t *__restrict__ bra, const amp_t *__restrict__ ket,
                                                  uint64_t namps, const HGroup *__restrict__ groups, int g0, int g1,
                                                  const HTerm *__restrict__ terms, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    // amplitudes outside, groups inside: bra_i is read once for all the groups of the launch and the ket reads of a wave stay inside
    // the 64-amplitude neighbourhoods i ^ x_g — (16 + 16 G) instead of 32 G bytes per amplitude for G groups (the remote contraction of a
    // partner chunk, distributed.py, runs tens of groups per launch)
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const amp_t b = bra[i];
        for (int g = g0; g < g1; ++g) {
            const HGroup gr = groups[g];
            const uint64_t jl = i ^ gr.x;
            const amp_t k = ket[jl];
            const uint64_t gj = gr.jbase | jl;
            double dr = 0.0, di = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                const double sg = parity_sign64(gj & ht.z);
                dr = fma(ht.cr, sg, dr);
                di = fma(ht.ci, sg, di);
            }
            // v = conj(b) * k
            const double vx = b.x * k.x + b.y * k.y;
            const double vy = b.x * k.y - b.y * k.x;
            acc.x += dr * vx - di * vy;
            acc.y += dr * vy + di * vx;
        }
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// <psi|H|psi> contribution of the groups [g0,g1) for a Hermitian sum (real coefficients before the i^ny fold):
// the (i,j) and (j,i) terms of a group are complex conjugates, so each pair is visited once:
//   E_g = 2 Re sum_{k} D(i) conj(a_i) a_j,  i = insert_zero(k, pivot), j = i ^ x   (x = 0: sum_i D(i) |a_i|^2)
// -> every amplitude is read once per group (16 B) instead of twice.
__global__ __launch_bounds__(256) void k_expect_pairs(const amp_t *__restrict__ st, uint64_t namps,
                                                      const HGroup *__restrict__ groups, int g0, int g1,
                                                      const HTerm *__restrict__ terms, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (int g = g0; g < g1; ++g) {
        const HGroup gr = groups[g];
        if (gr.x == 0) {
            for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
                const amp_t a = st[i];
                const uint64_t gi = gr.jbase | i;
                double d = 0.0;
                for (int t = gr.t0; t < gr.t1; ++t) {
                    const HTerm ht = terms[t];
                    d += parity64(gi & ht.z) ? -ht.cr : ht.cr;
                }
                acc += d * (a.x * a.x + a.y * a.y);
            }
        } else {
            const int pivot = 63 - __clzll(gr.x);
            for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < (namps >> 1); k += stride) {
                const uint64_t i = insert_zero(k, pivot), j = i ^ gr.x;
                const amp_t a = st[i], c = st[j];
                const uint64_t gj = gr.jbase | j;
                double dr = 0.0, di = 0.0;
                for (int t = gr.t0; t < gr.t1; ++t) {
                    const HTerm ht = terms[t];
                    const double sg = parity_sign64(gj & ht.z);
                    dr = fma(ht.cr, sg, dr);
                    di = fma(ht.ci, sg, di);
                }
                const double vx = a.x * c.x + a.y * c.y;  // conj(a_i) a_j
                const double vy = a.x * c.y - a.y * c.x;
                acc += 2.0 * (dr * vx - di * vy);
            }
        }
    }
    double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// out[slot] = sum of `count` complex partials (single block, fixed order)
__global__ __launch_bounds__(256) void k_reduce(const double2 *__restrict__ partials, int64_t count,
                                                double2 *__restrict__ out, int slot) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    for (int64_t i = threadIdx.x; i < count; i += 256) {
        acc.x += partials[i].x;
        acc.y += partials[i].y;
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) out[slot] = t;
}

// out_i (+)= scale * sum_g D_g(i) in_{i ^ x_g}   (sigma = H psi, Taylor steps of exp(theta A))
// mode 0: out = val ; mode 1: out = val and acc += val
__global__ __launch_bounds__(256) void k_apply_sum(amp_t *__restrict__ out, const amp_t *__restrict__ in,
                                                   amp_t *__restrict__ acc, uint64_t namps, uint64_t base,
                                                   const HGroup *__restrict__ groups, int ngroups,
                                                   const HTerm *__restrict__ terms, double scale_re, double scale_im,
                                                   double ident_re, double ident_im) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const amp_t self = in[i];
        double sx = ident_re * self.x - ident_im * self.y;
        double sy = ident_re * self.y + ident_im * self.x;
        for (int g = 0; g < ngroups; ++g) {
            const HGroup gr = groups[g];
            const uint64_t jl = i ^ gr.x;
            const amp_t k = in[jl];
            const uint64_t gj = base | jl;
            double dr = 0.0, di = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                const double sg = parity_sign64(gj & ht.z);
                dr = fma(ht.cr, sg, dr);
                di = fma(ht.ci, sg, di);
            }
            group_coeff_snap(dr, di, gr.tiny);
            sx += dr * k.x - di * k.y;
            sy += dr * k.y + di * k.x;
        }
        amp_t r;
        r.x = scale_re * sx - scale_im * sy;
        r.y = scale_re * sy + scale_im * sx;
        out[i] = r;
        if (acc) {
            amp_t a = acc[i];
            a.x += r.x;
            a.y += r.y;
            acc[i] = a;
        }
    }
}

// The same step for the listed output indices only (Taylor steps of exp(theta A) on a state of a few determinants: every
// index the series can reach is in the list, k_support_expand; the buffers hold zeros elsewhere, or `listed` says where they count).  Same arithmetic, same
// order: the listed amplitudes equal k_apply_sum's bit for bit.
__global__ __launch_bounds__(256) void k_apply_sum_list(amp_t *__restrict__ out, const amp_t *__restrict__ in,
                                                        amp_t *__restrict__ acc, const uint64_t *__restrict__ idx,
                                                        uint64_t count, uint64_t base, const HGroup *__restrict__ groups,
                                                        int ngroups, const HTerm *__restrict__ terms, double scale_re,
                                                        double scale_im, const uint32_t *__restrict__ listed) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= count) return;
    const uint64_t i = idx[e];
    double sx = 0.0, sy = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        const HGroup gr = groups[g];
        const uint64_t jl = i ^ gr.x;
        // `listed` (bitmap of the list): `in` is defined on the list only — a ket outside it holds an exact zero of the series
        amp_t k = make_double2(0.0, 0.0);
        if (!listed || ((listed[jl >> 5] >> (jl & 31)) & 1u)) k = in[jl];
        const uint64_t gj = base | jl;
        double dr = 0.0, di = 0.0;
        for (int t = gr.t0; t < gr.t1; ++t) {
            const HTerm ht = terms[t];
            const double sg = parity_sign64(gj & ht.z);
            dr = fma(ht.cr, sg, dr);
            di = fma(ht.ci, sg, di);
        }
        group_coeff_snap(dr, di, gr.tiny);
        sx += dr * k.x - di * k.y;
        sy += dr * k.y + di * k.x;
    }
    amp_t r;
    r.x = scale_re * sx - scale_im * sy;
    r.y = scale_re * sy + scale_im * sx;
    out[i] = r;
    if (acc) {
        amp_t a = acc[i];
        a.x += r.x;
        a.y += r.y;
        acc[i] = a;
    }
}

// dst[i] = src[i] for the listed indices i
__global__ __launch_bounds__(256) void k_list_copy(amp_t *__restrict__ dst, const amp_t *__restrict__ src, const uint64_t *__restrict__ idx,
                                                   uint64_t count) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e < count) dst[idx[e]] = src[idx[e]];
}

// bitmap of the listed indices
__global__ __launch_bounds__(256) void k_support_mark(const uint64_t *__restrict__ idx, uint64_t count, uint32_t *__restrict__ bitmap) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e < count) atomicOr(&bitmap[idx[e] >> 5], 1u << (idx[e] & 31));
}
// one round of the closure of the list under the operator's x-groups: ket j = idx[e], e in [first, last), reaches j ^ x_g when
// D_g(j) != 0; indices not yet in the bitmap are appended behind position *total (capacity cap: *total may run past it, the
// host then falls back to the pass over the register)
__global__ __launch_bounds__(256) void k_support_expand(uint64_t *__restrict__ idx, uint64_t first, uint64_t last, uint64_t cap,
                                                        uint64_t base, const HGroup *__restrict__ groups, int ngroups,
                                                        const HTerm *__restrict__ terms, uint32_t *__restrict__ bitmap,
                                                        unsigned long long *__restrict__ total) {
    const uint64_t e = first + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= last) return;
    const uint64_t jl = idx[e], gj = base | jl;
    for (int g = 0; g < ngroups; ++g) {
        const HGroup gr = groups[g];
        double dr = 0.0, di = 0.0;
        for (int t = gr.t0; t < gr.t1; ++t) {
            const HTerm ht = terms[t];
            const double sg = parity_sign64(gj & ht.z);
            dr = fma(ht.cr, sg, dr);
            di = fma(ht.ci, sg, di);
        }
        group_coeff_snap(dr, di, gr.tiny);
        if (dr == 0.0 && di == 0.0) continue;
        const uint64_t i = jl ^ gr.x;
        const uint32_t bit = 1u << (i & 31);
        if (atomicOr(&bitmap[i >> 5], bit) & bit) continue;
        const unsigned long long pos = atomicAdd(total, 1ull);
        if (pos < cap) idx[pos] = i;
    }
}

// out_i (+)= sum_g D_g(jbase_g | (i ^ x_g)) in_{i ^ x_g}: a Pauli sum applied to an explicit ket buffer that may be ANOTHER
// shard of a distributed register (jbase = the global index bits of that shard); accumulate = 0 overwrites out
__global__ __launch_bounds__(256) void k_apply_terms(amp_t *__restrict__ out, const amp_t *__restrict__ in, uint64_t namps,
                                                     const HGroup *__restrict__ groups, int ngroups,
                                                     const HTerm *__restrict__ terms, int accumulate) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        double sx = 0.0, sy = 0.0;
        for (int g = 0; g < ngroups; ++g) {
            const HGroup gr = groups[g];
            const uint64_t jl = i ^ gr.x;
            const amp_t k = in[jl];
            const uint64_t gj = gr.jbase | jl;
            double dr = 0.0, di = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                const double sg = parity_sign64(gj & ht.z);
                dr = fma(ht.cr, sg, dr);
                di = fma(ht.ci, sg, di);
            }
            group_coeff_snap(dr, di, gr.tiny);
            sx += dr * k.x - di * k.y;
            sy += dr * k.y + di * k.x;
        }
        amp_t r = accumulate ? out[i] : make_double2(0.0, 0.0);
        r.x += sx;
        r.y += sy;
        out[i] = r;
    }
}

// pool gradient screen: block (chunk c, operator k) sums its slice [c, c+1) * namps / gridDim.x of
// val_k = sum_{t in op k} sum_i conj(sig_i) (-1)^{parity((i^x_t)&z_t)} (cr_t + i ci_t) psi_{i^x_t}
// into partials[k * gridDim.x + c].  One chunk per operator while the state re-streams from L2/MALL (n <= 22: the
// partial IS the value); larger registers are cut into chunks so that operators x chunks workgroups fill the chip in
// ONE launch, and k_reduce_rows2 adds the chunks of an operator in a fixed order (deterministic ranking).
__global__ __launch_bounds__(256) void k_pool_grad(const amp_t *__restrict__ sig, const amp_t *__restrict__ psi,
                                                   uint64_t namps, uint64_t base, const int64_t *__restrict__ offsets,
                                                   const uint64_t *__restrict__ xs, const HTerm *__restrict__ terms,
                                                   int64_t op0, double2 *__restrict__ partials) {
    // xs carries the LOCAL part of every x mask; `base` is the global index base of the KET shard (the own shard's, or
    // the partner's when the operators' x masks share one global part — ovqe_bilinear_batch)
    __shared__ double2 red[4];
    const int64_t op = op0 + blockIdx.y;
    const uint64_t len = namps / gridDim.x, i0 = (uint64_t)blockIdx.x * len, i1 = i0 + len;
    double2 acc = make_double2(0.0, 0.0);
    const int64_t t1 = offsets[op + 1];
    for (int64_t t = offsets[op]; t < t1;) {
        // consecutive strings with one x mask (the 2 / 8 JW strings of a fermionic excitation) share the pass over the
        // pairs: D(j) = sum_t (+-)(cr + i ci), one read of sigma_i and psi_j for the whole run
        const uint64_t x = xs[t];
        int64_t te = t + 1;
        while (te < t1 && xs[te] == x) ++te;
        for (uint64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const uint64_t jl = i ^ x;
            const amp_t b = sig[i], k = psi[jl];
            double cr = 0.0, ci = 0.0;
            for (int64_t u = t; u < te; ++u) {
                const HTerm ht = terms[u];
                const double sg = parity_sign64((base | jl) & ht.z);
                cr = fma(ht.cr, sg, cr);
                ci = fma(ht.ci, sg, ci);
            }
            const double vx = b.x * k.x + b.y * k.y;
            const double vy = b.x * k.y - b.y * k.x;
            acc.x += cr * vx - ci * vy;
            acc.y += cr * vy + ci * vx;
        }
        t = te;
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[(size_t)(op - op0) * gridDim.x + blockIdx.x] = t;
}

// ---- the same screen over the SUPPORT of psi -------------------------------------------------------------------------
// An ADAPT state of a few operators occupies a few determinants of the register (at most the particle-number / spin sector
// of the reference determinant): val_k only has contributions from kets j with psi_j != 0.  The support is listed once per
// screen (ascending indices, so the summation order is fixed) and every operator walks the list instead of the register.
constexpr int NZ_PER_THREAD = 8;
__global__ __launch_bounds__(256) void k_nz_count(const amp_t *__restrict__ st, uint64_t namps, uint32_t *__restrict__ counts) {
    __shared__ uint32_t w[4];
    // the block's 256 * NZ_PER_THREAD amplitudes, consecutive lanes on consecutive amplitudes (only the count matters here)
    const uint64_t i0 = (uint64_t)blockIdx.x * 256u * NZ_PER_THREAD + threadIdx.x;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < NZ_PER_THREAD; ++k)
        if (i0 + 256u * k < namps) {
            const amp_t a = st[i0 + 256u * k];
            if (a.x != 0.0 || a.y != 0.0) ++c;
        }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
// ascending indices and amplitudes of the non-zero entries; start[b] = exclusive sum of counts
__global__ __launch_bounds__(256) void k_nz_fill(const amp_t *__restrict__ st, uint64_t namps, const uint64_t *__restrict__ start,
                                                 uint64_t *__restrict__ idx, amp_t *__restrict__ val) {
    __shared__ uint32_t w[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * NZ_PER_THREAD;
    amp_t a[NZ_PER_THREAD];
    uint32_t c = 0;
    for (int k = 0; k < NZ_PER_THREAD; ++k) {
        a[k] = i0 + k < namps ? st[i0 + k] : make_double2(0.0, 0.0);
        if (a[k].x != 0.0 || a[k].y != 0.0) ++c;
    }
    uint32_t incl = c;   // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t pos = start[blockIdx.x] + incl - c;
    for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) pos += w[k];
    for (int k = 0; k < NZ_PER_THREAD; ++k)
        if (a[k].x != 0.0 || a[k].y != 0.0) {
            idx[pos] = i0 + k;
            val[pos++] = a[k];
        }
}

// Pattern table of a same-x run of pool terms (k_pool_grad_nz): the z masks of the strings of one excitation differ on a few
// bits V only (the Y positions among the excitation's orbitals), so D(j) = sum_t c_t (-1)^{|j & z_t|} = (-1)^{|j & z_0|} tab[j on V]:
// 2^|V| <= 16 sums made once on the host (same terms, same order, fma: the same doubles the term loop produces) instead of a
// loop over the run's terms per amplitude.  nv < 0: no table (more than four differing bits): the term loop.
struct PoolRun {
    uint64_t z0;
    uint32_t vpos;   // four 6-bit bit numbers of V
    int32_t nv;      // bits in V, or -1
    int32_t toff;    // first entry of the run's table (double2 per pattern)
    int32_t pad;
};
// block (chunk c, operator k): the slice [c, c+1) * count / gridDim.x of the support list; same value as k_pool_grad
__global__ __launch_bounds__(256) void k_pool_grad_nz(const amp_t *__restrict__ sig, const uint64_t *__restrict__ idx,
                                                      const amp_t *__restrict__ val, uint64_t count, uint64_t base,
                                                      const int64_t *__restrict__ offsets, const uint64_t *__restrict__ xs,
                                                      const HTerm *__restrict__ terms, int64_t op0, double2 *__restrict__ partials,
                                                      const PoolRun *__restrict__ runs, const double2 *__restrict__ tabs) {
    __shared__ double2 red[4];
    const int64_t op = op0 + blockIdx.y;
    const uint64_t e0 = count * blockIdx.x / gridDim.x, e1 = count * (blockIdx.x + 1ull) / gridDim.x;
    double2 acc = make_double2(0.0, 0.0);
    const int64_t t0 = offsets[op], t1 = offsets[op + 1];
    // a listed amplitude is read ONCE per operator and meets all of the operator's same-x runs (the list was re-read per run: 36 GB
    // for 665 operators x 600 k amplitudes); per run the coefficient first, sigma's amplitude only where it is not zero: an
    // excitation connects about one determinant in sixteen, and the gather is a random 16 bytes of the register
    for (uint64_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const uint64_t jl = idx[e], gj = base | jl;
        const amp_t k = val[e];
        for (int64_t t = t0; t < t1;) {
            const uint64_t x = xs[t];
            int64_t te = t + 1;
            while (te < t1 && xs[te] == x) ++te;
            double cr = 0.0, ci = 0.0;
            const PoolRun run = runs ? runs[t] : PoolRun{0ull, 0u, -1, 0, 0};
            if (run.nv >= 0) {
                const uint32_t pat = (uint32_t)((gj >> (run.vpos & 63u)) & 1ull) | (uint32_t)(((gj >> ((run.vpos >> 6) & 63u)) & 1ull) << 1) |
                                     (uint32_t)(((gj >> ((run.vpos >> 12) & 63u)) & 1ull) << 2) |
                                     (uint32_t)(((gj >> ((run.vpos >> 18) & 63u)) & 1ull) << 3);
                const double2 d = tabs[run.toff + (int)(pat & ((1u << run.nv) - 1u))];
                const double sg = parity_sign64(gj & run.z0);
                cr = d.x * sg;
                ci = d.y * sg;
            } else {
                for (int64_t u = t; u < te; ++u) {
                    const HTerm ht = terms[u];
                    const double sg = parity_sign64(gj & ht.z);
                    cr = fma(ht.cr, sg, cr);
                    ci = fma(ht.ci, sg, ci);
                }
            }
            t = te;
            if (cr == 0.0 && ci == 0.0) continue;
            const amp_t b = sig[jl ^ x];
            const double vx = b.x * k.x + b.y * k.y;
            const double vy = b.x * k.y - b.y * k.x;
            acc.x += cr * vx - ci * vy;
            acc.y += cr * vy + ci * vx;
        }
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[(size_t)(op - op0) * gridDim.x + blockIdx.x] = t;
}

// out[row] = sum of the row's `count` double2 partials (fixed order)
__global__ __launch_bounds__(256) void k_reduce_rows2(const double2 *__restrict__ partials, int count,
                                                      double2 *__restrict__ out) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    const double2 *row = partials + (size_t)blockIdx.x * count;
    for (int i = threadIdx.x; i < count; i += 256) {
        acc.x += row[i].x;
        acc.y += row[i].y;
    }
    const double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_basis(amp_t *__restrict__ st, uint64_t namps, uint64_t local_index,
                                                    int has_one, double2 one) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        st[i] = (has_one && i == local_index) ? one : make_double2(0.0, 0.0);
    }
}

// counter-based integer hash -> two doubles in [-1,1): exactly reproducible on the host
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t v) {
    v += 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
}
__host__ __device__ __forceinline__ double unit_pm1(uint64_t bits) {
    // 53 random bits -> [0,1) -> [-1,1)
    return (double)(bits >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

__global__ __launch_bounds__(256) void k_randomize(amp_t *__restrict__ st, uint64_t namps, uint64_t base,
                                                   uint64_t seed, double scale, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const uint64_t g = base | i;
        const uint64_t h = mix64(seed ^ mix64(g));
        amp_t a;
        a.x = unit_pm1(mix64(h ^ 0x1234567ull)) * scale;
        a.y = unit_pm1(mix64(h ^ 0x89ABCDEFull)) * scale;
        st[i] = a;
        acc.x += a.x * a.x + a.y * a.y;
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_scale(amp_t *__restrict__ st, uint64_t namps, double scale) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        amp_t a = st[i];
        a.x *= scale;
        a.y *= scale;
        st[i] = a;
    }
}

__global__ __launch_bounds__(256) void k_norm2(const amp_t *__restrict__ st, uint64_t namps,
                                               double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const amp_t a = st[i];
        acc.x += a.x * a.x + a.y * a.y;
    }
    double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// angle table of a program for one parameter vector: (z, coeff, phi0, pidx, ny) -> (z, cos, +-sin, odd)
struct RotSpec {  // = SmallRot of sv_small.hpp (declared here to keep this header self-contained)
    uint64_t z;
    double coeff, phi0;
    int32_t pidx, ny;
};
__global__ __launch_bounds__(256) void k_resolve_rots(const RotSpec *__restrict__ spec, int count,
                                                      const double *__restrict__ theta, RotParam *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= count) return;
    const RotSpec sr = spec[r];
    const double phi = sr.phi0 + (sr.pidx >= 0 ? sr.coeff * theta[sr.pidx] : 0.0);
    double sn, c;
    sincos(phi, &sn, &c);
    RotParam rp;
    rp.z = sr.z;
    rp.c = c;
    rp.s = (sr.ny & 2) ? -sn : sn;
    rp.odd = sr.ny & 1;
    rp.pad = 0;
    out[r] = rp;
}

// angle tables of a batch: out[b * count + r] from theta[b * K + .]  (blockIdx.y = b)
__global__ __launch_bounds__(256) void k_resolve_rots_batch(const RotSpec *__restrict__ spec, int count, const double *__restrict__ theta,
                                                            int K, RotParam *__restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= count) return;
    const RotSpec sr = spec[r];
    const double phi = sr.phi0 + (sr.pidx >= 0 ? sr.coeff * theta[(size_t)blockIdx.y * K + sr.pidx] : 0.0);
    double sn, c;
    sincos(phi, &sn, &c);
    RotParam rp;
    rp.z = sr.z;
    rp.c = c;
    rp.s = (sr.ny & 2) ? -sn : sn;
    rp.odd = sr.ny & 1;
    rp.pad = 0;
    out[(size_t)blockIdx.y * count + r] = rp;
}

// ---- real-amplitude state (2^n doubles): one-op sweeps for what does not fit a tile ----------------------------
// run of odd-ny rotations sharing x: u' = c u + s_i v, v' = c v + s_j u
__global__ __launch_bounds__(256) void k_rot_pairs_real(double *__restrict__ st, uint64_t npairs, int pivot, uint64_t x,
                                                        uint64_t base, const RotParam *__restrict__ rp, int nrot) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < npairs; k += stride) {
        const uint64_t i = insert_zero(k, pivot), j = i ^ x;
        double u = st[i], v = st[j];
        for (int r = 0; r < nrot; ++r) {
            const RotParam rr = rp[r];
            const int pi = parity64((base | i) & rr.z);
            const double si = (pi ^ 1) ? -rr.s : rr.s, sj = pi ? -rr.s : rr.s;  // odd ny: pj = pi ^ 1
            const double nu = rr.c * u + si * v, nv = rr.c * v + sj * u;
            u = nu;
            v = nv;
        }
        st[i] = u;
        st[j] = v;
    }
}

__global__ __launch_bounds__(256) void k_gate_real(double *__restrict__ st, uint64_t nwork, int kind, int b0, int b1) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < nwork; k += stride) {
        uint64_t i, j;
        if (kind == 2) {
            const int lo = b0 < b1 ? b0 : b1, hi = b0 < b1 ? b1 : b0;
            i = insert_zero(insert_zero(k, lo), hi) | (1ull << b0);
            j = i | (1ull << b1);
        } else {
            i = insert_zero(k, b0);
            j = i | (1ull << b0);
        }
        const double a = st[i], b = st[j];
        if (kind == 1) {
            const double r = 0.70710678118654752440;
            st[i] = (a + b) * r;
            st[j] = (a - b) * r;
        } else {
            st[i] = b;
            st[j] = a;
        }
    }
}

// pair-trick expectation of the groups [g0, g1) on a real state (terms with an imaginary folded coefficient vanish)
__global__ __launch_bounds__(256) void k_expect_pairs_real(const double *__restrict__ st, uint64_t namps,
                                                           const HGroup *__restrict__ groups, int g0, int g1,
                                                           const HTerm *__restrict__ terms,
                                                           double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (int g = g0; g < g1; ++g) {
        const HGroup gr = groups[g];
        const bool diag = gr.x == 0;
        const int pivot = diag ? 0 : 63 - __clzll(gr.x);
        const uint64_t cnt = diag ? namps : (namps >> 1);
        for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < cnt; k += stride) {
            const uint64_t i = diag ? k : insert_zero(k, pivot), j = i ^ gr.x;
            const uint64_t gj = gr.jbase | j;
            double d = 0.0;
            for (int t = gr.t0; t < gr.t1; ++t) {
                const HTerm ht = terms[t];
                d += parity64(gj & ht.z) ? -ht.cr : ht.cr;
            }
            acc += (diag ? 1.0 : 2.0) * d * st[i] * st[j];
        }
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_init_basis_real(double *__restrict__ st, uint64_t namps, uint64_t index) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) st[i] = i == index ? 1.0 : 0.0;
}

// ---- adjoint (reverse-mode) gradient of E(theta) = <psi(theta)|H|psi(theta)>  (ovqe_energy_gradient) ----------
// Backward pass over one same-x run, rotations in REVERSE order.  psi holds U_r...U_1|hf>, lam holds
// U_{r+1}^+...U_R^+ H|psi_R>; for every rotation r of the run the kernel accumulates
//   w_r = sum_pairs  Re/Im [ s_i conj(lam_i) psi_j + s_j conj(lam_j) psi_i ]      (Re for odd ny, Im for even)
// (dE/dtheta_p = sum_r 2 coeff_r kappa_r w_r, kappa = -1 if ny & 2; assembled on the host) and then un-rotates both
// states, psi <- U_r^+ psi, lam <- U_r^+ lam.  One read + one write of BOTH states per run.
constexpr int ADJ_MAX_ROT = 16;

__global__ __launch_bounds__(256) void k_adjoint_pairs(amp_t *__restrict__ psi, amp_t *__restrict__ lam,
                                                       uint64_t npairs, int pivot, uint64_t x, uint64_t base,
                                                       const RotParam *__restrict__ rp, int nrot,
                                                       double *__restrict__ partials /* [nrot][gridDim.x] */) {
    __shared__ double2 red[4];
    double w[ADJ_MAX_ROT];
#pragma unroll
    for (int r = 0; r < ADJ_MAX_ROT; ++r) w[r] = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < npairs; k += stride) {
        const uint64_t i = insert_zero(k, pivot), j = i ^ x;
        amp_t u = psi[i], v = psi[j], lu = lam[i], lv = lam[j];
#pragma unroll
        for (int q = 0; q < ADJ_MAX_ROT; ++q) {
            const int r = nrot - 1 - q;
            if (r < 0) continue;
            RotParam rr = rp[r];
            const int pi = parity64((base | i) & rr.z);
            const int pj = pi ^ rr.odd;
            const double si = pj ? -1.0 : 1.0, sj = pi ? -1.0 : 1.0;
            // t = s_i conj(lam_i) psi_j + s_j conj(lam_j) psi_i
            const double tr = si * (lu.x * v.x + lu.y * v.y) + sj * (lv.x * u.x + lv.y * u.y);
            const double ti = si * (lu.x * v.y - lu.y * v.x) + sj * (lv.x * u.y - lv.y * u.x);
            w[q] += rr.odd ? tr : ti;
            rr.s = -rr.s;  // U^+ = exp(+i phi P)
            rot_pair(u, v, rr, base | i);
            rot_pair(lu, lv, rr, base | i);
        }
        psi[i] = u;
        psi[j] = v;
        lam[i] = lu;
        lam[j] = lv;
    }
#pragma unroll
    for (int q = 0; q < ADJ_MAX_ROT; ++q) {
        if (q < nrot) {  // nrot is uniform: every thread takes the same path through the barriers of block_sum
            const double2 t = block_sum<256>(make_double2(w[q], 0.0), red);
            if (threadIdx.x == 0) partials[(size_t)(nrot - 1 - q) * gridDim.x + blockIdx.x] = t.x;
        }
    }
}

// diagonal run (x = 0, P = Z^z): w_r = sum_i Im[ s_i conj(lam_i) psi_i ]; un-rotation a <- (c + i s sigma) a
__global__ __launch_bounds__(256) void k_adjoint_diag(amp_t *__restrict__ psi, amp_t *__restrict__ lam, uint64_t namps,
                                                      uint64_t base, const RotParam *__restrict__ rp, int nrot,
                                                      double *__restrict__ partials) {
    __shared__ double2 red[4];
    double w[ADJ_MAX_ROT];
#pragma unroll
    for (int r = 0; r < ADJ_MAX_ROT; ++r) w[r] = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        amp_t a = psi[i], l = lam[i];
#pragma unroll
        for (int q = 0; q < ADJ_MAX_ROT; ++q) {
            const int r = nrot - 1 - q;
            if (r < 0) continue;
            const RotParam rr = rp[r];
            const double sg = parity64((base | i) & rr.z) ? -1.0 : 1.0;
            w[q] += sg * (l.x * a.y - l.y * a.x);
            const double s = -sg * rr.s;  // forward: (c - i s sg) a
            a = make_double2(rr.c * a.x + s * a.y, rr.c * a.y - s * a.x);
            l = make_double2(rr.c * l.x + s * l.y, rr.c * l.y - s * l.x);
        }
        psi[i] = a;
        lam[i] = l;
    }
#pragma unroll
    for (int q = 0; q < ADJ_MAX_ROT; ++q) {
        if (q < nrot) {  // nrot is uniform: every thread takes the same path through the barriers of block_sum
            const double2 t = block_sum<256>(make_double2(w[q], 0.0), red);
            if (threadIdx.x == 0) partials[(size_t)(nrot - 1 - q) * gridDim.x + blockIdx.x] = t.x;
        }
    }
}

// out[row] = sum of the row's `count` partials (one block per row, fixed order)
__global__ __launch_bounds__(256) void k_reduce_rows(const double *__restrict__ partials, int count,
                                                     double *__restrict__ out) {
    __shared__ double2 red[4];
    double acc = 0.0;
    const double *row = partials + (size_t)blockIdx.x * count;
    for (int i = threadIdx.x; i < count; i += 256) acc += row[i];
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) out[blockIdx.x] = t.x;
}

// ---- Lanczos support (ovqe_ground_state) ---------------------------------------------------------------
// partials[b] = sum_i conj(a_i) b_i over the block's grid-stride slice
__global__ __launch_bounds__(256) void k_dot(const amp_t *__restrict__ a, const amp_t *__restrict__ b, uint64_t namps,
                                             double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double2 acc = make_double2(0.0, 0.0);
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const amp_t u = a[i], v = b[i];
        acc.x += u.x * v.x + u.y * v.y;
        acc.y += u.x * v.y - u.y * v.x;
    }
    const double2 t = block_sum<256>(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// w <- w - alpha v - beta vprev ;  partials[b] = sum |w_i|^2   (vprev may be null)
__global__ __launch_bounds__(256) void k_lanczos_update(amp_t *__restrict__ w, const amp_t *__restrict__ v,
                                                        const amp_t *__restrict__ vprev, double alpha, double beta,
                                                        uint64_t namps, double2 *__restrict__ partials) {
    __shared__ double2 red[4];
    double acc = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        amp_t r = w[i];
        const amp_t a = v[i];
        r.x -= alpha * a.x;
        r.y -= alpha * a.y;
        if (vprev) {
            const amp_t b = vprev[i];
            r.x -= beta * b.x;
            r.y -= beta * b.y;
        }
        w[i] = r;
        acc += r.x * r.x + r.y * r.y;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// y <- (init ? 0 : y) + s * v
__global__ __launch_bounds__(256) void k_axpy_real(amp_t *__restrict__ y, const amp_t *__restrict__ v, double s,
                                                   uint64_t namps, int init) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < namps; i += stride) {
        const amp_t a = v[i];
        amp_t r = init ? make_double2(0.0, 0.0) : y[i];
        r.x += s * a.x;
        r.y += s * a.y;
        y[i] = r;
    }
}

__global__ __launch_bounds__(256) void k_gather(const amp_t *__restrict__ st, int64_t count,
                                                const uint64_t *__restrict__ idx, amp_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = st[idx[i]];
}

}  // namespace ovqe
