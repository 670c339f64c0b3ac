// sv_small.hpp — whole-circuit, parameter-batched energy kernel for small registers (n <= 16).
//
// At n = 12..14 (LiH / H2O) the state is 64..256 KiB: one sweep is far too short to amortise a
// launch, so one workgroup owns one statevector for the WHOLE evaluation: |HF> init, every fused
// Pauli-rotation pass / literal gate of the ansatz, and <psi|H|psi> — one launch for B parameter
// vectors (a finite-difference gradient is B = K+1).  The state lives in LDS when it fits
// (n <= 13, 128 KiB of the 160 KiB) and in a per-workgroup slice of an L2/MALL-resident workspace
// otherwise; cos/sin of every rotation angle are tabulated in LDS once per segment so the sweep
// itself is pure FMA work on 16-byte amplitude loads.
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

struct SmallArgs {
    int n;
    int64_t B;
    const double *theta;  // B x K
    int K;
    const SmallOp *ops;
    const SmallRot *rots;
    const SmallSeg *segs;
    int nsegs;
    const HGroup *groups;
    int ngroups;
    const HTerm *terms;
    double constant;
    uint64_t hf;
    amp_t *workspace;   // gridDim.x slices of 2^n amplitudes (global-state variant / final state)
    double *energies;   // B
    int keep_state;     // 1: copy the final state of b == 0 to workspace[0 .. 2^n)
    int cs_capacity;    // entries of the LDS cos/sin table
};

template <typename StatePtr, int NT>
__device__ __forceinline__ void small_pass_pair(StatePtr st, uint64_t npairs, const SmallOp &op,
                                                const SmallRot *__restrict__ rots, const double2 *cs, int rot_base) {
    for (uint64_t k = threadIdx.x; k < npairs; k += NT) {
        const uint64_t i = insert_zero(k, op.pivot), j = i ^ op.x;
        amp_t u = st[i], v = st[j];
        for (int r = 0; r < op.count; ++r) {
            const SmallRot sr = rots[op.first + r];
            const double2 c = cs[op.first + r - rot_base];
            RotParam rp;
            rp.z = sr.z;
            rp.c = c.x;
            rp.s = c.y;
            rp.odd = sr.ny & 1;
            rot_pair(u, v, rp, i);
        }
        st[i] = u;
        st[j] = v;
    }
}

template <typename StatePtr, int NT>
__device__ __forceinline__ void small_pass_diag(StatePtr st, uint64_t namps, const SmallOp &op,
                                                const SmallRot *__restrict__ rots, const double2 *cs, int rot_base) {
    for (uint64_t i = threadIdx.x; i < namps; i += NT) {
        amp_t a = st[i];
        for (int r = 0; r < op.count; ++r) {
            const SmallRot sr = rots[op.first + r];
            const double2 c = cs[op.first + r - rot_base];
            const double s = parity64(i & sr.z) ? -c.y : c.y;
            amp_t t;
            t.x = c.x * a.x + s * a.y;
            t.y = c.x * a.y - s * a.x;
            a = t;
        }
        st[i] = a;
    }
}

template <typename StatePtr, int NT>
__device__ __forceinline__ void small_pass_gate(StatePtr st, uint64_t namps, const SmallOp &op) {
    if (op.kind == OP_CNOT) {
        const int cb = op.first, tb = op.count;
        const int lo = cb < tb ? cb : tb, hi = cb < tb ? tb : cb;
        for (uint64_t k = threadIdx.x; k < (namps >> 2); k += NT) {
            const uint64_t i = insert_zero(insert_zero(k, lo), hi) | (1ull << cb), j = i | (1ull << tb);
            const amp_t a = st[i], b = st[j];
            st[i] = b;
            st[j] = a;
        }
        return;
    }
    for (uint64_t k = threadIdx.x; k < (namps >> 1); k += NT) {
        const uint64_t i = insert_zero(k, op.pivot), j = i | op.x;
        const amp_t a = st[i], b = st[j];
        if (op.kind == OP_H) {
            const double r = 0.70710678118654752440;
            st[i] = make_double2((a.x + b.x) * r, (a.y + b.y) * r);
            st[j] = make_double2((a.x - b.x) * r, (a.y - b.y) * r);
        } else {
            st[i] = b;
            st[j] = a;
        }
    }
}

template <bool LDS_STATE, int NT>
__global__ __launch_bounds__(NT) void k_small_vqe(SmallArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t namps = 1ull << A.n;
    // LDS layout: [state (LDS_STATE)] [cos/sin table] [reduction scratch]
    amp_t *lds_state = reinterpret_cast<amp_t *>(smem);
    double2 *cs = reinterpret_cast<double2 *>(smem + (LDS_STATE ? namps * sizeof(amp_t) : 0));
    double2 *red = cs + A.cs_capacity;
    amp_t *gst = A.workspace + (uint64_t)blockIdx.x * namps;

    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        const double *th = A.theta + b * A.K;
        // |HF>
        for (uint64_t i = threadIdx.x; i < namps; i += NT) {
            const amp_t v = make_double2(i == A.hf ? 1.0 : 0.0, 0.0);
            if (LDS_STATE) lds_state[i] = v; else gst[i] = v;
        }
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
                if (LDS_STATE) {
                    if (op.kind == OP_PAIR) small_pass_pair<amp_t *, NT>(lds_state, namps >> 1, op, A.rots, cs, sg.rot0);
                    else if (op.kind == OP_DIAG) small_pass_diag<amp_t *, NT>(lds_state, namps, op, A.rots, cs, sg.rot0);
                    else small_pass_gate<amp_t *, NT>(lds_state, namps, op);
                } else {
                    if (op.kind == OP_PAIR) small_pass_pair<amp_t *, NT>(gst, namps >> 1, op, A.rots, cs, sg.rot0);
                    else if (op.kind == OP_DIAG) small_pass_diag<amp_t *, NT>(gst, namps, op, A.rots, cs, sg.rot0);
                    else small_pass_gate<amp_t *, NT>(gst, namps, op);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // <psi|H|psi>: Hermitian pair trick — (i,j) and (j,i) contributions are complex conjugates
        double acc = 0.0;
        for (int g = 0; g < A.ngroups; ++g) {
            const HGroup gr = A.groups[g];
            if (gr.x == 0) {
                for (uint64_t i = threadIdx.x; i < namps; i += NT) {
                    const amp_t a = LDS_STATE ? lds_state[i] : gst[i];
                    double d = 0.0;
                    for (int t = gr.t0; t < gr.t1; ++t) {
                        const HTerm ht = A.terms[t];
                        d += parity64(i & ht.z) ? -ht.cr : ht.cr;
                    }
                    acc += d * (a.x * a.x + a.y * a.y);
                }
            } else {
                const int p = 63 - __clzll(gr.x);
                for (uint64_t k = threadIdx.x; k < (namps >> 1); k += NT) {
                    const uint64_t i = insert_zero(k, p), j = i ^ gr.x;
                    const amp_t a = LDS_STATE ? lds_state[i] : gst[i];
                    const amp_t c = LDS_STATE ? lds_state[j] : gst[j];
                    double dr = 0.0, di = 0.0;
                    for (int t = gr.t0; t < gr.t1; ++t) {
                        const HTerm ht = A.terms[t];
                        const bool neg = parity64(j & ht.z);
                        dr += neg ? -ht.cr : ht.cr;
                        di += neg ? -ht.ci : ht.ci;
                    }
                    const double vx = a.x * c.x + a.y * c.y;  // conj(a_i) a_j
                    const double vy = a.x * c.y - a.y * c.x;
                    acc += 2.0 * (dr * vx - di * vy);
                }
            }
        }
        const double2 tot = block_sum<NT>(make_double2(acc, 0.0), red);
        if (threadIdx.x == 0) A.energies[b] = tot.x + A.constant;
        if (A.keep_state && b == 0 && LDS_STATE) {
            for (uint64_t i = threadIdx.x; i < namps; i += NT) A.workspace[i] = lds_state[i];
        }
        __syncthreads();
    }
}

}  // namespace ovqe
