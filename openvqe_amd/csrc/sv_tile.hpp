// sv_tile.hpp — LDS-tiled multi-op sweep of the streaming path (n >= 13).
//
// A literal gate (X / H / CNOT / RX / RY / RZ) or a same-x run of Pauli rotations only mixes amplitudes whose
// indices differ in its "mixing" bits: the x mask of a rotation, the target of a gate.  z masks and CNOT controls
// are diagonal — they only select a sign / a branch.  So a run of consecutive ops whose mixing bits fit a set S of
// M index bits acts independently on each of the 2^(n-M) tiles {base | deposit(e, S)}: one workgroup loads its tile
// (16 B x 2^M) into LDS, applies the whole run there (arithmetic and order identical to the one-op-per-sweep
// kernels), and writes it back — ONE read + ONE write of the state for the whole run instead of one per op.
// S always contains the lowest index bits, so global accesses stay in contiguous chunks of >= 256 B.
//
// The QUCCSD templates of the reference (ref:openvqe/common_files/circuit.py:13-106: ~25 literal gates on 2 or 4
// qubits) and the 2-/4-qubit x masks of JW excitations are exactly this shape.
#pragma once
#include "sv_small.hpp"

namespace ovqe {

constexpr int TILE_ROT_CAP = 256;  // rotation entries staged in LDS per segment

struct TileSeg {       // one HBM sweep
    uint64_t smask;    // the tile's index bits (|smask| = M)
    uint64_t mask_lo;  // the lowest log2(NT) tile bits: filled from the thread index
    uint64_t mask_hi;  // the remaining tile bits: filled from the trip counter
    int32_t op0, op1;  // TileOp range
    int32_t rot0, rot1;  // rotation range (ops are consecutive, so their table entries are too)
};

struct TileOp {
    uint32_t x;      // tile-local x mask (OP_PAIR); 1 << local target (OP_X, OP_H, OP_CNOT)
    int16_t kind;    // SmallOpKind
    int16_t pivot;   // local pivot / target bit
    int32_t first;   // OP_PAIR / OP_DIAG: first rotation (absolute);  OP_CNOT: local control bit, or
                     // -1 - (global bit) when the control lies outside the tile
    int32_t count;   // rotations of the run
};

struct TileRot {     // static part of a rotation inside its segment
    uint64_t zout;   // z outside the tile: a per-tile sign
    uint32_t zin;    // z on the tile bits, tile-local
    uint32_t pad;
};

// pdep(v, mask): spread the low bits of v over the set bits of mask (ascending); mask is wave-uniform
__device__ __forceinline__ uint64_t spread_bits(uint32_t v, uint64_t mask) {
    uint64_t r = 0;
    while (mask) {
        const int p = __ffsll((long long)mask) - 1;
        r |= (uint64_t)(v & 1u) << p;
        v >>= 1;
        mask &= mask - 1ull;
    }
    return r;
}

template <int M, int NT, bool NTL>
__global__ __launch_bounds__(NT) void k_tile_sweep(amp_t *__restrict__ st, uint64_t base, TileSeg seg,
                                                   const TileOp *__restrict__ ops, const TileRot *__restrict__ trot,
                                                   const RotParam *__restrict__ rp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NEL = 1u << M;
    constexpr int TRIPS = NEL / NT;
    constexpr int U = (NEL / 2 / NT) >= 4 ? 4 : ((NEL / 2 / NT) >= 2 ? 2 : 1);
    double2 *tile = reinterpret_cast<double2 *>(smem);
    RotLds *tab = reinterpret_cast<RotLds *>(smem + (size_t)NEL * sizeof(double2));
    v2d *p = reinterpret_cast<v2d *>(st);

    // tile base: the block index spread over the index bits NOT in the tile
    uint64_t tb = blockIdx.x;
    for (uint64_t mk = seg.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, seg.mask_lo);
    const uint64_t gbase = base | tb;

    v2d reg[TRIPS];
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
        const uint64_t g = tb | glow | spread_bits((uint32_t)j, seg.mask_hi);
        reg[j] = NTL ? __builtin_nontemporal_load(&p[g]) : p[g];
    }
    // rotation table of the segment, per-tile signs folded into sin
    for (int r = seg.rot0 + (int)threadIdx.x; r < seg.rot1; r += NT) {
        const RotParam rr = rp[r];
        const TileRot tr = trot[r];
        RotLds rl;
        rl.c = rr.c;
        rl.s = parity64(gbase & tr.zout) ? -rr.s : rr.s;
        rl.z = tr.zin;
        rl.odd = (uint32_t)rr.odd;
        rl.pad = 0;
        tab[r - seg.rot0] = rl;
    }
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) tile[threadIdx.x + j * NT] = make_double2(reg[j].x, reg[j].y);
    __syncthreads();

    for (int o = seg.op0; o < seg.op1; ++o) {
        const TileOp top = ops[o];
        SmallOp op;
        op.x = top.x;
        op.kind = top.kind;
        op.first = top.first;
        op.count = top.count;
        op.pivot = top.pivot;
        if (op.kind == OP_PAIR) {
            small_pass_pair<false, NT, U>(tile, NEL >> 1, op, tab + (op.first - seg.rot0));
        } else if (op.kind == OP_DIAG) {
            small_pass_diag<NT>(tile, NEL, op, tab + (op.first - seg.rot0));
        } else if (op.kind == OP_CNOT) {
            if (top.first >= 0) {
                op.first = top.first;   // local control
                op.count = top.pivot;   // local target
                small_pass_gate<false, NT>(tile, NEL, op);
            } else if ((gbase >> (-1 - top.first)) & 1ull) {
                op.kind = OP_X;         // control bit is set on the whole tile
                small_pass_gate<false, NT>(tile, NEL, op);
            }
        } else {
            small_pass_gate<false, NT>(tile, NEL, op);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
        const double2 a = tile[threadIdx.x + j * NT];
        const v2d t = {a.x, a.y};
        const uint64_t g = tb | glow | spread_bits((uint32_t)j, seg.mask_hi);
        if (NTL) __builtin_nontemporal_store(t, &p[g]); else p[g] = t;
    }
}

}  // namespace ovqe
