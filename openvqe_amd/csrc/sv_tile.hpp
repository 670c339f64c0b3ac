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
#include <type_traits>

namespace ovqe {

constexpr int TILE_ROT_CAP = 256;  // rotation entries staged in LDS per segment
constexpr int TILE_SWEEP_LOG_NT = 9;  // threads per workgroup of k_tile_sweep

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
    int32_t count;   // rotations of the run / active patterns (OP_TAB)
    uint32_t zc;     // OP_TAB: the run's z mask outside x, on the tile bits
};

struct TileRot {     // static part of a rotation inside its segment
    uint64_t zout;   // z outside the tile: a per-tile sign
    uint32_t zin;    // z on the tile bits, tile-local
    uint32_t pad;
};

// LDS bank swizzle of a tile.  The pairs of a (group, pattern) entry / of a rotation share their low index bits whenever
// the x mask sits on the low qubits, i.e. every lane of the wave would hit the SAME bank (measured: LDS bank-conflict
// cycles = 95 % of the busy cycles of k_tile_expect).  The 16-byte element v is therefore kept at v ^ ((v >> 3) & 7):
// elements that agree in their low bits but differ above spread over the 8 element slots of a 128-byte LDS row, and
// consecutive elements stay conflict-free.  For a real tile the 16-byte element is a PAIR of amplitudes: amplitude e
// lives at e ^ (((e >> 4) & 7) << 1).
template <bool REAL>
__device__ __forceinline__ uint32_t tile_swz(uint32_t e) {
    return REAL ? e ^ (((e >> 4) & 7u) << 1) : e ^ ((e >> 3) & 7u);
}
__device__ __forceinline__ uint32_t tile_swz_v(uint32_t v) { return v ^ ((v >> 3) & 7u); }
template <bool REAL>
struct TileView {  // what the passes of sv_small.hpp index instead of the raw LDS pointer
    typename Amp<REAL>::T *p;
    __device__ __forceinline__ typename Amp<REAL>::T &operator[](uint32_t e) const { return p[tile_swz<REAL>(e)]; }
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

// REAL = true: the state is stored as 2^n doubles (a program whose rotations all have an odd number of Y, starting from
// a basis state, keeps the amplitudes real): the same code moves 16-byte elements = PAIRS of amplitudes, i.e. the
// segment's masks live in the index space of the pairs (amplitude index >> 1, bit 0 always inside the tile), and the
// tile holds 2^M doubles — twice the amplitudes per LDS byte, so one more mixing bit per sweep at half the HBM bytes.
template <int M, int NT, bool NTL, bool REAL>
__global__ __launch_bounds__(NT) void k_tile_sweep(void *__restrict__ st, uint64_t base, TileSeg seg,
                                                   const TileOp *__restrict__ ops, const TileRot *__restrict__ trot,
                                                   const RotParam *__restrict__ rp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Amp<REAL>::T amp;
    constexpr uint32_t NEL = 1u << M;                 // amplitudes per tile
    constexpr uint32_t NELV = REAL ? NEL / 2 : NEL;   // 16-byte elements per tile
    constexpr int TRIPS = NELV / NT;
    constexpr int U = (NEL / 2 / NT) >= 4 ? 4 : ((NEL / 2 / NT) >= 2 ? 2 : 1);
    const TileView<REAL> tile{reinterpret_cast<amp *>(smem)};  // bank-swizzled (see tile_swz)
    double2 *tilev = reinterpret_cast<double2 *>(smem);
    RotLds *tab = reinterpret_cast<RotLds *>(smem + (size_t)NELV * sizeof(double2));
    v2d *p = reinterpret_cast<v2d *>(st);

    // tile base: the block index spread over the index bits NOT in the tile
    uint64_t tb = blockIdx.x;
    for (uint64_t mk = seg.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, seg.mask_lo);
    const uint64_t gbase = base | (REAL ? tb << 1 : tb);

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
    for (int j = 0; j < TRIPS; ++j) tilev[tile_swz_v(threadIdx.x + j * NT)] = make_double2(reg[j].x, reg[j].y);
    __syncthreads();

    for (int o = seg.op0; o < seg.op1; ++o) {
        const TileOp top = ops[o];
        SmallOp op;
        op.x = top.x;
        op.kind = top.kind;
        op.first = top.first;
        op.count = top.count;
        op.pivot = top.pivot;
        if (op.kind == OP_TAB) {
            op.zc = top.zc;
            op.fixmask = top.x;
            small_pass_tab<REAL, NT>(tile, M, op, tab + (op.first - seg.rot0));
        } else if (op.kind == OP_PAIR) {
            small_pass_pair<REAL, NT, U>(tile, NEL >> 1, op, tab + (op.first - seg.rot0));
        } else if (op.kind == OP_DIAG) {
            if constexpr (!REAL) small_pass_diag<NT>(tile, NEL, op, tab + (op.first - seg.rot0));
        } else if (op.kind == OP_CNOT) {
            if (top.first >= 0) {
                op.first = top.first;   // local control
                op.count = top.pivot;   // local target
                small_pass_gate<REAL, NT>(tile, NEL, op);
            } else if ((gbase >> (-1 - top.first)) & 1ull) {
                op.kind = OP_X;         // control bit is set on the whole tile
                small_pass_gate<REAL, NT>(tile, NEL, op);
            }
        } else {
            small_pass_gate<REAL, NT>(tile, NEL, op);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
        const double2 a = tilev[tile_swz_v(threadIdx.x + j * NT)];
        const v2d t = {a.x, a.y};
        const uint64_t g = tb | glow | spread_bits((uint32_t)j, seg.mask_hi);
        if (NTL) __builtin_nontemporal_store(t, &p[g]); else p[g] = t;
    }
}

// ---- <psi|H|psi> on tiles --------------------------------------------------------------------------------
// The x-groups of a Hermitian Pauli sum are covered by tile bit sets: every group whose x mask lies inside the sweep's
// set S is evaluated from the LDS copy of the tile (E_g = 2 Re sum_pairs D(j) conj(a_i) a_j, the pair trick of
// k_expect_pairs), so the state is read ONCE per sweep instead of once per x-group (JW Hamiltonians: weight-2/-4
// masks; ~25 groups per sweep at 24 qubits).
// Work items are (group, pattern) ENTRIES: for a fixed pattern of the pair's bits on the x positions, the x part of
// every z mask is a constant sign, so terms that agree outside x merge on the host — the 8 strings of a JW double
// excitation become ONE coefficient on 2 of the 8 patterns, XX/YY number-operator families halve — and patterns whose
// coefficients cancel exactly are never visited.  One wave owns an entry (<= 256 pairs); z bits outside the tile
// are per-tile signs, folded into the coefficients while a chunk of the term table is staged in LDS.
constexpr int TILE_TERM_CAP = 512;
constexpr int TILE_APPLY_GROUPS = 128;  // x-groups of a chunk staged in LDS (operator-application form)
constexpr int TILE_APPLY_TERMS = 16;    // terms per piece of an x-group in that form
constexpr int TILE_ENTRY_PAIRS = 256;
constexpr int TILE_EXPECT_LOG_NT = 9;   // threads per workgroup of k_tile_expect (256 measured faster than 1024)

struct ExSweep {
    uint64_t smask, mask_lo, mask_hi;
    int32_t c0, c1;   // chunk range (entries with more than two merged terms: one wave per entry)
    int32_t i0, i1;   // flat items (entries with one or two merged terms: one LANE per 64 pairs)
    int32_t a0, a1;   // chunk range of the operator-application form (k_tile_apply)
};
struct ExAGroupT {    // x-group of the sweep with its raw terms (k_tile_apply)
    uint32_t x;       // tile-local x mask
    int32_t t0, t1;   // terms (absolute, in the apply term table)
    int32_t pad;
};
// A (group, pattern) entry with at most two merged terms — nearly all of them: a JW double excitation leaves ONE
// coefficient per active pattern — costs more in per-entry set-up than in arithmetic when a whole wave serves it.
// Such entries are cut into items of TILE_ITEM_PAIRS pairs and every LANE walks its own item (incremental index,
// x positions skipped): the set-up is paid per lane, 64 items side by side, all of equal length.
constexpr int TILE_ITEM_PAIRS = 64;
struct ExFlatT {
    uint32_t x, ibits;      // tile-local x mask / pattern bits
    uint32_t zin0, zin1;    // the terms' z on the tile bits (x positions cleared)
    uint64_t zout0, zout1;  // ... and outside the tile
    double c0r, c0i, c1r, c1i;
};
struct ExItemT {
    uint32_t entry;   // index into the flat entries
    uint32_t istart;  // tile-local index of the item's first pair (pattern bits included)
    uint32_t count;   // pairs (<= TILE_ITEM_PAIRS, even)
    uint32_t pad;
};
struct ExChunkT {
    int32_t g0, g1, t0, t1;  // entries, terms
};
struct ExEntryT {
    uint32_t x;       // tile-local x mask (0: diagonal group)
    uint32_t ibits;   // the pattern: bits of i on the non-pivot x positions (pivot bit of i is 0)
    int32_t t0, t1;   // merged terms (absolute)
    int32_t k0, nk;   // free-index range [k0, k0 + nk), nk <= TILE_ENTRY_PAIRS (unsplit entries: <= TILE_UNSPLIT_PAIRS)
    int32_t real_only;  // every ci == 0
    int32_t pad;
    // Round 6, entries of one or two terms (tile_entry_pairs): the TRIP part of the pair index, host-side.  The index of pair k0 + lane +
    // 64 t is dep(lane) | dep(k0 + 64 t) | ibits with disjoint bit sets, the bank swizzle and a term's sign are linear over XOR: trip t's
    // share of the swizzled byte offset and of every term's sign is wave-uniform and tile-independent — tabulated here.  What remains
    // per pair on the device is one XOR per address and one three-input bit operation per term.
    uint32_t ph[16];  // swizzled byte offset of trip t's share of the pair index, t = 0 .. nk / 64 - 1 (the XOR combinations of the
                      // trip-0 offset with one basis offset per set bit of t, made by the host: the device loads them as scalars)
    uint32_t tsign;   // bit t: term 0's parity on trip t's share of the PARTNER index (lane share excluded); bit 16 + t: term 1's
    uint32_t pad2[7];
};
static_assert(sizeof(ExEntryT) == 128, "one entry = two cache lines of scalar loads");
constexpr int TILE_UNSPLIT_PAIRS = 1024;   // pairs per piece of an unsplit entry (16 rows of a wave: the per-entry set-up amortised)
struct ExTermT {
    uint64_t zout;    // z outside the tile
    uint32_t zin;     // z on the tile bits, x positions cleared
    uint32_t pad;
    double cr, ci;    // i^ny and the pattern's sign folded
};
struct ExTermLds {
    double cr, ci;
    uint32_t zin, pad;
};

// Sparse tiles of k_tile_expect: the pieces (x-group restricted to <= TILE_APPLY_TERMS of its terms) of one staging
// chunk over NQ x 64 entries of the tile's non-zero list: sum_i conj(a_i) D(i ^ x) a_{i ^ x}.  The lane's entries (index,
// amplitude) are loaded once; per piece one LDS gather of the partner amplitude per entry, then per (term, entry) four
// vector instructions: j & z, popcount, the coefficient's high word + parity << 31 (a carry-free sign flip), and the add.
constexpr int TILE_SPARSE_TERMS = 1024;   // terms staged in LDS per chunk of the sparse path
constexpr int TILE_SPARSE_GROUPS = 384;   // pieces per chunk
template <bool REAL, int NQ, bool SWZ = true>
__device__ __forceinline__ double tile_sparse_pieces(const typename Amp<REAL>::T *tile, const uint16_t *nz, int nnz, int lb,
                                                     uint32_t lane, const ExAGroupT *sg, int g_begin, int g_end, int g_step,
                                                     const ExTermLds *st) {
    typedef typename Amp<REAL>::T amp;
    uint32_t ii[NQ];
    amp a[NQ];
    bool live[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int l = lb + 64 * q + (int)lane;
        live[q] = l < nnz;
        ii[q] = live[q] ? nz[l] : 0u;
        a[q] = tile[SWZ ? tile_swz<REAL>(ii[q]) : ii[q]];
    }
    double part = 0.0;
    for (int g = g_begin; g < g_end; g += g_step) {
        const ExAGroupT gr = sg[g];
        const uint32_t xl = __builtin_amdgcn_readfirstlane(gr.x);
        const int t0 = __builtin_amdgcn_readfirstlane(gr.t0), t1 = __builtin_amdgcn_readfirstlane(gr.t1);
        uint32_t jj[NQ];
        double vx[NQ], vy[NQ], dr[NQ], di[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            jj[q] = ii[q] ^ xl;
            const amp c = tile[SWZ ? tile_swz<REAL>(jj[q]) : jj[q]];
            if constexpr (REAL) {
                vx[q] = live[q] ? a[q] * c : 0.0;
                vy[q] = 0.0;
            } else {
                vx[q] = live[q] ? a[q].x * c.x + a[q].y * c.y : 0.0;
                vy[q] = live[q] ? a[q].x * c.y - a[q].y * c.x : 0.0;
            }
            dr[q] = 0.0;
            di[q] = 0.0;
        }
        for (int t = t0; t < t1; ++t) {
            const ExTermLds tl = st[t];
            const uint64_t cr = (uint64_t)__double_as_longlong(tl.cr), ci = (uint64_t)__double_as_longlong(tl.ci);
            const uint32_t crl = (uint32_t)cr, crh = (uint32_t)(cr >> 32), cil = (uint32_t)ci, cih = (uint32_t)(ci >> 32);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t par = (uint32_t)__popc(jj[q] & tl.zin);
                dr[q] += __hiloint2double((int)(crh + (par << 31)), (int)crl);
                if constexpr (!REAL) di[q] += __hiloint2double((int)(cih + (par << 31)), (int)cil);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) part += dr[q] * vx[q] - di[q] * vy[q];
    }
    return part;
}

// the pairs of one (group, pattern) entry — or of a piece of an unsplit group — with one or two merged terms: lanes over the free
// index, four trips in flight.
// Round 6: lane part / trip part.  The pair index is dep(lane) | dep(k0 + 64 t) | ibits; swizzle and signs are linear over XOR.  The
// lane's share (swizzled byte offset, its parity on every term folded into the coefficient's high word) is computed once per entry;
// the trip's share comes from the HOST (ExEntryT::pb / tsign: a trip-0 value and one basis value per bit of t) and is combined by
// a handful of scalar XORs per group of four trips.  A pair costs: one XOR for its address, one for the partner's, one per term for
// the sign, the arithmetic.  (A first form that walked the trip index on the scalar unit per trip, and one that computed the basis on
// the device per entry, were SLOWER than the round-5 loop: an entry is 4 .. 8 rows of a wave, the set-up has to stay small — DESIGN §4.)
template <bool REAL, uint32_t NEL, bool ONE, bool RO, bool FULL>
__device__ __forceinline__ double tile_entry_pairs(const typename Amp<REAL>::T *tile, const ExEntryT &en, const ExTermLds &l0, const ExTermLds &l1,
                                                   uint32_t lane) {
    typedef typename Amp<REAL>::T amp;
    constexpr uint32_t AB = (uint32_t)sizeof(amp);
    const uint32_t x = en.x, nk = (uint32_t)en.nk;
    const uint32_t xf = en.pad ? (uint32_t)en.pad : en.x;   // (an unsplit entry: en.x is the pivot bit, en.pad the whole x mask)
    const uint32_t il = deposit_index(lane, x) & (NEL - 1u);
    const uint32_t pl = tile_swz<REAL>(il) * AB;
    const uint32_t g0 = (uint32_t)__popc(il & l0.zin) << 31, g1 = (uint32_t)__popc(il & l1.zin) << 31;
    const uint32_t c0rh = (uint32_t)__double2hiint(l0.cr) ^ g0, c0rl = (uint32_t)__double2loint(l0.cr);
    const uint32_t c1rh = (uint32_t)__double2hiint(l1.cr) ^ g1, c1rl = (uint32_t)__double2loint(l1.cr);
    const uint32_t c0ih = (uint32_t)__double2hiint(l0.ci) ^ g0, c0il = (uint32_t)__double2loint(l0.ci);
    const uint32_t c1ih = (uint32_t)__double2hiint(l1.ci) ^ g1, c1il = (uint32_t)__double2loint(l1.ci);
    const uint32_t sxb = tile_swz<REAL>(xf) * AB;
    const uint32_t ts = en.tsign;
    const char *base = reinterpret_cast<const char *>(tile);
    double part = 0.0;
    // groups of four trips, written out four times: the table entries are scalar registers by NAME (a loop over r would index the
    // table through memory)
    auto four = [&](auto rc) {
        constexpr uint32_t r = decltype(rc)::value, k = 256u * r;
        amp a[4], c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t off = pl ^ en.ph[4 * r + q];
            a[q] = *reinterpret_cast<const amp *>(base + off);
            c[q] = *reinterpret_cast<const amp *>(base + (off ^ sxb));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            constexpr uint32_t t0 = 4u * r;
            const uint32_t t = t0 + (uint32_t)q;
            // sign bit of trip t into the coefficient's high word: c ^ ((ts << (31 - t)) & 0x80000000) — one shift on the scalar unit,
            // one three-input bit operation on the vector unit
            const uint32_t m0 = (ts << (31u - t)) & 0x80000000u, m1 = (ts << (15u - t)) & 0x80000000u;
            double dr = __hiloint2double((int)(c0rh ^ m0), (int)c0rl), di = 0.0;
            if constexpr (!RO) di = __hiloint2double((int)(c0ih ^ m0), (int)c0il);
            if constexpr (!ONE) {
                dr += __hiloint2double((int)(c1rh ^ m1), (int)c1rl);
                if constexpr (!RO) di += __hiloint2double((int)(c1ih ^ m1), (int)c1il);
            }
            double v;
            if constexpr (REAL) {
                v = dr * (a[q] * c[q]);
            } else {
                const double wx = a[q].x * c[q].x + a[q].y * c[q].y;
                v = dr * wx;
                if constexpr (!RO) v -= di * (a[q].x * c[q].y - a[q].y * c[q].x);
            }
            if constexpr (FULL) part += v;
            else part += (k + 64u * q + lane < nk) ? v : 0.0;
        }
    };
    four(std::integral_constant<uint32_t, 0>{});
    if (nk > 256u) {
        four(std::integral_constant<uint32_t, 1>{});
        if (nk > 512u) {
            four(std::integral_constant<uint32_t, 2>{});
            if (nk > 768u) four(std::integral_constant<uint32_t, 3>{});
        }
    }
    return part;
}

template <int M, int NT, bool NTL, bool REAL>
__global__ __launch_bounds__(NT) void k_tile_expect(const void *__restrict__ st, uint64_t base, ExSweep sw,
                                                    const ExChunkT *__restrict__ chunks,
                                                    const ExEntryT *__restrict__ entries,
                                                    const ExTermT *__restrict__ terms,
                                                    const ExFlatT *__restrict__ flats,
                                                    const ExItemT *__restrict__ items, double2 *__restrict__ partials,
                                                    int accumulate, const ExChunkT *__restrict__ achunks,
                                                    const ExAGroupT *__restrict__ agroups,
                                                    const ExTermT *__restrict__ aterms, int sparse_den,
                                                    int *__restrict__ sparse_tiles = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Amp<REAL>::T amp;
    constexpr uint32_t NEL = 1u << M;
    constexpr uint32_t NELV = REAL ? NEL / 2 : NEL;  // 16-byte elements (REAL: pairs of amplitudes, see k_tile_sweep)
    constexpr int TRIPS = NELV / NT;
    constexpr int PP = TILE_ENTRY_PAIRS / 64;  // pairs per lane
    const amp *tile = reinterpret_cast<const amp *>(smem);
    double2 *tilev = reinterpret_cast<double2 *>(smem);
    // [tile][dense path: term table | sparse path: staged terms + pieces (same bytes)][reduction][scan][non-zero list]
    // sparse_den < 0: DENSE layout — the launch left the sparse path's staging area and non-zero list out (launch_tile_expect):
    // [tile][term table of one chunk][reduction]
    const bool dense_layout = sparse_den < 0;
    const bool skip_diag = sparse_den == -2;   // the diagonal group is evaluated by k_tile_diag (dense registers, run_expectation_tiled)
    ExTermLds *lt = reinterpret_cast<ExTermLds *>(smem + (size_t)NELV * sizeof(double2));
    ExTermLds *spt = lt;                                             // sparse path: staged terms ...
    ExAGroupT *spg = reinterpret_cast<ExAGroupT *>(spt + (dense_layout ? TILE_TERM_CAP : TILE_SPARSE_TERMS));  // ... and pieces of a pass
    double2 *red = reinterpret_cast<double2 *>(spg + (dense_layout ? 0 : TILE_SPARSE_GROUPS));
    int *scan = reinterpret_cast<int *>(red + NT / 64);             // NT / 64 wave totals + the tile's count
    uint16_t *nz = reinterpret_cast<uint16_t *>(scan + NT / 64 + 2);  // tile-local indices of the non-zero amplitudes
    static_assert(TILE_SPARSE_TERMS >= TILE_TERM_CAP, "the dense term table lives in the sparse staging bytes");
    const v2d *p = reinterpret_cast<const v2d *>(st);

    uint64_t tb = blockIdx.x;
    for (uint64_t mk = sw.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, sw.mask_lo);
    const uint64_t gbase = base | (REAL ? tb << 1 : tb);
    {
        v2d reg[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            const uint64_t g = tb | glow | spread_bits((uint32_t)j, sw.mask_hi);
            reg[j] = NTL ? __builtin_nontemporal_load(&p[g]) : p[g];
        }
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) tilev[tile_swz_v(threadIdx.x + j * NT)] = make_double2(reg[j].x, reg[j].y);
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    double acc = 0.0;
    __syncthreads();  // tile visible
    // ---- sparse tiles.  A UCC-type state lives on a particle-number / spin sector (24 qubits, 10 electrons: 3.7 % of the
    // amplitudes), so nearly every pair product of the entry walks below multiplies exact zeros.  The tile's non-zero
    // amplitudes are compacted (ascending index: thread t owns NEL / NT consecutive amplitudes, block-wide exclusive
    // scan — a fixed order, so the sum stays reproducible), and when they are few the sweep's x-groups are evaluated in
    // operator-application form over that list only: E += sum_{i non-zero} conj(a_i) D_g(i ^ x) a_{i ^ x}, one wave per
    // group, lanes over the list.  Skipped work is multiplication by exact zeros; dense tiles take the entry walks.
    if (sparse_den > 0) {
        constexpr int K = NEL / NT;
        uint32_t found[K];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t e = threadIdx.x * K + k;
            const amp a = tile[tile_swz<REAL>(e)];
            bool nzero;
            if constexpr (REAL) nzero = a != 0.0; else nzero = a.x != 0.0 || a.y != 0.0;
            found[k] = e;
            cnt += nzero ? 1 : 0;
            if (!nzero) found[k] = 0xffffffffu;
        }
        int incl = cnt;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += up;
        }
        if (lane == 63) scan[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) {
            const int c = scan[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if ((int64_t)total * sparse_den <= (int64_t)NEL) {
            if (sparse_tiles && threadIdx.x == 0 && blockIdx.y == 0) atomicAdd(sparse_tiles, 1);   // (the host's census of the first sweep)
            int pos = before + incl - cnt;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (found[k] != 0xffffffffu) nz[pos++] = (uint16_t)found[k];
            const int nnz = __builtin_amdgcn_readfirstlane(total);
            // the sweep's pieces are one contiguous range of the apply tables (host chunks of <= TILE_TERM_CAP terms /
            // TILE_APPLY_GROUPS pieces); up to two host chunks are staged in LDS per pass — term signs of the z bits outside
            // the tile folded in —, then every wave walks its pieces of the pass without further barriers
            for (int ch = sw.a0; ch < sw.a1; ch += 2) {
                const ExChunkT c0 = achunks[ch], c1 = achunks[min(ch + 1, sw.a1 - 1)];
                const int tb0 = c0.t0, tb1 = c1.t1, gb0 = c0.g0, gb1 = c1.g1;
                __syncthreads();  // list complete / previous pass's tables no longer read
                for (int t = tb0 + (int)threadIdx.x; t < tb1; t += NT) {
                    const ExTermT et = aterms[t];
                    const bool neg = parity64(gbase & et.zout);
                    ExTermLds l;
                    l.cr = neg ? -et.cr : et.cr;
                    l.ci = neg ? -et.ci : et.ci;
                    l.zin = et.zin;
                    l.pad = 0;
                    spt[t - tb0] = l;
                }
                for (int g = gb0 + (int)threadIdx.x; g < gb1; g += NT) {
                    ExAGroupT gr = agroups[g];
                    gr.t0 -= tb0;
                    gr.t1 -= tb0;
                    spg[g - gb0] = gr;
                }
                __syncthreads();
                const int gfirst = wave + (NT / 64) * (int)blockIdx.y, gstep = (NT / 64) * (int)gridDim.y, ng = gb1 - gb0;
                for (int lb = 0; lb < nnz; lb += 256) {  // up to four list entries per lane share one walk over the pieces
                    switch (min(4, (nnz - lb + 63) >> 6)) {
                    case 1: acc += tile_sparse_pieces<REAL, 1>(tile, nz, nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
                    case 2: acc += tile_sparse_pieces<REAL, 2>(tile, nz, nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
                    case 3: acc += tile_sparse_pieces<REAL, 3>(tile, nz, nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
                    default: acc += tile_sparse_pieces<REAL, 4>(tile, nz, nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
                    }
                }
            }
            __syncthreads();
            const double2 t = block_sum<NT>(make_double2(acc, 0.0), red);
            if (threadIdx.x == 0) {
                const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
                if (accumulate) {
                    const double2 o = partials[slot];
                    partials[slot] = make_double2(o.x + t.x, o.y);
                } else {
                    partials[slot] = t;
                }
            }
            return;
        }
    }
    for (int t = sw.i0 + (int)threadIdx.x + NT * (int)blockIdx.y; t < sw.i1; t += NT * (int)gridDim.y) {
        const ExItemT it = items[t];
        const ExFlatT fe = flats[it.entry];
        const bool s0 = parity64(gbase & fe.zout0), s1 = parity64(gbase & fe.zout1);
        const double c0r = s0 ? -fe.c0r : fe.c0r, c1r = s1 ? -fe.c1r : fe.c1r;
        const double c0i = s0 ? -fe.c0i : fe.c0i, c1i = s1 ? -fe.c1i : fe.c1i;
        const uint32_t x = fe.x, keep = ~fe.x;
        // (round 5: 69 -> ~50 vector instructions per pair — the kernel is bound by their issue, SQ counters in profiles/r5_tilexp.
        // The swizzle is linear over XOR, so the partner's LDS address is one XOR away from the own one; a term's sign is one XOR on
        // the high word of its coefficient, the parity shifted into the sign bit, instead of a compare and two selects.)
        const uint32_t sx = tile_swz<REAL>(x);
        const uint32_t c0rh = (uint32_t)__double2hiint(c0r), c0rl = (uint32_t)__double2loint(c0r);
        const uint32_t c1rh = (uint32_t)__double2hiint(c1r), c1rl = (uint32_t)__double2loint(c1r);
        const uint32_t c0ih = (uint32_t)__double2hiint(c0i), c0il = (uint32_t)__double2loint(c0i);
        const uint32_t c1ih = (uint32_t)__double2hiint(c1i), c1il = (uint32_t)__double2loint(c1i);
        uint32_t i = it.istart;
        double part = 0.0;
        for (uint32_t c = 0; c < it.count; c += 2) {
            const uint32_t i0 = i, j0 = i0 ^ x;
            const uint32_t i1 = ((((i0 | x) + 1u) & keep) | fe.ibits) & (NEL - 1u), j1 = i1 ^ x;
            i = ((((i1 | x) + 1u) & keep) | fe.ibits) & (NEL - 1u);
            const uint32_t p0 = tile_swz<REAL>(i0), p1 = tile_swz<REAL>(i1);
            const amp a0 = tile[p0], b0 = tile[p0 ^ sx];
            const amp a1 = tile[p1], b1 = tile[p1 ^ sx];
            const uint32_t n00 = (uint32_t)__popc(j0 & fe.zin0) << 31, n01 = (uint32_t)__popc(j0 & fe.zin1) << 31;
            const uint32_t n10 = (uint32_t)__popc(j1 & fe.zin0) << 31, n11 = (uint32_t)__popc(j1 & fe.zin1) << 31;
            const double d0 = __hiloint2double((int)(c0rh ^ n00), (int)c0rl) + __hiloint2double((int)(c1rh ^ n01), (int)c1rl);
            const double d1 = __hiloint2double((int)(c0rh ^ n10), (int)c0rl) + __hiloint2double((int)(c1rh ^ n11), (int)c1rl);
            if constexpr (REAL) {
                part += d0 * (a0 * b0) + d1 * (a1 * b1);
            } else {
                const double e0 = __hiloint2double((int)(c0ih ^ n00), (int)c0il) + __hiloint2double((int)(c1ih ^ n01), (int)c1il);
                const double e1 = __hiloint2double((int)(c0ih ^ n10), (int)c0il) + __hiloint2double((int)(c1ih ^ n11), (int)c1il);
                part += d0 * (a0.x * b0.x + a0.y * b0.y) - e0 * (a0.x * b0.y - a0.y * b0.x);
                part += d1 * (a1.x * b1.x + a1.y * b1.y) - e1 * (a1.x * b1.y - a1.y * b1.x);
            }
        }
        acc += x ? 2.0 * part : part;
    }
    for (int ch = sw.c0; ch < sw.c1; ++ch) {
        const ExChunkT ck = chunks[ch];
        __syncthreads();  // previous chunk's term table no longer read
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
        __syncthreads();
        // small registers have fewer tiles than the chip has CUs: gridDim.y workgroups share a tile's entries
        for (int g = ck.g0 + wave + (NT / 64) * (int)blockIdx.y; g < ck.g1; g += (NT / 64) * (int)gridDim.y) {
            const ExEntryT en = entries[g];
            if (skip_diag && en.x == 0u) continue;
            const ExTermLds *gt = lt + (en.t0 - ck.t0);
            const int nt = en.t1 - en.t0;
            double part = 0.0;
            if (nt <= 2) {
                // the common case after merging (a JW double excitation leaves ONE coefficient per active pattern): one pair per lane
                // per trip, four trips at a time; variants by what the entry does not need (uniform per entry): a second term, imaginary
                // coefficients, the bound check of a ragged last trip
                const ExTermLds l0 = gt[0], l1 = gt[nt - 1];
                const bool ro = REAL || en.real_only, full = (en.nk & 255) == 0;
#define OVQE_TEP(ONE_, RO_, FULL_) part = tile_entry_pairs<REAL, NEL, ONE_, RO_, FULL_>(tile, en, l0, l1, lane)
                if (nt == 1) {
                    if (ro) { if (full) OVQE_TEP(true, true, true); else OVQE_TEP(true, true, false); }
                    else { if (full) OVQE_TEP(true, false, true); else OVQE_TEP(true, false, false); }
                } else {
                    if (ro) { if (full) OVQE_TEP(false, true, true); else OVQE_TEP(false, true, false); }
                    else { if (full) OVQE_TEP(false, false, true); else OVQE_TEP(false, false, false); }
                }
#undef OVQE_TEP
            } else {
            double vx[PP], vy[PP], dr[PP], di[PP];
            uint32_t jj[PP];
#pragma unroll
            for (int m = 0; m < PP; ++m) {
                const uint32_t k = (uint32_t)en.k0 + lane + 64u * m;
                const bool live = lane + 64u * m < (uint32_t)en.nk;
                const uint32_t i = deposit_index(k, en.x) | en.ibits;
                jj[m] = i ^ en.x;
                const amp a = tile[tile_swz<REAL>(i & (NEL - 1u))], c = tile[tile_swz<REAL>(jj[m] & (NEL - 1u))];
                if constexpr (REAL) {
                    vx[m] = live ? a * c : 0.0;
                    vy[m] = 0.0;
                } else {
                    vx[m] = live ? a.x * c.x + a.y * c.y : 0.0;  // conj(a_i) a_j   (x = 0: |a_i|^2)
                    vy[m] = live ? a.x * c.y - a.y * c.x : 0.0;
                }
                dr[m] = 0.0;
                di[m] = 0.0;
            }
            if (REAL || en.real_only) {
                for (int t = 0; t < nt; ++t) {
                    const ExTermLds l = gt[t];
#pragma unroll
                    for (int m = 0; m < PP; ++m) dr[m] = fma(l.cr, parity_sign(jj[m] & l.zin), dr[m]);
                }
            } else {
                for (int t = 0; t < nt; ++t) {
                    const ExTermLds l = gt[t];
#pragma unroll
                    for (int m = 0; m < PP; ++m) {
                        const double sg = parity_sign(jj[m] & l.zin);
                        dr[m] = fma(l.cr, sg, dr[m]);
                        di[m] = fma(l.ci, sg, di[m]);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < PP; ++m) part += dr[m] * vx[m] - di[m] * vy[m];
            }
            acc += en.x ? 2.0 * part : part;
        }
    }
    __syncthreads();
    const double2 t = block_sum<NT>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) {
        const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        if (accumulate) {
            const double2 o = partials[slot];
            partials[slot] = make_double2(o.x + t.x, o.y);
        } else {
            partials[slot] = t;
        }
    }
}

// ---- the DIAGONAL group by a fast Walsh-Hadamard transform (round 6) ------------------------------------------------------------
// The x = 0 group of a Hermitian sum is D(i) |a_i|^2 with D(i) = sum_t c_t (-1)^{|i & z_t|}: hundreds of Z strings (a JW Hamiltonian's
// number operators and Coulomb terms; 30 % of the bench's random strings) evaluated for EVERY amplitude — as many term evaluations
// as all the other groups of the cover together.  On a contiguous tile of 2^M amplitudes i = (tile, e) the sign splits into a
// per-tile sign (z bits above the tile) and (-1)^{|e & zin|}: D over the tile is the Walsh-Hadamard transform of the sparse vector
// W[zin] = sum of the signed coefficients with that zin — M 2^M additions instead of T 2^M term evaluations (M = 12, T = 300: 25 x).
// One workgroup per tile: |a|^2 in registers, W in LDS (32 KB), unique zin values filled from a host-built CSR (deterministic
// order), M butterfly stages, dot product.  Launched for dense registers only (the census of run_expectation_tiled): there the
// sweeps of k_tile_expect skip their x = 0 entries (sparse_den = -2).
struct DiagTermT {
    uint64_t zout;   // z above the tile bits
    double c;
};
template <int M, int NT, bool NTL, bool REAL>
__global__ __launch_bounds__(NT) void k_tile_diag(const void *__restrict__ st, uint64_t base, const uint32_t *__restrict__ uzin,
                                                  const int32_t *__restrict__ uoff, const DiagTermT *__restrict__ dterms, int nu,
                                                  double2 *__restrict__ partials, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NEL = 1u << M;
    constexpr int K = NEL / NT;          // amplitudes per thread
    double *W = reinterpret_cast<double *>(smem);
    double2 *red = reinterpret_cast<double2 *>(W + NEL);
    const uint64_t gbase = base | ((uint64_t)blockIdx.x << M);
    double w2[K];
    bool any = false;
    if constexpr (REAL) {
        const v2d *p = reinterpret_cast<const v2d *>(st) + ((size_t)blockIdx.x << (M - 1));
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {     // pair v = tid + k NT: amplitudes 2 v, 2 v + 1
            const v2d r = NTL ? __builtin_nontemporal_load(&p[threadIdx.x + k * NT]) : p[threadIdx.x + k * NT];
            w2[2 * k] = r.x * r.x;
            w2[2 * k + 1] = r.y * r.y;
            any |= r.x != 0.0 || r.y != 0.0;
        }
    } else {
        const v2d *p = reinterpret_cast<const v2d *>(st) + ((size_t)blockIdx.x << M);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const v2d r = NTL ? __builtin_nontemporal_load(&p[threadIdx.x + k * NT]) : p[threadIdx.x + k * NT];
            w2[k] = r.x * r.x + r.y * r.y;
            any |= r.x != 0.0 || r.y != 0.0;
        }
    }
    if (!__syncthreads_or(any)) {          // a tile of zeros (a sector-sparse state on the dense path)
        if (threadIdx.x == 0 && !accumulate) partials[blockIdx.x] = make_double2(0.0, 0.0);
        return;
    }
    for (uint32_t e = threadIdx.x; e < NEL; e += NT) W[e] = 0.0;
    __syncthreads();
    for (int u = (int)threadIdx.x; u < nu; u += NT) {
        double w = 0.0;
        for (int j = uoff[u]; j < uoff[u + 1]; ++j) {
            const DiagTermT dt = dterms[j];
            w += parity64(gbase & dt.zout) ? -dt.c : dt.c;
        }
        W[uzin[u]] = w;
    }
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < M; ++b) {
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            const uint32_t q = threadIdx.x + (uint32_t)k * NT;
            const uint32_t low = (1u << b) - 1u;
            const uint32_t i = ((q & ~low) << 1) | (q & low), j = i | (1u << b);
            const double v = W[i], w = W[j];
            W[i] = v + w;
            W[j] = v - w;
        }
        __syncthreads();
    }
    double acc = 0.0;
    if constexpr (REAL) {
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            const uint32_t e = 2u * (threadIdx.x + (uint32_t)k * NT);
            acc += w2[2 * k] * W[e] + w2[2 * k + 1] * W[e + 1];
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) acc += w2[k] * W[threadIdx.x + (uint32_t)k * NT];
    }
    const double2 t = block_sum<NT>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) {
        if (accumulate) {
            const double2 o = partials[blockIdx.x];
            partials[blockIdx.x] = make_double2(o.x + t.x, o.y);
        } else {
            partials[blockIdx.x] = t;
        }
    }
}

// ---- <psi|H|psi> on the COMPACTED support (the tile cover above, fed from a compact state) -------------------------
// A state whose non-zero amplitudes are a small, parameter-independent subset S of the register (the particle-number /
// spin sector of a UCC-type ansatz: 3.7 % at 24 qubits / 10 electrons) is gathered once per evaluation into psic[k] =
// psi[sup[k]]; for every sweep of the cover the host keeps S sorted by tile — (tile-local index, compact id) per element,
// offsets per tile — so a workgroup builds its LDS tile from ~150 gathered amplitudes instead of reading 2^M of them from
// HBM, and evaluates the sweep's pieces over exactly that list (tile_sparse_pieces).  Same arithmetic as the sparse
// tiles of k_tile_expect, summation order fixed by the host's stable sort.  The support is found structurally-by-example
// (non-zeros of the state prepared at a generic parameter vector) and guarded at every evaluation: the compact state
// must carry the whole norm, otherwise the evaluation is redone on the dense cover (ovqe_sv.hip, compact cover).
template <bool REAL>
__global__ __launch_bounds__(256) void k_compact_gather(const typename Amp<REAL>::T *__restrict__ st,
                                                        const uint32_t *__restrict__ sup, uint32_t K,
                                                        typename Amp<REAL>::T *__restrict__ psic,
                                                        double2 *__restrict__ partials) {
    // psic[k] = psi[sup[k]] (ascending indices: one touch of every occupied line of the dense state) + the norm carried
    // by the support, per workgroup
    __shared__ double2 red[4];
    double acc = 0.0;
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < K; k += gridDim.x * 256u) {
        const typename Amp<REAL>::T a = st[sup[k]];
        psic[k] = a;
        if constexpr (REAL) acc += a * a; else acc += a.x * a.x + a.y * a.y;
    }
    const double2 t = block_sum<256>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// vals[s * K + pos] = psic[cid[s * K + pos]]: the compact state in the tile order of every sweep (coalesced writes, the
// source is cache-resident)
template <bool REAL>
__global__ __launch_bounds__(256) void k_compact_permute(const typename Amp<REAL>::T *__restrict__ psic,
                                                         const uint32_t *__restrict__ cid, uint64_t total,
                                                         typename Amp<REAL>::T *__restrict__ vals) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < total; k += (uint64_t)gridDim.x * 256u)
        vals[k] = psic[cid[k]];
}

template <int M, int NT, bool REAL>
__global__ __launch_bounds__(NT) void k_tile_expect_compact(const typename Amp<REAL>::T *__restrict__ vals,
                                                            const uint16_t *__restrict__ loc,
                                                            const uint32_t *__restrict__ off, uint64_t base,
                                                            const ExSweep *__restrict__ sweeps, uint32_t K,
                                                            const ExChunkT *__restrict__ achunks,
                                                            const ExAGroupT *__restrict__ agroups,
                                                            const ExTermT *__restrict__ aterms,
                                                            double2 *__restrict__ partials, int term_cap,
                                                            int group_cap, int chunks_per_pass) {
    // ONE launch for the whole cover: blockIdx.y = sweep, blockIdx.x = tile.  The compute-heavy sweeps (hundreds of
    // x-groups) and the latency-bound ones (a handful of groups per tile) share the chip instead of running back to back.
    const ExSweep sw = sweeps[blockIdx.y];
    vals += (size_t)blockIdx.y * K;
    loc += (size_t)blockIdx.y * K;
    off += (size_t)blockIdx.y * (gridDim.x + 1);
    constexpr int accumulate = 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Amp<REAL>::T amp;
    constexpr uint32_t NEL = 1u << M;
    amp *tile = reinterpret_cast<amp *>(smem);                                   // natural order (no bank swizzle)
    ExTermLds *spt = reinterpret_cast<ExTermLds *>(smem + (size_t)NEL * sizeof(amp));
    ExAGroupT *spg = reinterpret_cast<ExAGroupT *>(spt + term_cap);
    double2 *red = reinterpret_cast<double2 *>(spg + group_cap);
    uint16_t *nz = reinterpret_cast<uint16_t *>(red + NT / 64);
    const uint32_t n0 = off[blockIdx.x], nnz = off[blockIdx.x + 1] - n0;
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (nnz == 0) {   // no support in this tile (uniform): contributes nothing
        if (threadIdx.x == 0 && !accumulate) partials[slot] = make_double2(0.0, 0.0);
        return;
    }
    uint64_t tb = blockIdx.x;
    for (uint64_t mk = sw.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t gbase = base | (REAL ? tb << 1 : tb);
    {
        double2 *tz = reinterpret_cast<double2 *>(smem);
        constexpr uint32_t NV = NEL * sizeof(amp) / sizeof(double2);
        for (uint32_t v = threadIdx.x; v < NV; v += NT) tz[v] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nnz; k += NT) {
        const uint32_t l = loc[n0 + k];
        tile[l] = vals[n0 + k];
        nz[k] = (uint16_t)l;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u;
    double acc = 0.0;
    for (int ch = sw.a0; ch < sw.a1; ch += chunks_per_pass) {
        const ExChunkT c0 = achunks[ch], c1 = achunks[min(ch + chunks_per_pass, (int)sw.a1) - 1];
        const int tb0 = c0.t0, tb1 = c1.t1, gb0 = c0.g0, gb1 = c1.g1;
        __syncthreads();  // tile + list complete / previous pass's tables no longer read
        for (int t = tb0 + (int)threadIdx.x; t < tb1; t += NT) {
            const ExTermT et = aterms[t];
            const bool neg = parity64(gbase & et.zout);
            ExTermLds l;
            l.cr = neg ? -et.cr : et.cr;
            l.ci = neg ? -et.ci : et.ci;
            l.zin = et.zin;
            l.pad = 0;
            spt[t - tb0] = l;
        }
        for (int g = gb0 + (int)threadIdx.x; g < gb1; g += NT) {
            ExAGroupT gr = agroups[g];
            gr.t0 -= tb0;
            gr.t1 -= tb0;
            spg[g - gb0] = gr;
        }
        __syncthreads();
        const int gfirst = wave, gstep = NT / 64, ng = gb1 - gb0;
        for (int lb = 0; lb < (int)nnz; lb += 256) {
            switch (min(4, ((int)nnz - lb + 63) >> 6)) {
            case 1: acc += tile_sparse_pieces<REAL, 1, false>(tile, nz, (int)nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
            case 2: acc += tile_sparse_pieces<REAL, 2, false>(tile, nz, (int)nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
            case 3: acc += tile_sparse_pieces<REAL, 3, false>(tile, nz, (int)nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
            default: acc += tile_sparse_pieces<REAL, 4, false>(tile, nz, (int)nnz, lb, lane, spg, gfirst, ng, gstep, spt); break;
            }
        }
    }
    __syncthreads();
    const double2 t = block_sum<NT>(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) {
        if (accumulate) {
            const double2 o = partials[slot];
            partials[slot] = make_double2(o.x + t.x, o.y);
        } else {
            partials[slot] = t;
        }
    }
}

// ---- the non-empty tiles of every sweep, for a state given as the list of its non-zero amplitudes ----------------------
// Block s: bitmap (LDS) of the tiles of sweep s that hold a listed index, then their numbers into lists[s * cap ..) and
// counts[s].  A tile's number is its index bits outside the sweep's tile set, packed in ascending order (the blockIdx -> tile
// map of the tile kernels).
__global__ __launch_bounds__(256) void k_tile_lists(const uint64_t *__restrict__ idx, uint64_t count,
                                                    const uint64_t *__restrict__ smasks, int nbits, uint32_t ntiles, uint32_t cap,
                                                    uint32_t *__restrict__ lists, uint32_t *__restrict__ counts) {
    extern __shared__ uint32_t tl_bitmap[];
    __shared__ uint32_t found;
    const uint32_t words = (ntiles + 31u) >> 5;
    const uint64_t sm = smasks[blockIdx.x];
    for (uint32_t w = threadIdx.x; w < words; w += 256) tl_bitmap[w] = 0u;
    if (threadIdx.x == 0) found = 0u;
    __syncthreads();
    for (uint64_t e = threadIdx.x; e < count; e += 256) {
        const uint64_t i = idx[e];
        uint32_t t = 0u;
        int pos = 0;
        for (int b = 0; b < nbits; ++b)
            if (!((sm >> b) & 1ull)) t |= (uint32_t)((i >> b) & 1ull) << pos++;
        atomicOr(&tl_bitmap[t >> 5], 1u << (t & 31u));
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < words; w += 256) {
        uint32_t v = tl_bitmap[w];
        while (v) {
            const uint32_t b = (uint32_t)__ffs((int)v) - 1u;
            v &= v - 1u;
            const uint32_t pos = atomicAdd(&found, 1u);
            if (pos < cap) lists[(size_t)blockIdx.x * cap + pos] = (w << 5) + b;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = min(found, cap);
}

// ---- out = ident * in + H in on tiles ------------------------------------------------------------------------
// Same cover as k_tile_expect, operator-application form (sigma = H psi of the ADAPT screens and of the adjoint
// gradient, the Lanczos matrix-vector product): a thread OWNS its output amplitudes (registers), walks the sweep's
// x-groups and adds D_g(j) in_j, j = e ^ x, from the LDS copy of the input tile — no atomics, fixed summation order.
// The output is accumulated across the sweeps (first sweep: out = ident * in + ..., later: out += ...), so a sweep
// moves 48 bytes per amplitude where the gather kernel k_apply_sum re-reads the input once per x-group.  VALU-bound
// (every term of every group is evaluated for every amplitude): about twice as fast as the gather kernel at 24 qubits.

template <int M, int NT, bool NTL>
__global__ __launch_bounds__(NT) void k_tile_apply(const amp_t *__restrict__ in, amp_t *__restrict__ out, uint64_t base,
                                                   ExSweep sw, const ExChunkT *__restrict__ chunks,
                                                   const ExAGroupT *__restrict__ groups,
                                                   const ExTermT *__restrict__ terms, int first, double ident,
                                                   const uint32_t *__restrict__ tile_list,
                                                   const uint32_t *__restrict__ tile_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t NEL = 1u << M;
    constexpr int TRIPS = NEL / NT;
    double2 *tile = reinterpret_cast<double2 *>(smem);
    ExTermLds *lt = reinterpret_cast<ExTermLds *>(smem + (size_t)NEL * sizeof(double2));
    ExAGroupT *lg = reinterpret_cast<ExAGroupT *>(lt + TILE_TERM_CAP);
    const v2d *p = reinterpret_cast<const v2d *>(in);
    v2d *q = reinterpret_cast<v2d *>(out);

    uint64_t tb = blockIdx.x;
    if (tile_list) {   // only the tiles of this sweep that hold a non-zero input amplitude (k_tile_lists); out was zeroed
        if (blockIdx.x >= *tile_count) return;
        tb = tile_list[blockIdx.x];
    }
    for (uint64_t mk = sw.smask; mk; mk &= mk - 1ull) tb = insert_zero(tb, __ffsll((long long)mk) - 1);
    const uint64_t glow = spread_bits(threadIdx.x, sw.mask_lo);
    const uint64_t gbase = base | tb;
    double2 self[TRIPS], acc[TRIPS];
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
        const uint64_t g = tb | glow | spread_bits((uint32_t)j, sw.mask_hi);
        const v2d r = NTL ? __builtin_nontemporal_load(&p[g]) : p[g];
        self[j] = make_double2(r.x, r.y);
        acc[j] = make_double2(0.0, 0.0);
        tile[tile_swz_v(threadIdx.x + j * NT)] = self[j];
    }
    {
        // a tile of zeros contributes nothing (the ADAPT state of a few operators, a UCC state's particle-number sector:
        // most tiles of the register): the sweep then costs this tile its read only
        bool any = false;
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) any |= self[j].x != 0.0 || self[j].y != 0.0;
        if (!__syncthreads_or(any)) {
            if (first) {
#pragma unroll
                for (int j = 0; j < TRIPS; ++j) {
                    v2d r;
                    r.x = 0.0;
                    r.y = 0.0;
                    q[tb | glow | spread_bits((uint32_t)j, sw.mask_hi)] = r;
                }
            }
            return;
        }
    }
    for (int ch = sw.a0; ch < sw.a1; ++ch) {
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
                je[j] = (threadIdx.x + j * NT) ^ xl;
                k[j] = tile[tile_swz_v(je[j])];
                dr[j] = 0.0;
                di[j] = 0.0;
            }
            if (__builtin_amdgcn_readfirstlane(gr.pad) & 1) {   // real coefficients only (every group of a real-symmetric H)
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
                    // +-1.0 from the parity bit (3 integer ops), then one exact FMA per component: half the VALU work of
                    // compare + select + add
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
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
        const uint64_t g = tb | glow | spread_bits((uint32_t)j, sw.mask_hi);
        v2d r;
        if (first) {
            r.x = ident * self[j].x + acc[j].x;
            r.y = ident * self[j].y + acc[j].y;
        } else {
            const v2d o = q[g];
            r.x = o.x + acc[j].x;
            r.y = o.y + acc[j].y;
        }
        q[g] = r;
    }
}

}  // namespace ovqe
