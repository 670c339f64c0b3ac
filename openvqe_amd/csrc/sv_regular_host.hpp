// sv_regular_host.hpp — host side of the sector path on REGULAR supports (full cosets of a program's Z2 symmetries): the symmetry
// analysis and the per-sweep tables of k_sector_sweep_reg / k_sector_adjoint_reg (sv_sector.hpp).  Plain C++17, no HIP: included by
// sv_sector.hpp, and compiled on its own under AddressSanitizer + UBSan by tests/cpu/regular_tables_check.cpp, which replays a
// sweep from these tables on the host and compares it with the pair-by-pair definition (tests/test_sanitizer.py).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace ovqe {

struct SecBuildOp {   // one compact op = one OP_TAB op, or one rotation of an OP_PAIR run
    uint64_t x;       // mixing mask
    uint64_t zs;      // sign = parity(i & zs) ^ flip, i = the pair's member that matches the pattern
    int32_t pat0, npat;
    int32_t flip, tab0;  // rotation table entries tab0 + pattern
};
struct SecPat {
    uint64_t pm, pv;  // i is the first member of an active pair when (i & pm) == pv
};

struct SecRegOp {         // 32 dwords; the first 16 are what the sweep kernel reads (scalar loads, uniform per op), the host fills everything
    uint32_t w_nsel;      // w | nsel << 16
    uint32_t zt;          // sign mask in tile-number space
    uint32_t sel_t[2];    // selector masks in tile-number space
    uint32_t dep[8];      // 16-bit halves: BYTE offset (swizzled) of member e of a group, e = pattern over the mixing bits (ascending)
    uint32_t pad[4];
    // host side (the group words carry their effect)
    uint32_t xs;          // kept mixing bits, slot space (w of them)
    uint32_t zin;         // sign mask, slot space (outside xs)
    uint32_t sel_in[2];   // selector masks, slot space
    uint32_t tab;         // first entry of the op in the sweep's (c, s) table (= 8 x its number)
    uint32_t gpos[2];     // nibble k = slot position of bit k of the group number (the m - w positions outside xs; see below)
    uint32_t pad2[9];
};
struct SecRegHead {       // the kernel's view of a record
    uint32_t w_nsel, zt, sel_t[2], dep[8];
    uint32_t pad[4];      // a block of two ops (w_nsel bit 24): [0] = whether own B sits on A's sign / selector mask (bits 0, 1), own A on B's
                          // (bits 2, 3); [1], [2] = zt, sel_t[0] of op B
};
static_assert(sizeof(SecRegOp) == 128, "SecRegOp is read as 32 dwords");
// LDS bank swizzle of a tile of doubles, linear over XOR (swz(a ^ b) = swz(a) ^ swz(b)): a group's members are base ^ spread(e).
// The bank of an 8-byte slot is its low five address bits (32 slots per LDS cycle for reads, 16 for writes).  Slot bit p < 5 IS bank
// bit p; slot bit p >= 5 adds the 5-bit COLUMN SEC_REG_SWZ_COL[p - 5] to the bank bits.  The 32 lanes of a read group reach 32 different
// banks iff the columns of the five slot positions that carry the lane bits are linearly independent over GF(2), and the 16 lanes of a
// write group 16 different slots of the 16 that one write cycle serves iff the columns of the first four are independent in their low four
// bits; the host picks those positions per op among the positions the op does not mix (build_reg_ops: group_order).  Round 4's swizzle used the columns
// e_(p mod 5): an op that mixes both members of a residue class of two (positions {2,7}, {3,8}, {4,9}) left no independent five — 17 % of the
// ops of the reference's QUCCSD list ran with two-way conflicts on every access (SQ_LDS_BANK_CONFLICT: 30 k cycles per CU and sweep,
// profiles/r5_quccsd24).  These columns were searched (4000 random draws) for the fewest position sets whose complement fails either test: of
// the 495 ways to mix four of twelve positions ONE does (round 4: 150), of the 220 three-position sets none (32).
constexpr uint32_t SEC_REG_SWZ_COL[8] = {28u, 14u, 11u, 5u, 19u, 31u, 21u, 7u};   // slot positions 5 .. 12
constexpr inline uint32_t sec_reg_bank_column(int p) { return p < 5 ? (1u << p) : SEC_REG_SWZ_COL[p - 5]; }
constexpr inline uint32_t sec_reg_swz(uint32_t v) {
    uint32_t low = v & 31u;
    for (int k = 0; k < 8; ++k)
        if ((v >> (5 + k)) & 1u) low ^= SEC_REG_SWZ_COL[k];
    return (v & ~31u) | low;
}


constexpr uint32_t SEC_REG_GSTRIDE = 2048;   // words per op: the groups of a one-bit op in a 4096-slot tile
constexpr uint32_t SEC_REG_TSTRIDE = 8;

// ---- regular supports: the Z2 symmetries of a program and the per-sweep tables of k_sector_sweep_reg ---------------------------
// Z-type operators Z^g that commute with every rotation string = the GF(2) orthogonal complement of the x masks.  Returned in
// the form the sweeps need: one generator per FREE bit f (a column without pivot in the reduced row echelon form of the x
// masks, pivots = highest bits), g_f = {f} + G_f with G_f on pivot ("kept") bits above f only.
inline void z2_symmetries(const std::vector<uint64_t> &xs, int n, std::vector<int> &freebits, std::vector<uint64_t> &G) {
    uint64_t basis[64] = {};
    for (uint64_t v : xs)
        while (v) {
            const int p = 63 - __builtin_clzll(v);
            if (!basis[p]) {
                basis[p] = v;
                break;
            }
            v ^= basis[p];
        }
    for (int p = 0; p < n; ++p)
        if (basis[p])
            for (int q = p + 1; q < n; ++q)
                if ((basis[q] >> p) & 1ull) basis[q] ^= basis[p];
    freebits.clear();
    G.clear();
    for (int f = 0; f < n; ++f) {
        if (basis[f]) continue;
        uint64_t g = 0;
        for (int p = f + 1; p < n; ++p)
            if (basis[p] && ((basis[p] >> f) & 1ull)) g |= 1ull << p;
        freebits.push_back(f);
        G.push_back(g);
    }
}
// The same symmetries seen from ONE sweep (tile bit set S): generators recombined so that each has a dependent bit of its own
// INSIDE S — the lowest S-bit of its row after elimination from the low end — and nowhere else; the members of a tile, sorted by
// their inside bits, are then numbered by the other ("kept") inside bits.  false: the generators are dependent on S (some tiles
// would be empty, others larger: not the regular layout).
inline bool sweep_symmetries(const std::vector<int> &freebits, const std::vector<uint64_t> &G, uint64_t S, int n,
                             std::vector<int> &dep, std::vector<uint64_t> &Gs) {
    std::vector<uint64_t> rows;
    for (size_t k = 0; k < freebits.size(); ++k) rows.push_back(G[k] | (1ull << freebits[k]));
    std::vector<int> pivot(rows.size(), -1);
    for (int p = 0; p < n; ++p) {
        if (!((S >> p) & 1ull)) continue;
        size_t r = rows.size();
        for (size_t k = 0; k < rows.size() && r == rows.size(); ++k)
            if (pivot[k] < 0 && ((rows[k] >> p) & 1ull)) r = k;
        if (r == rows.size()) continue;
        pivot[r] = p;
        for (size_t k = 0; k < rows.size(); ++k)
            if (k != r && ((rows[k] >> p) & 1ull)) rows[k] ^= rows[r];
    }
    dep.clear();
    Gs.clear();
    for (size_t k = 0; k < rows.size(); ++k) {
        if (pivot[k] < 0) return false;
        dep.push_back(pivot[k]);
        Gs.push_back(rows[k] & ~(1ull << pivot[k]));
    }
    return true;
}
inline uint32_t host_pext(uint64_t v, uint64_t mask) {
    uint32_t r = 0;
    int k = 0;
    for (; mask; mask &= mask - 1ull, ++k) r |= (uint32_t)((v >> __builtin_ctzll(mask)) & 1ull) << k;
    return r;
}
// the SecRegOp list, the table-entry map and the group words of one sweep (tile bit set S, its dependent bits `freebits` with their
// generators G); false: an op outside the kernel's shapes (more than 4 kept mixing bits, more than 2 dependent bits inside x).
// pairs: two consecutive ops with three kept mixing bits each, two of them shared, and the same dependent bits in x become one
// BLOCK — the 16 slots over (shared, own A, own B) in registers, both ops between one read and one write of them.
//
// RUNS WITHOUT BARRIERS (round 5; wave_bits > 0): the groups of an op are numbered so that bits lane_bits .. lane_bits + wave_bits - 1
// of the group number — the WAVE that takes the group in a workgroup of 2^(lane_bits + wave_bits) threads — sit on slot positions
// that a whole run of consecutive ops does not mix.  Every op of the run then keeps each wave inside its own 2^(m - wave_bits) slots:
// what a wave reads was written by itself, the LDS pipeline keeps a wave's accesses in order, and the barrier between two ops of a
// run is not needed at all — the waves drift apart and one wave's rotations run under another's LDS traffic.  A run ends where the
// next op would leave fewer than wave_bits unmixed positions (or after run_cap units); the op (block: its first op) BEFORE which no
// barrier is needed... is marked on its predecessor: w_nsel bit 27 of a unit's first op = "no barrier after this unit".
inline bool build_reg_ops(const std::vector<SecBuildOp> &sops, const std::vector<SecPat> &spats, uint64_t S, uint64_t all, uint64_t hf,
                          const std::vector<int> &freebits, const std::vector<uint64_t> &G, int mbits, bool pairs, std::vector<SecRegOp> &out,
                          std::vector<uint32_t> &emap, std::vector<uint32_t> &gwords, int lane_bits = 6, int wave_bits = 0, int run_cap = 64) {
    uint64_t freemask = 0;
    for (int f : freebits) freemask |= 1ull << f;
    const uint64_t kept_in = S & ~freemask, outside = all & ~S;
    auto par = [](uint64_t v) { return (uint32_t)(__builtin_popcountll(v) & 1); };
    if (mbits > 13) return false;
    out.clear();
    emap.clear();
    gwords.clear();
    const size_t n = sops.size();
    struct Aux {
        uint64_t xk = 0;
        int w = 0;
        std::vector<size_t> sel_f, sign_f;
    };
    std::vector<Aux> aux(n);
    out.assign(n, SecRegOp{});
    // pass 1: masks of every op
    for (size_t i = 0; i < n; ++i) {
        const SecBuildOp &b = sops[i];
        Aux &a = aux[i];
        SecRegOp &r = out[i];
        a.xk = b.x & ~freemask;
        a.w = __builtin_popcountll(a.xk);
        if (a.w < 1 || a.w > 4 || (b.x & ~S)) return false;
        r.xs = host_pext(a.xk, kept_in);
        // sign of a pair = parity(first member & zs) ^ flip: the kept bits outside x go into the group's masks (dependent bits outside x
        // bring the kept part of their generator along), everything on x and every constant into the table entries
        uint64_t zeff = b.zs;
        for (size_t k = 0; k < freebits.size(); ++k) {
            const uint64_t fb = 1ull << freebits[k];
            if (b.x & fb) a.sel_f.push_back(k);
            else if (b.zs & fb) {
                a.sign_f.push_back(k);
                zeff ^= G[k];
            }
        }
        zeff &= ~b.x & ~freemask;
        if (a.sel_f.size() > 2 || a.w + (int)a.sel_f.size() > 4) return false;
        r.w_nsel = (uint32_t)a.w | ((uint32_t)a.sel_f.size() << 16);
        r.zin = host_pext(zeff & S, kept_in);
        r.zt = host_pext(zeff & ~S, outside);
        for (size_t q = 0; q < a.sel_f.size(); ++q) {
            const uint64_t go = G[a.sel_f[q]] & ~b.x;
            r.sel_in[q] = host_pext(go & S, kept_in);
            r.sel_t[q] = host_pext(go & ~S, outside);
        }
    }
    // pass 2: blocks
    std::vector<char> second(n, 0), first(n, 0);
    for (size_t i = 0; pairs && i + 1 < n; ++i) {
        const Aux &a = aux[i], &c = aux[i + 1];
        if (a.w == 3 && c.w == 3 && a.sel_f == c.sel_f && a.sel_f.size() <= 1 && __builtin_popcountll(a.xk | c.xk) == 4 && mbits >= 4) {
            first[i] = second[i + 1] = 1;
            ++i;
        }
    }
    // numbering of the groups: bits 0..4 of the group number (the lanes of a read group) on five slot positions whose bank columns are
    // linearly independent (sec_reg_swz), lowest first; `wave` (a run without barriers): those positions, ascending, become bits
    // lane_bits.. of the group number
    auto reduce_by = [](const uint32_t (&basis)[5], uint32_t v) {
        for (int b = 4; b >= 0; --b)
            if (basis[b] && ((v >> b) & 1u)) v ^= basis[b];
        return v;
    };
    auto group_order = [&](uint32_t mixing, uint32_t wave) {
        std::vector<int> free_pos, order;
        for (int p = 0; p < mbits; ++p)
            if (!(((mixing | wave) >> p) & 1u)) free_pos.push_back(p);
        std::vector<char> used(free_pos.size(), 0);
        uint32_t basis4[5] = {}, basis[5] = {};   // basis[b]: a vector whose highest set bit is b
        // lane bits 0..3 (a write group): columns independent in their low four bits
        for (size_t k = 0; k < free_pos.size() && order.size() < 4; ++k) {
            const uint32_t v = reduce_by(basis4, sec_reg_bank_column(free_pos[k]) & 15u);
            if (!v) continue;
            basis4[31 - __builtin_clz(v)] = v;
            const uint32_t f = reduce_by(basis, sec_reg_bank_column(free_pos[k]));   // (independent in four bits => independent in five)
            basis[31 - __builtin_clz(f)] = f;
            used[k] = 1;
            order.push_back(free_pos[k]);
        }
        // lane bit 4 (the other half of a read group): a column outside the span of those four
        for (size_t k = 0; k < free_pos.size() && order.size() == 4; ++k) {
            if (used[k]) continue;
            const uint32_t v = reduce_by(basis, sec_reg_bank_column(free_pos[k]));
            if (!v) continue;
            basis[31 - __builtin_clz(v)] = v;
            used[k] = 1;
            order.push_back(free_pos[k]);
        }
        // (fewer than five independent columns among the free positions: the remaining lane bits take the lowest positions left — a
        // two-way conflict per missing dimension)
        for (size_t k = 0; k < free_pos.size(); ++k)
            if (!used[k]) order.push_back(free_pos[k]);
        if (wave) {
            std::vector<int> wp;
            for (uint32_t m = wave; m; m &= m - 1u) wp.push_back(__builtin_ctz(m));
            order.insert(order.begin() + lane_bits, wp.begin(), wp.end());
        }
        return order;
    };
    // dimensions of the bank space that the read lanes of an op would miss (each costs a two-way bank conflict on its accesses)
    auto missing_residues = [&](uint32_t mixing, uint32_t wave) {
        uint32_t basis[5] = {}, basis4[5] = {};
        int rank = 0, rank4 = 0, nfree = 0;
        for (int p = 0; p < mbits; ++p) {
            if (((mixing | wave) >> p) & 1u) continue;
            ++nfree;
            const uint32_t v = reduce_by(basis, sec_reg_bank_column(p));
            if (v) {
                basis[31 - __builtin_clz(v)] = v;
                ++rank;
            }
            const uint32_t v4 = reduce_by(basis4, sec_reg_bank_column(p) & 15u);
            if (v4) {
                basis4[31 - __builtin_clz(v4)] = v4;
                ++rank4;
            }
        }
        return nfree >= 5 ? (5 - rank) + (4 - rank4) : 0;   // (fewer than five free positions: the groups do not fill a read cycle anyway)
    };
    // units (a block of two ops or one op), their mixing positions, and the runs that need no barrier inside
    std::vector<uint32_t> wave_of(n, 0u);
    if (wave_bits > 0) {
        struct Unit {
            size_t i;
            uint32_t mixing;
            int nbits;
        };
        std::vector<Unit> units;
        bool feasible = true;
        for (size_t i = 0; i < n; ++i) {
            if (second[i]) continue;
            Unit u{i, first[i] ? (out[i].xs | out[i + 1].xs) : out[i].xs, first[i] ? 4 : aux[i].w};
            feasible = feasible && mbits - u.nbits >= lane_bits + wave_bits;
            units.push_back(u);
        }
        const uint32_t allpos = (1u << mbits) - 1u;
        auto close_run = [&](size_t u0, size_t u1, uint32_t avail) {
            if (u1 <= u0) return;
            uint32_t best = 0;
            int best_score = 1 << 30;
            // every wave_bits-subset of the unmixed positions (Gosper's walk over the masks of that weight, filtered to `avail`)
            for (uint32_t sub = (1u << wave_bits) - 1u; sub <= allpos;) {
                const uint32_t cur = sub;
                {   // next mask of the same weight
                    const uint32_t c = sub & (0u - sub), r = sub + c;
                    sub = r ? (((r ^ sub) >> 2) / c) | r : allpos + 1u;
                }
                if (cur & ~avail) continue;
                const uint32_t sub_mask = cur;
                int score = 0;
                for (size_t u = u0; u < u1; ++u) score += missing_residues(units[u].mixing, sub_mask) - missing_residues(units[u].mixing, 0u);
                if (score < best_score || (score == best_score && sub_mask > best)) {
                    best_score = score;
                    best = sub_mask;
                }
            }
            for (size_t u = u0; u < u1; ++u) {
                wave_of[units[u].i] = best;
                if (first[units[u].i]) wave_of[units[u].i + 1] = best;
                if (u + 1 < u1) out[units[u].i].w_nsel |= 1u << 27;   // the next unit of the run follows without a barrier
            }
        };
        if (feasible) {
            size_t u0 = 0;
            uint32_t avail = allpos;
            for (size_t u = 0; u < units.size(); ++u) {
                const uint32_t na = avail & ~units[u].mixing;
                if (u > u0 && (__builtin_popcount(na) < wave_bits || (int)(u - u0) >= run_cap)) {
                    close_run(u0, u, avail);
                    u0 = u;
                    avail = allpos & ~units[u].mixing;
                } else {
                    avail = na;
                }
            }
            close_run(u0, units.size(), avail);
        }
    }
    for (size_t i = 0; i < n; ++i) {
        const SecBuildOp &b = sops[i];
        const Aux &a = aux[i];
        SecRegOp &r = out[i];
        const int w = a.w, nsel = (int)a.sel_f.size();
        // mixing positions in the order the table's patterns and a group's members follow: ascending — a block: shared, own A, own B
        std::vector<int> xp;       // slot positions: members of the group / block
        std::vector<int> xpos;     // index positions of this op's kept mixing bits, its pattern order (the last one is 0 in member A)
        uint32_t mixing = r.xs;
        if (first[i] || second[i]) {
            const size_t ia = first[i] ? i : i - 1, ib = ia + 1;
            const uint64_t sh = aux[ia].xk & aux[ib].xk, own = a.xk & ~sh;
            for (uint64_t m = sh; m; m &= m - 1ull) xpos.push_back(__builtin_ctzll(m));
            xpos.push_back(__builtin_ctzll(own));
            if (first[i]) {
                const uint64_t ownb = aux[ib].xk & ~sh;
                mixing = out[ia].xs | out[ib].xs;
                auto slotpos = [&](uint64_t bit) { return (int)__builtin_popcountll(kept_in & (bit - 1ull)); };
                for (uint64_t m = sh; m; m &= m - 1ull) xp.push_back(slotpos(m & (0ull - m)));
                xp.push_back(slotpos(own));
                xp.push_back(slotpos(ownb));
                const SecRegOp &rb = out[ib];
                const uint32_t pa = 1u << xp[2], pb = 1u << xp[3];
                r.w_nsel |= 1u << 24;
                if (r.sel_in[0] & pb) r.w_nsel |= 1u << 25;      // own B moves A's selector: A's second half takes the other variant's entries
                if (rb.sel_in[0] & pa) r.w_nsel |= 1u << 26;     // own A moves B's
                r.pad[0] = ((rb.zin & pa) ? 4u : 0u) | ((rb.sel_in[0] & pa) ? 8u : 0u) | ((r.zin & pb) ? 1u : 0u) | ((r.sel_in[0] & pb) ? 2u : 0u);
                r.pad[1] = rb.zt;
                r.pad[2] = rb.sel_t[0];
            }
        } else {
            for (uint64_t m = a.xk; m; m &= m - 1ull) xpos.push_back(__builtin_ctzll(m));
        }
        if (xp.empty())
            for (uint32_t m = r.xs; m; m &= m - 1u) xp.push_back(__builtin_ctz(m));
        const std::vector<int> order = group_order(mixing, wave_of[i]);
        for (size_t k = 0; k < order.size() && k < 16; ++k) r.gpos[k >> 3] |= (uint32_t)order[k] << (4 * (k & 7));
        // swizzled BYTE offsets of the members: pattern e spread over the positions of xp
        for (int e = 0; e < (1 << xp.size()); ++e) {
            uint32_t d = 0;
            for (size_t k = 0; k < xp.size(); ++k)
                if (e & (1 << k)) d |= 1u << xp[k];
            r.dep[e >> 1] |= ((sec_reg_swz(d) << 3) & 0xffffu) << (16 * (e & 1));
        }
        // group words: byte address of the swizzled base slot, parities on the sign / selector masks (a block: of both ops)
        const size_t g0 = gwords.size();
        gwords.resize(g0 + SEC_REG_GSTRIDE, 0u);
        if ((1u << order.size()) > SEC_REG_GSTRIDE) return false;
        const SecRegOp *rb = first[i] ? &out[i + 1] : nullptr;
        for (uint32_t g = 0; g < (1u << order.size()); ++g) {
            uint32_t bslot = 0;
            for (size_t k = 0; k < order.size(); ++k) bslot |= ((g >> k) & 1u) << order[k];
            uint32_t wd = (sec_reg_swz(bslot) << 3) | ((uint32_t)(__builtin_popcount(bslot & r.zin) & 1) << 16) |
                          ((uint32_t)(__builtin_popcount(bslot & r.sel_in[0]) & 1) << 17);
            if (rb) wd |= ((uint32_t)(__builtin_popcount(bslot & rb->zin) & 1) << 18) | ((uint32_t)(__builtin_popcount(bslot & rb->sel_in[0]) & 1) << 19);
            else wd |= (uint32_t)(__builtin_popcount(bslot & r.sel_in[1]) & 1) << 18;
            gwords[g0 + g] = wd;
        }
        r.tab = (uint32_t)emap.size();
        const int np = 1 << (w - 1);
        for (uint32_t sigma = 0; sigma < (1u << nsel); ++sigma)
            for (int q = 0; q < np; ++q) {
                uint64_t A = 0;   // member A of the pair (kept mixing bits = q in pattern order, the last of them 0), on the x positions
                for (int k = 0; k < w; ++k)
                    if (q & (1 << k)) A |= 1ull << xpos[k];
                for (size_t k = 0; k < a.sel_f.size(); ++k) {
                    const size_t fk = a.sel_f[k];
                    const int f = freebits[fk];
                    const uint32_t s_f = (uint32_t)((hf >> f) & 1ull) ^ par(hf & G[fk]);
                    if (s_f ^ par(A & G[fk]) ^ ((sigma >> k) & 1u)) A |= 1ull << f;
                }
                const uint64_t B = A ^ b.x;
                uint32_t entry = 0xffffffffu;
                for (int p = 0; p < b.npat; ++p) {
                    const SecPat &pt = spats[(size_t)b.pat0 + p];
                    const bool a_first = (A & pt.pm) == pt.pv, b_first = (A & pt.pm) == (pt.pv ^ (b.x & pt.pm));
                    if (!a_first && !b_first) continue;
                    const uint64_t F = a_first ? A : B;
                    uint32_t sign = par(F & b.zs) ^ (uint32_t)(b.flip & 1);
                    for (size_t fk : a.sign_f) sign ^= (uint32_t)((hf >> freebits[fk]) & 1ull) ^ par(hf & G[fk]) ^ par(F & G[fk]);
                    if (!a_first) sign ^= 1u;   // roles swapped: the table rotates (A, B) as (u, v)
                    entry = (uint32_t)(b.tab0 + p) | (sign << 31);
                    break;
                }
                emap.push_back(entry);
            }
        emap.resize((size_t)r.tab + SEC_REG_TSTRIDE, 0xffffffffu);   // 8 entries per op (2^nsel variants of 2^(w-1) pairs)
    }
    return true;
}



// ---- per-wave streams of the irregular supports' circuit sweeps (k_sector_sweep3, sv_sector.hpp): the host side of their plan ----
struct SecWaveRun {
    int32_t op_end;       // ops [previous op_end, op_end) of the sweep
    uint32_t cmask;       // index bits (inside the tile, outside every mixing mask of the run) that number the slot classes
};
// Waves that share a tile's rows, from the pair words per op of the sweep's largest tile: what bounds a sweep is the LDS traffic of that
// tile, and a row costs the same whether eight or sixty-four of its lanes hold a pair — as few waves as keep the rows of an op about
// full (measured at 24 qubits: 16 waves = rows 40 % full = no faster than a barrier per op).  0: no streams (small tiles: a round of
// the second sweep form costs no more than a row, and the streams take time to build).
inline int sec_stream_wave_count(double pairs_per_op, int max_waves) {
    if (pairs_per_op < 24.0) return 0;
    int nw = 1;
    while (nw < max_waves && pairs_per_op > 64.0 * nw) nw *= 2;
    return nw;
}
// The sweep's op list cut into runs whose mixing masks together leave log2(nw) + 1 (nw = 1: any number) of the tile's M index bits
// S untouched: those bits (at most class_bits of them, the highest) number slot classes that no op of the run leaves, so a wave
// that owns a class for the length of a run needs no barrier inside it.  false: an op mixes more bits than a run may.
inline bool sec_plan_wave_runs(const std::vector<SecBuildOp> &sops, uint64_t S, int M, int nw, int class_bits, std::vector<SecWaveRun> &runs) {
    runs.clear();
    const int wbits = __builtin_ctz((unsigned)nw);
    const int U = M - wbits - (nw > 1 ? 1 : 0);      // two slot classes per wave at least
    if (U < 2) return false;
    const int nops = (int)sops.size();
    for (int k = 0; k < nops;) {
        uint64_t u = 0;
        int l = k;
        while (l < nops && __builtin_popcountll(u | sops[l].x) <= U) u |= sops[l++].x;
        if (l == k) return false;
        const uint64_t cm = S & ~u;
        uint64_t cmask = 0;
        for (int b = 63, c = 0; b >= 0 && c < class_bits; --b)
            if ((cm >> b) & 1ull) {
                cmask |= 1ull << b;
                ++c;
            }
        runs.push_back({(int32_t)l, (uint32_t)cmask});
        k = l;
    }
    return true;
}

}  // namespace ovqe
