// CPU self-check of the regular-support tables (openvqe_amd/csrc/sv_regular_host.hpp), compiled with g++ under
// AddressSanitizer + UBSan by tests/test_sanitizer.py — the host-side planner of this round's kernels never ran under a
// sanitizer on the GPU box (no GPU sanitizers on the pool).
//
// For random programs with Z2 symmetries (rotation strings with even overlap with t random generators, ops as the sector
// path's SecBuildOp / SecPat lists: mixing mask, patterns, sign mask, flip), random tile bit sets and a random reference
// determinant, a sweep is REPLAYED from the tables exactly as k_sector_sweep_reg does it — swizzled tile of 2^m slots, group
// words, member offsets, selector variants, table entries, blocks of two ops — and compared, amplitude by amplitude, with
// the pair-by-pair definition of the same ops on the coset (what k_sec_pairs / k_sector_sweep implement with pair words).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "../../openvqe_amd/csrc/sv_regular_host.hpp"

using namespace ovqe;

static uint32_t par(uint64_t v) { return (uint32_t)(__builtin_popcountll(v) & 1); }
static uint64_t pdep(uint64_t v, uint64_t mask) {
    uint64_t r = 0;
    for (int k = 0; mask; mask &= mask - 1ull, ++k) r |= ((v >> k) & 1ull) << __builtin_ctzll(mask);
    return r;
}

struct Case {
    int n, t, M;
    std::vector<uint64_t> gens;   // full generators
    uint64_t hf;
    std::vector<SecBuildOp> ops;
    std::vector<SecPat> pats;
    std::vector<double> c, s;     // angle table
    uint64_t S;
};

static bool in_coset(const Case &C, uint64_t i) {
    for (uint64_t g : C.gens)
        if (par(i & g) != par(C.hf & g)) return false;
    return true;
}

// pair-by-pair definition (sec_rotate of sv_sector.hpp): u = first member, u' = c u + s v, v' = c v - s u, s = sign ? -sin : sin
static void reference_apply(const Case &C, const SecBuildOp &b, std::map<uint64_t, double> &psi) {
    std::map<uint64_t, double> out = psi;
    for (auto &kv : psi) {
        const uint64_t i = kv.first;
        for (int p = 0; p < b.npat; ++p) {
            const SecPat &pt = C.pats[(size_t)b.pat0 + p];
            if ((i & pt.pm) == pt.pv) {
                const uint64_t j = i ^ b.x;
                const uint32_t sign = par(i & b.zs) ^ (uint32_t)(b.flip & 1);
                const double cc = C.c[(size_t)b.tab0 + p], ss = sign ? -C.s[(size_t)b.tab0 + p] : C.s[(size_t)b.tab0 + p];
                const double u = psi.at(i), v = psi.at(j);
                out[i] = cc * u + ss * v;
                out[j] = cc * v - ss * u;
                break;
            }
            if ((i & pt.pm) == (pt.pv ^ (b.x & pt.pm))) break;   // second member: its partner does the work
        }
    }
    psi.swap(out);
}

// ---- per-wave streams of the irregular supports (sec_stream_wave_count, sec_plan_wave_runs): the runs partition the op list in order,
// are maximal, their class bits lie inside the tile and outside every mixing mask of the run — and, replayed on a random sparse tile,
// the classes are closed under the run's ops, so applying a run CLASS BY CLASS (what the waves do, each in op order on the slots of its
// classes, no barrier) gives the amplitudes of applying it op by op, bit for bit.
static int check_stream_plans(std::mt19937_64 &rng, int cases) {
    auto rnd = [&](uint64_t m) { return (uint64_t)(rng() % m); };
    int checked = 0;
    for (int cs = 0; cs < cases; ++cs) {
        const int n = 10 + (int)rnd(9), M = 6 + (int)rnd((uint64_t)std::min(n - 6, 11)) ;
        uint64_t S = 0;
        while (__builtin_popcountll(S) < M) S |= 1ull << rnd((uint64_t)n);
        std::vector<int> sbits;
        for (int b = 0; b < n; ++b)
            if ((S >> b) & 1ull) sbits.push_back(b);
        const int nops = 1 + (int)rnd(40);
        std::vector<SecBuildOp> ops((size_t)nops);
        uint64_t prev = 0;
        for (int o = 0; o < nops; ++o) {
            uint64_t x = 0;
            const int w = rnd(3) ? 4 : 2;
            if (prev && rnd(3)) {                       // neighbours in the list share most of their bits, as UCCSD's do
                x = prev;
                x &= ~(1ull << sbits[rnd((uint64_t)M)]);
            }
            while (__builtin_popcountll(x) < std::min(w, M)) x |= 1ull << sbits[rnd((uint64_t)M)];
            ops[(size_t)o] = SecBuildOp{x, 0, 0, 1, 0, o};
            prev = x;
        }
        const int nw = 1 << (int)rnd(5);
        std::vector<SecWaveRun> runs;
        const bool ok = sec_plan_wave_runs(ops, S, M, nw, 6, runs);
        const int wbits = __builtin_ctz((unsigned)nw), U = M - wbits - (nw > 1 ? 1 : 0);
        bool wide = U < 2;
        for (const SecBuildOp &b : ops) wide = wide || __builtin_popcountll(b.x) > U;
        if (!ok) {
            if (!wide) { printf("case %d: plan declined without a wide op\n", cs); return -1; }
            continue;
        }
        if (wide) { printf("case %d: plan accepted with an op wider than a run\n", cs); return -1; }
        int o0 = 0;
        for (size_t r = 0; r < runs.size(); ++r) {
            const int o1 = runs[r].op_end;
            if (o1 <= o0 || o1 > nops) { printf("case %d: run %zu does not advance\n", cs, r); return -1; }
            uint64_t u = 0;
            for (int o = o0; o < o1; ++o) u |= ops[(size_t)o].x;
            const uint64_t cm = runs[r].cmask;
            if ((cm & ~S) || (cm & u) || __builtin_popcountll(cm) > 6 || __builtin_popcountll(u) > U ||
                (nw > 1 && __builtin_popcountll(cm) < std::min(6, wbits + 1))) { printf("case %d: class mask of run %zu\n", cs, r); return -1; }
            if (o1 < nops && __builtin_popcountll(u | ops[(size_t)o1].x) <= U) { printf("case %d: run %zu is not maximal\n", cs, r); return -1; }
            o0 = o1;
        }
        if (o0 != nops) { printf("case %d: the runs do not cover the op list\n", cs); return -1; }
        // replay on a sparse tile: indices that agree outside S, a random half of them in the support
        const uint64_t outside = rnd(1ull << n) & ~S;
        std::map<uint64_t, double> amp_a, amp_b;
        for (uint64_t k = 0; k < (1ull << M); ++k)
            if (rnd(2)) {
                const uint64_t i = pdep(k, S) | outside;
                amp_a[i] = amp_b[i] = (double)rnd(1000) / 997.0 - 0.5;
            }
        auto rotate = [&](std::map<uint64_t, double> &amp, const SecBuildOp &b, uint64_t i, int o) {
            const uint64_t j = i ^ b.x;
            if (i > j) return;
            auto pj = amp.find(j);
            if (pj == amp.end()) return;
            const double c = std::cos(0.3 + 0.1 * o), sn = std::sin(0.3 + 0.1 * o);
            double &u = amp[i], &v = pj->second;
            const double nu = c * u + sn * v, nv = c * v - sn * u;
            u = nu;
            v = nv;
        };
        o0 = 0;
        for (size_t r = 0; r < runs.size(); ++r) {
            const int o1 = runs[r].op_end;
            const uint64_t cm = runs[r].cmask;
            for (int o = o0; o < o1; ++o)                      // (a) op by op
                for (auto &kv : amp_a) rotate(amp_a, ops[(size_t)o], kv.first, o);
            std::map<uint64_t, std::vector<uint64_t>> by_class;   // (b) class by class, each in op order
            for (auto &kv : amp_b) by_class[kv.first & cm].push_back(kv.first);
            for (auto &cl : by_class)
                for (int o = o0; o < o1; ++o)
                    for (uint64_t i : cl.second) {
                        if (amp_b.count(i ^ ops[(size_t)o].x) && (((i ^ ops[(size_t)o].x) & cm) != (i & cm))) { printf("case %d: a pair leaves its class\n", cs); return -1; }
                        rotate(amp_b, ops[(size_t)o], i, o);
                    }
            o0 = o1;
        }
        for (auto &kv : amp_a)
            if (kv.second != amp_b[kv.first]) { printf("case %d: class-by-class replay differs at %llx\n", cs, (unsigned long long)kv.first); return -1; }
        ++checked;
    }
    for (double p : {0.0, 23.9, 24.0, 64.0, 64.1, 128.0, 129.0, 400.0, 1e6}) {
        const int nw = sec_stream_wave_count(p, 16);
        if ((p < 24.0) != (nw == 0) || (nw && (nw & (nw - 1))) || nw > 16 || (nw > 1 && p <= 32.0 * nw) || (nw && nw < 16 && p > 64.0 * nw)) {
            printf("wave count %d for %.1f pairs per op\n", nw, p);
            return -1;
        }
    }
    return checked;
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? std::atoll(argv[2]) : 12345);
    auto rnd = [&](uint64_t m) { return (uint64_t)(rng() % m); };
    int checked = 0, blocks_seen = 0, selectors_seen = 0, declined = 0, runs_joined = 0;
    double worst = 0.0;
    for (int cs = 0; cs < cases; ++cs) {
        Case C;
        C.n = 8 + (int)rnd(5);
        C.t = 1 + (int)rnd(3);
        const uint64_t all = (1ull << C.n) - 1ull;
        // generators: parities over random disjoint classes of the bits (like the spin parities), sometimes recombined
        std::vector<uint64_t> cls((size_t)C.t, 0);
        for (int b = 0; b < C.n; ++b) cls[rnd((uint64_t)C.t)] |= 1ull << b;
        bool ok = true;
        for (uint64_t g : cls) ok = ok && __builtin_popcountll(g) >= 2;
        if (!ok) { --cs; continue; }
        C.gens = cls;
        if (C.t >= 2 && rnd(2)) C.gens[1] ^= C.gens[0];
        C.hf = rnd(1ull << C.n);
        // ops: x with even overlap with every class, weight 2 or 4; runs that share three of four bits (blocks) on purpose
        const int nops = 4 + (int)rnd(10);
        uint64_t prev_x = 0;
        for (int o = 0; o < nops; ++o) {
            uint64_t x = 0;
            if (prev_x && __builtin_popcountll(prev_x) == 4 && rnd(2)) {   // move one bit inside its class
                uint64_t bits[4];
                int k = 0;
                for (uint64_t m = prev_x; m; m &= m - 1ull) bits[k++] = m & (0ull - m);
                const uint64_t out = bits[rnd(4)];
                uint64_t klass = 0;
                for (uint64_t g : cls)
                    if (g & out) klass = g;
                const uint64_t cand = klass & ~prev_x;
                if (cand) {
                    uint64_t pick = cand;
                    for (uint64_t r = rnd((uint64_t)__builtin_popcountll(cand)); r; --r) pick &= pick - 1ull;
                    x = (prev_x ^ out) | (pick & (0ull - pick));
                }
            }
            while (!x) {
                const int pairs = 1 + (int)rnd(2);
                uint64_t cand = 0;
                for (int p = 0; p < pairs; ++p) {
                    const uint64_t g = cls[rnd((uint64_t)C.t)] & ~cand;
                    if (__builtin_popcountll(g) < 2) continue;
                    uint64_t a = g, b2;
                    for (uint64_t r = rnd((uint64_t)__builtin_popcountll(g)); r; --r) a &= a - 1ull;
                    a &= 0ull - a;
                    b2 = g & ~a;
                    for (uint64_t r = rnd((uint64_t)__builtin_popcountll(b2)); r; --r) b2 &= b2 - 1ull;
                    b2 &= 0ull - b2;
                    cand |= a | b2;
                }
                x = cand;
            }
            prev_x = x;
            SecBuildOp b = {};
            b.x = x;
            b.zs = rnd(1ull << C.n) & (rnd(3) ? ~x : ~0ull);   // mostly outside x, sometimes overlapping it
            b.flip = (int32_t)rnd(2);
            b.pat0 = (int32_t)C.pats.size();
            b.tab0 = (int32_t)C.c.size();
            if (rnd(4) == 0) {   // an OP_PAIR rotation: first member = pivot bit clear
                const uint64_t pivot = 1ull << (63 - __builtin_clzll(x));
                C.pats.push_back(SecPat{pivot, 0ull});
                b.npat = 1;
            } else {             // an OP_TAB run: a random set of active patterns over x (a pattern and its complement never both)
                std::vector<uint64_t> pv;
                const int w = __builtin_popcountll(x);
                for (uint64_t e = 0; e < (1ull << w); ++e) {
                    const uint64_t v = pdep(e, x);
                    bool dup = false;
                    for (uint64_t q : pv) dup = dup || q == v || q == (v ^ x);
                    if (!dup && rnd(3)) pv.push_back(v);
                }
                if (pv.empty()) pv.push_back(0);
                for (uint64_t v : pv) C.pats.push_back(SecPat{x, v});
                b.npat = (int32_t)pv.size();
            }
            for (int p = 0; p < b.npat; ++p) {
                const double phi = 0.1 + 1e-3 * (double)rnd(2000);
                C.c.push_back(std::cos(phi));
                C.s.push_back(std::sin(phi));
            }
            C.ops.push_back(b);
        }
        // the analysis must find the symmetries of the x masks (at least the t we built in)
        std::vector<uint64_t> xs;
        for (const SecBuildOp &b : C.ops) xs.push_back(b.x);
        std::vector<int> freebits;
        std::vector<uint64_t> G;
        z2_symmetries(xs, C.n, freebits, G);
        if ((int)freebits.size() < C.t) { std::printf("case %d: %zu symmetries found, %d built in\n", cs, freebits.size(), C.t); return 1; }
        for (size_t k = 0; k < freebits.size(); ++k)
            for (uint64_t x : xs)
                if (par(x & (G[k] | (1ull << freebits[k])))) { std::printf("case %d: generator does not commute\n", cs); return 1; }
        // the coset of ALL found symmetries through hf
        C.gens.clear();
        for (size_t k = 0; k < freebits.size(); ++k) C.gens.push_back(G[k] | (1ull << freebits[k]));
        const int t = (int)freebits.size();
        // one sweep over the ops that fit a random tile bit set
        uint64_t S = 0;
        size_t take = 0;
        const int Mmax = std::min(C.n - 1, t + 2 + (int)rnd(6));
        while (take < C.ops.size() && __builtin_popcountll(S | C.ops[take].x) <= Mmax) S |= C.ops[take++].x;
        if (!take) { ++declined; continue; }
        while (__builtin_popcountll(S) < Mmax) S |= 1ull << rnd((uint64_t)C.n);
        const int M = __builtin_popcountll(S);
        std::vector<int> dep;
        std::vector<uint64_t> depG;
        if (M <= t || !sweep_symmetries(freebits, G, S, C.n, dep, depG)) { ++declined; continue; }
        std::vector<SecBuildOp> sops(C.ops.begin(), C.ops.begin() + (long)take);
        std::vector<SecRegOp> rops;
        std::vector<uint32_t> emap, gwords;
        const bool pairs = rnd(4) != 0;
        // runs without barriers: "waves" of 2^lane_bits groups, 2^wave_bits of them (the product: 6 and 2; small here so that the tiles of
        // these registers have several waves); 0 wave bits = a barrier after every unit
        const int lane_bits = 1 + (int)rnd(2), wave_bits = (int)rnd(3), run_cap = 1 + (int)rnd(6);
        if (!build_reg_ops(sops, C.pats, S, all, C.hf, dep, depG, M - t, pairs, rops, emap, gwords, lane_bits, wave_bits, run_cap)) { ++declined; continue; }
        // ---- the coset, a random real state on it
        std::map<uint64_t, double> psi;
        for (uint64_t i = 0; i <= all; ++i)
            if (in_coset(C, i)) psi[i] = (double)rnd(2001) / 1000.0 - 1.0;
        if (psi.size() != (size_t)(1ull << (C.n - t))) { std::printf("case %d: coset size\n", cs); return 1; }
        std::map<uint64_t, double> ref = psi;
        for (const SecBuildOp &b : sops) reference_apply(C, b, ref);
        // ---- replay from the tables
        uint64_t depmask = 0;
        for (int f : dep) depmask |= 1ull << f;
        const uint64_t kept_in = S & ~depmask, outside = all & ~S;
        const int m = M - t;
        const uint32_t nslots = 1u << m, ntiles = 1u << (C.n - M);
        auto member = [&](uint32_t tile, uint32_t slot) {
            uint64_t i = pdep(tile, outside) | pdep(slot, kept_in);
            for (size_t k = 0; k < dep.size(); ++k) {
                const uint32_t s_f = (uint32_t)((C.hf >> dep[k]) & 1ull) ^ par(C.hf & depG[k]);
                if (s_f ^ par(i & depG[k])) i |= 1ull << dep[k];
            }
            return i;
        };
        std::vector<double> T(emap.size() * 2);
        for (size_t e = 0; e < emap.size(); ++e) {
            const uint32_t mm = emap[e];
            T[2 * e] = mm == 0xffffffffu ? 1.0 : C.c[mm & 0x7fffffffu];
            T[2 * e + 1] = mm == 0xffffffffu ? 0.0 : ((mm >> 31) ? -C.s[mm & 0x7fffffffu] : C.s[mm & 0x7fffffffu]);
        }
        std::map<uint64_t, double> got;
        for (uint32_t tile = 0; tile < ntiles; ++tile) {
            std::vector<double> lds(nslots + 64, 0.0);
            for (uint32_t k = 0; k < nslots; ++k) {
                const uint64_t i = member(tile, k);
                if (!psi.count(i)) { std::printf("case %d: slot (%u, %u) is not a member\n", cs, tile, k); return 1; }
                lds[sec_reg_swz(k)] = psi.at(i);
            }
            // who wrote a slot last since the last barrier (-1: nobody): inside a run a wave may only touch slots that no OTHER wave wrote
            std::vector<int> writer(nslots + 64, -1);
            int run_len = 1;
            for (size_t o = 0; o < rops.size();) {
                const SecRegOp &r = rops[o];
                const uint32_t *gw = gwords.data() + o * SEC_REG_GSTRIDE;
                const int w = (int)(r.w_nsel & 0xffffu), nsel = (int)((r.w_nsel >> 16) & 0xffu);
                const bool block = (r.w_nsel >> 24) & 1u;
                const int nbits = block ? 4 : w, na = 1 << nbits;
                selectors_seen += nsel > 0;
                uint32_t depo[16];
                for (int e = 0; e < na; ++e) depo[e] = (r.dep[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                const uint32_t tz = par(tile & r.zt);
                for (uint32_t g = 0; g < (nslots >> nbits); ++g) {
                    const uint32_t wd = gw[g], sb = wd & 0xffffu;
                    double a[16];
                    const int wave_of_g = (int)((g >> lane_bits) & ((1u << wave_bits) - 1u));
                    for (int e = 0; e < na; ++e) {
                        const size_t at = (sb ^ depo[e]) >> 3;
                        if (writer.at(at) >= 0 && writer[at] != wave_of_g) {
                            std::printf("case %d: op %zu, group %u (wave %d) touches a slot wave %d wrote since the last barrier\n", cs, o, g, wave_of_g, writer[at]);
                            return 1;
                        }
                        writer[at] = wave_of_g;
                        a[e] = lds.at(at);
                    }
                    if (!block) {
                        const uint32_t neg = ((wd >> 16) & 1u) ^ tz;
                        uint32_t sel = 0;
                        if (nsel > 0) sel = ((wd >> 17) & 1u) ^ par(tile & r.sel_t[0]);
                        if (nsel > 1) sel |= (((wd >> 18) & 1u) ^ par(tile & r.sel_t[1])) << 1;
                        const int np = na / 2;
                        for (int q = 0; q < np; ++q) {
                            const size_t e = (size_t)o * SEC_REG_TSTRIDE + sel * (uint32_t)np + (uint32_t)q;
                            const double cc = T[2 * e], ss = T[2 * e + 1], sg = neg ? -1.0 : 1.0;
                            const double u = a[q], v = sg * a[na - 1 - q];
                            a[q] = cc * u + ss * v;
                            a[na - 1 - q] = sg * (cc * v - ss * u);
                        }
                    } else {
                        ++blocks_seen;
                        const uint32_t fl = r.pad[0];
                        const uint32_t tzB = par(tile & r.pad[1]), tsA = nsel ? par(tile & r.sel_t[0]) : 0u, tsB = nsel ? par(tile & r.pad[2]) : 0u;
                        for (uint32_t v = 0; v < 2; ++v) {
                            const uint32_t neg = ((wd >> 16) & 1u) ^ tz ^ (v & fl);
                            const uint32_t sel = nsel ? (((wd >> 17) & 1u) ^ tsA ^ (v & (fl >> 1))) & 1u : 0u;
                            for (int q = 0; q < 4; ++q) {
                                const size_t e = (size_t)o * SEC_REG_TSTRIDE + sel * 4u + (uint32_t)q;
                                const double cc = T[2 * e], ss = T[2 * e + 1], sg = neg ? -1.0 : 1.0;
                                const double u = a[8 * v + q], x2 = sg * a[8 * v + 7 - q];
                                a[8 * v + q] = cc * u + ss * x2;
                                a[8 * v + 7 - q] = sg * (cc * x2 - ss * u);
                            }
                        }
                        for (uint32_t v = 0; v < 2; ++v) {
                            const uint32_t neg = ((wd >> 18) & 1u) ^ tzB ^ (v & (fl >> 2));
                            const uint32_t sel = nsel ? (((wd >> 19) & 1u) ^ tsB ^ (v & (fl >> 3))) & 1u : 0u;
                            for (int q = 0; q < 4; ++q) {
                                const size_t e = (size_t)(o + 1) * SEC_REG_TSTRIDE + sel * 4u + (uint32_t)q;
                                const double cc = T[2 * e], ss = T[2 * e + 1], sg = neg ? -1.0 : 1.0;
                                const double u = a[4 * v + q], x2 = sg * a[4 * v + 11 - q];
                                a[4 * v + q] = cc * u + ss * x2;
                                a[4 * v + 11 - q] = sg * (cc * x2 - ss * u);
                            }
                        }
                    }
                    for (int e = 0; e < na; ++e) lds.at((sb ^ depo[e]) >> 3) = a[e];
                }
                if ((r.w_nsel >> 27) & 1u) {   // the next unit follows without a barrier
                    ++runs_joined;
                    if (++run_len > run_cap) { std::printf("case %d: a run longer than run_cap\n", cs); return 1; }
                    if (wave_bits == 0 || o + (block ? 2 : 1) >= rops.size()) { std::printf("case %d: run flag where no run can be\n", cs); return 1; }
                } else {
                    std::fill(writer.begin(), writer.end(), -1);   // barrier
                    run_len = 1;
                }
                o += block ? 2 : 1;
            }
            for (uint32_t k = 0; k < nslots; ++k) got[member(tile, k)] = lds[sec_reg_swz(k)];
        }
        for (auto &kv : ref) {
            const double d = std::fabs(kv.second - got.at(kv.first));
            worst = std::max(worst, d);
            if (d > 1e-12) {
                std::printf("case %d: n=%d t=%d M=%d ops=%zu pairs=%d: amplitude %llu differs by %.3e\n", cs, C.n, t, M, take, (int)pairs,
                            (unsigned long long)kv.first, d);
                return 1;
            }
        }
        ++checked;
    }
    std::printf("regular tables ok: %d sweeps replayed (%d blocks of two ops, %d ops with selectors, %d barriers left out), %d declined, worst |delta| %.1e\n", checked,
                blocks_seen, selectors_seen, runs_joined, declined, worst);
    const int plans = check_stream_plans(rng, cases);
    if (plans < 0) return 3;
    std::printf("stream plans ok: %d plans replayed class by class\n", plans);
    return checked >= cases / 5 && plans >= cases / 5 ? 0 : 2;
}
