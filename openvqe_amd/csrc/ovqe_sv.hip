// ovqe_sv.hip — C ABI (include/ovqe_sv.h) + host-side engine of the MI355X statevector backend.
// gfx950 only; no CPU fallback: every entry point needs a live device.
//
// One translation unit, split along its sections (each .inc is included exactly once, in this order):
//   this file        handle, options, device buffers, streaming-path launches, sector_prepare / program bookkeeping, handle creation,
//                    the first C-ABI entry points (version, errors, options, create / destroy, adopt)
//   tile_host.inc    planners + launches of the LDS-tiled <H>, compact cover, tile sweeps            (kernels: sv_tile.hpp)
//   sector_host.inc  symmetry-sector tables, sweeps, <H>, adjoint, Lanczos inside the support        (kernels: sv_sector.hpp)
//   gates_host.inc   literal gate programs -> small ops / Clifford-frame rotations                   (sv_small.hpp, sv_frame_host.hpp)
//   sparse_host.inc  support-compacted path                                                          (kernels: sv_sparse.hpp)
//   abi_unit.inc / abi_eval.inc / abi_adapt.inc / abi_solvers.inc   extern "C" entry points by family (include/ovqe_sv.h)
//   cross_host.inc   planned Pauli sums on a shard of the partitioned register, ovqe_xsum_*          (kernels: sv_cross.hpp)
#include "../../include/ovqe_sv.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <complex>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "sv_kernels.hpp"
#include "sv_small.hpp"
#include "sv_sparse.hpp"
#include "sv_tile.hpp"
#include "sv_sector.hpp"
#include "sv_cross.hpp"
#include "sv_frame_host.hpp"
#include <hipcub/hipcub.hpp>
#include <unordered_map>
#include <unordered_set>

using namespace ovqe;

namespace {

using ovqe_frame::PauliRaw;
using ovqe_frame::pauli_mul;
using ovqe_frame::FrameEmit;
using ovqe_frame::FrameTrack;
using ovqe_frame::track_clifford_frame;
using ovqe_frame::clifford_amplitude_on_host;

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Device blocks released while a table build is running are kept for the next allocations instead of going back to the driver:
// hipFree synchronises the device and unmaps (about 0.1 ms a call), and the per-sweep temporaries of a build repeat their sizes sweep
// after sweep (N2 UCCSD: table build 123 -> 109 ms; an ADAPT run rebuilds its tables every macro-iteration).  The kept blocks belong to
// the handle (ovqe_destroy frees them); a DevBlockScope makes them the target of free / the source of allocations for the duration
// of one build and trims them to 256 MB / 512 blocks on the way out.
struct DevBlockCache {
    std::vector<DevBuf> blocks;
    static thread_local DevBlockCache *current;
    void flush() {
        for (DevBuf &b : blocks)
            if (b.p) (void)hipFree(b.p);
        blocks.clear();
    }
    void trim(size_t max_bytes, size_t max_blocks) {
        std::sort(blocks.begin(), blocks.end(), [](const DevBuf &a, const DevBuf &b) { return a.cap < b.cap; });
        size_t total = 0;
        for (const DevBuf &b : blocks) total += b.cap;
        while (!blocks.empty() && (total > max_bytes || blocks.size() > max_blocks)) {   // largest first
            total -= blocks.back().cap;
            (void)hipFree(blocks.back().p);
            blocks.pop_back();
        }
    }
    void *take(size_t want, size_t *cap) {   // smallest kept block that holds `want` without wasting more than an eighth of it
        int best = -1;
        for (int i = 0; i < (int)blocks.size(); ++i)
            if (blocks[i].cap >= want && blocks[i].cap <= want + want / 8 + 4096 && (best < 0 || blocks[i].cap < blocks[best].cap)) best = i;
        if (best < 0) return nullptr;
        void *p = blocks[best].p;
        *cap = blocks[best].cap;
        blocks[best] = blocks.back();
        blocks.pop_back();
        return p;
    }
};
thread_local DevBlockCache *DevBlockCache::current = nullptr;
struct DevBlockScope {
    DevBlockCache *outer;
    explicit DevBlockScope(DevBlockCache &c) : outer(DevBlockCache::current) { DevBlockCache::current = &c; }
    ~DevBlockScope() {
        DevBlockCache::current->trim((size_t)256 << 20, 512);
        DevBlockCache::current = outer;
    }
};


struct TilePlan {  // segmentation of an op list into LDS-tiled sweeps (sv_tile.hpp)
    std::vector<TileSeg> tsegs;
    std::vector<TileOp> tops;
    std::vector<TileRot> trots;
    std::vector<int32_t> plan;  // >= 0: tile segment; < 0: op (-1 - index) launched as its own sweep
    DevBuf d_tops, d_trots;
};

struct HamDev {  // grouped Pauli sum resident on the device
    std::vector<HGroup> groups;
    std::vector<HTerm> terms;
    DevBuf d_groups, d_terms;
    double constant = 0.0;
    bool set = false;
    // tile cover of the x-groups (sv_tile.hpp k_tile_expect), built lazily for (tile_bits, tile_low)
    int tile_bits = -1, tile_low = -1;
    bool tile_real = false;       // cover built for a real-amplitude state (masks in pair-index space, M + 1 bits)
    int version = 0;              // ham_real: the version of the stored Hamiltonian it was copied from
    std::vector<ExSweep> tsweeps;
    std::vector<int> tsweep_terms;  // apply-form terms per sweep: the compute weight of a sweep on sparse tiles
    int n_rest = 0;  // groups that fit no tile keep their own sweep (k_expect_pairs)
    int64_t tile_work = 0, tile_entries = 0, tile_terms = 0;  // pair x term evaluations per tile over all sweeps
    DevBuf d_tchunks, d_tgroups, d_tterms, d_tflats, d_titems, d_rest;
    DevBuf d_achunks, d_agroups, d_aterms;  // operator-application form of the cover (k_tile_apply, sparse tiles)
    std::vector<ExChunkT> h_achunks;        // host copy (launch geometry of the compact cover)
    int cover_id = 0;                       // bumped whenever the cover is rebuilt
    // the diagonal group for k_tile_diag (Walsh-Hadamard form, dense registers): unique tile-local z masks, CSR of their terms
    DevBuf d_dzin, d_doff, d_dterms;
    int diag_nu = 0, diag_bits = 0, diag_sweep = -1;   // unique masks, tile bits of the form, the cover sweep that holds the group
};

// Pauli sum planned ONCE for a shard of the partitioned register (sv_cross.hpp, cross_host.inc): the terms without an x part on the
// rank bits as a HamDev of their own (tile cover, pair trick), the others grouped by partner shard with a pass list each
struct CrossCover {          // the groups of one partner (rank difference d)
    uint64_t d = 0;
    int ngroups = 0, nterms = 0;
    bool small = false;      // chunks below the tile sizes: k_cross_small, one launch per class of high x bits
    int M = 0;               // tile bits of the passes
    std::vector<CrossPass> passes;
    DevBuf d_achunks, d_agroups, d_aterms;          // tile form
    std::vector<uint64_t> class_h;                   // small form: x bits above the chunk per class ...
    std::vector<std::pair<int, int>> class_groups;   // ... and its group range
    DevBuf d_groups, d_terms;
};
struct CrossRawGroup {
    uint64_t x;                 // local x mask
    std::vector<HTerm> terms;   // full z masks, i^ny folded
};
struct CrossSum {
    int chunk_bits = 0;
    bool hermitian = false;  // every coefficient real: expectation values allowed
    HamDev local;            // d = 0
    bool has_local = false;
    std::vector<std::pair<uint64_t, std::vector<CrossRawGroup>>> raw;   // the terms of every rank difference d != 0, ascending
    std::vector<CrossCover> partners[2];   // their pass lists: [0] complex amplitudes, [1] real amplitudes (option "real_state"); built at first use
    bool built[2] = {false, false};
    DevBuf d_part;           // per-workgroup partial sums of the remote contractions of one expectation value
    size_t part_slots = 0;
};

// compact cover (sv_tile.hpp k_tile_expect_compact): the support of the program's states, sorted by tile for every sweep
struct CompactCover {
    bool valid = false, disabled = false;
    int prog_version = -1, ham_version = -1, cover_id = -1;
    int seen = 0;                 // evaluations of this (program, Hamiltonian) pair before the cover is built
    uint32_t K = 0, max_nnz = 0;
    uint64_t ntiles = 0;
    DevBuf d_sup, d_psic, d_loc, d_cid, d_off, d_sweeps;
};

// sector path (sv_sector.hpp): the program and the Hamiltonian restated on the support of the program's states
struct SectorLayout {   // the support sorted for one tile bit set
    uint32_t smask = 0;
    int M = 0;
    uint32_t ntiles = 0, max_tile = 0;
    DevBuf d_cid, d_off, d_src;
};
struct SectorSeg {      // one sweep of the circuit
    SectorLayout L;
    int nops = 0;
    int rot0 = 0, nrot = 0;       // the sweep's range of the angle table
    uint32_t hf_pos = 0;
    uint64_t npairs = 0;
    DevBuf d_tab0, d_poff, d_pairs;
    uint32_t max_op_pairs = 0;    // most pair words of one op in one tile (k_sector_sweep<NT, true> needs them to fit a staging buffer)
    DevBuf d_srcpad;              // gather indices of the sweep, tile-padded (k_sec_pad_src): first form of the sweep kernel
    DevBuf d_dstpad;              // scatter indices into the next sweep's tile-padded order (k_sector_sweep2)
    DevBuf d_wide, d_rounds;      // 64-bit pair words, rounds per (tile, chunk) (k_sec_widen)
    DevBuf d_stream, d_rowhdr, d_rowinfo;   // per-wave streams of the third sweep form (k_sector_sweep3): rows of 64 pair words, cos/sin base per row, run boundaries per (tile, wave)
    int nruns = 0;                // runs of the sweep's op list (0: no streams; the second form serves the sweep)
    int stream_waves = 0;         // waves per workgroup that share the rows
    uint64_t stream_rows = 0;
    DevBuf d_bdst;                // scatter indices into the PREVIOUS sweep's tile-padded order (k_sector_adjoint2)
    DevBuf d_torder;              // tiles by population, largest first (sweeps with many tiles per CU)
    uint32_t maxchunks = 0;
    DevBuf d_regops, d_reggw;     // regular supports (k_sector_sweep_reg): the sweep's SecRegOp list, its group words
    uint32_t reg_kept = 0;        // its kept inside bits (index space; slot = pext(index, reg_kept))
    DevBuf d_regsrc, d_reggslot, d_regoslot;   // gather in RUNS: source position / slot of gather step j; slot stored at position j (see k_sector_sweep_reg)
    int nregtab = 0;              // entries of its (c, s) table ...
    uint32_t regtab0 = 0;         // ... from this entry of the engine's table on
};
struct SectorHSweep {   // one sweep of the materialised <H>
    SectorLayout L;
    uint64_t nnz = 0;
    int ndict = 0;          // magnitudes in the sweep's dictionary (0: the coded stream keeps explicit values)
    int packed = 0;         // 1: the coded words are stored as 24-bit elements
    DevBuf d_cbase, d_clen, d_cwords, d_cvals, d_dict, d_xbase, d_xlen, d_xwords, d_xvals, d_order;   // row format (sv_sector.hpp)
    DevBuf d_torder;              // tiles by population, largest first
};
struct SectorEngine {
    bool valid = false, disabled = false;
    bool h_tables = false;        // the materialised <H> is part of the engine (else: circuit only, <H> by the compact cover)
    int prog_version = -1, ham_version = -1;
    int seen = 0;                 // evaluations of this (program, Hamiltonian) pair before the tables are built
    int probe_mode = 0;           // 0: support probed with one angle per PARAMETER; 1: one per rotation (sector_orphaned)
    double last_eval_us = 0.0;    // host time from the <H> launch to the result of the last lone evaluation (whether the next one polls)
    uint32_t K = 0, max_tile = 0, h_max_tile = 0;
    uint32_t hf_final = 0;        // position of |hf> in the final circuit order (= the order of the <H> tables' vectors)
    uint32_t last_fci_block = 0;  // determinants of the block ovqe_sector_ground_state diagonalised last
    int last_fci_rounds = 0;      // matvec rounds its reachability search took to saturate
    int M = 0, Mh = 0;            // index bits per tile: circuit sweeps, <H> sweeps
    int sb = 13;                  // slot bits of the pair words
    bool regular = false;         // the support is a full coset of the program's Z2 symmetries and every sweep has its SecRegOp list
    int reg_m = 0;                // slot bits of a tile then (tile bits minus the free bits)
    int reg_plan_threads = 0;     // workgroup size the barrier-free runs were planned for (0: none planned)
    bool coset_assumed = false;   // the support is the coset of the program's Z2 symmetries, taken without a probe (gate lists)
    bool coset_rejected = false;  // ... and the check of the first build found it mostly empty: this program is probed
    uint32_t freemask = 0;        // the free (dependent) index bits of the coset
    DevBuf d_regmap, d_regtab;    // table-entry map of all sweeps (angle-table entry | sign << 31, or none), the (c, s) table of the evaluation
    uint32_t nregtab = 0;
    uint64_t npairs = 0, nnz = 0;
    size_t bytes = 0;
    size_t pad_elems = 0;         // doubles of a state buffer in tile-padded form (largest sweep; 0: no dst tables)
    uint32_t chunk = 2048;        // pair words per chunk of the 64-bit tables (k_sec_widen)
    std::vector<SectorSeg> segs;
    std::vector<SectorHSweep> hs;
    DevBuf d_sup, d_buf[2], d_hdesc, d_flag;
    DevBuf d_lam[2], d_w, d_wpart;   // adjoint gradient: lambda (ping-pong), per-entry sums, per-tile partials
    DevBuf d_bbuf[2], d_brp, d_benergies;   // batched evaluations: state slices (ping-pong), angle tables, energies
    size_t budget = 0;
    int h_max_dict = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // option "sector_profile": start / after the sweeps / after <H>
    float last_circuit_ms = 0.f, last_expect_ms = 0.f;
    size_t h_stream_bytes = 0;    // bytes k_sector_expect reads per evaluation (elements + index arrays)
};

}  // namespace

struct ovqe_sv {
    int n_local = 0, n_global = 0, device = 0;
    uint64_t shard = 0, base = 0, namps = 0;
    hipStream_t stream = nullptr;
    amp_t *state = nullptr;
    bool own_state = true;
    amp_t *scratch[2] = {nullptr, nullptr};
    std::string err;

    // reduction workspace
    DevBuf d_partials, d_result;
    // rotation tables
    DevBuf d_rp;
    RotParam *h_rp = nullptr;  // pinned
    size_t h_rp_cap = 0;
    double2 *h_result = nullptr;  // pinned, small
    DevBlockCache kept_blocks;    // device blocks released by table builds, kept for the next build (DevBlockScope)
    double *h_fin = nullptr, *d_fin = nullptr;   // mapped: energy + flag of a sector evaluation, written by k_sector_finish
    bool fin_failed = false;
    int opt_poll_result = 1;   // lone evaluations: watch the mapped result slot instead of synchronising the stream (poll_mapped_slot)
    int opt_sector_fused_reduce = 1;
    int opt_sector_pairs_form = 2;   // pair-table builder: 2 = k_sec_pairs2 (ops staged in LDS, no barrier per op), 1 = first form
    // small batches (the one-evaluation-per-call loops of scipy's optimisers): parameters and energies travel through one
    // pinned, device-mapped buffer that the fused kernels read / write directly — launch + sync instead of two copies,
    // two event records and their completion round trips
    double *h_io = nullptr;
    double *d_io = nullptr;       // device alias of h_io
    static constexpr size_t IO_DOUBLES = 131072;

    HamDev ham;
    // Gate programs whose Clifford part does not close (compile_gate_program_frame): the compiled program is the Pauli-rotation
    // sequence alone, energies are evaluated with the stored Hamiltonian conjugated by the net Clifford operator
    // (<C phi|H|C phi> = <phi|C^+ H C|phi>: every term stays one Pauli string), ovqe_prepare_state applies the Clifford
    // gates literally behind the rotations.  ham_conj takes ham's place for the duration of an energy / gradient call.
    HamDev ham_conj;
    bool frame_open = false;
    std::vector<uint64_t> frame_img;          // images of X_q, Z_q under the net Clifford: (x, z, k) triples, 2 n of them
    std::vector<int32_t> tail_gates;          // the Clifford gates in order: (opcode, b0, b1, quarter-turn sign) quadruples
    std::vector<uint64_t> user_x, user_z;     // the stored Hamiltonian as the caller gave it
    std::vector<double> user_c;
    double user_const = 0.0;
    int ham_versions = 0;                     // version numbers are unique over ham and ham_conj
    HamDev ham_adhoc;             // last Hermitian sum evaluated by ovqe_expectation / ovqe_bilinear on the own state
    std::vector<uint64_t> adhoc_x, adhoc_z;
    std::vector<double> adhoc_c;
    // compiled program
    bool prog_set = false;
    int32_t K = 0;
    uint64_t hf = 0;
    std::vector<SmallOp> ops;    // sequential program (streaming path)
    std::vector<SmallRot> rots;
    std::vector<SmallOp> sops;   // fused-kernel program: ops with commuting runs turned into OP_TAB
    std::vector<SmallRot> srots; // its table entries (sequential rotations and OP_TAB patterns)
    std::vector<int32_t> sop_src;  // source op (index into ops) of every fused-program op
    std::vector<uint64_t> sop_zc;  // OP_TAB: the run's common z mask outside x, all 64 bits
    std::vector<SmallSeg> segs;
    DevBuf d_ops, d_rots, d_segs, d_stream;
    DevBuf d_rots_seq;            // the sequential program's rotations (device-side angle resolution)
    std::vector<uint16_t> idx_stream;  // precomputed (sign<<15 | index) streams of the OP_TAB ops
    int cs_capacity = 512;
    // batched evaluation workspace
    DevBuf d_theta, d_energies, d_workspace;
    // support-compacted program (sv_sparse.hpp): built lazily for the current (program, Hamiltonian)
    bool sp_tried = false, sp_valid = false;
    int sp_m = 0, sp_nops = 0, sp_nent = 0;
    int sp_nrows4 = 0;            // rows of the throughput kernel (multiple of four; 0: not built)
    int sp_nprim = 0;             // distinct angles of the program = entries of that kernel's cos/sin table
    DevBuf d_sp_rows, d_sp_rows64, d_sp_prim;
    int sp_nrows8 = 0;            // rows of 64 of the latency kernel (multiple of eight; 0: not built)
    int sp_mp = 0, sp_hf = 0;     // slots of the compact state (support padded to a multiple of 32 when renumbered), slot of |hf>
    int64_t sp_conflicts_before = 0, sp_conflicts_after = 0;   // colliding lane pairs per evaluation, discovery order / renumbered
    int64_t sp_npairs = 0;
    DevBuf d_sp_ops, d_sp_pairs, d_sp_entries;
    // device copy of the ADAPT pool of the last ovqe_pool_gradients call (+ its host image for the change test)
    DevBuf d_pg_off, d_pg_xs, d_pg_terms, d_pg_out, d_pg_part;
    DevBuf d_pg_runs, d_pg_tabs;   // pattern tables of the pool's same-x runs (PoolRun, k_pool_grad_nz)
    bool pg_tables = false;
    DevBuf d_nz_cnt, d_nz_start, d_nz_idx, d_nz_val, d_nz_bitmap;
    DevBuf d_exp_groups, d_exp_terms;   // x-groups / terms of the operator of ovqe_apply_exp_pauli_sum
    // d_nz_idx[0, nz_super_count) is known to CONTAIN the support of the state: set by ovqe_init_basis / ovqe_apply_exp_pauli_sum when they
    // leave, cleared by the next entry point whatever it is (OVQE_ENTER; nz_super_prev = what that entry found) — the chain of exact
    // exponentials behind an ADAPT screen state then lists the support once instead of scanning the register per operator.  Never on
    // a state the caller can write behind the library's back (adopted buffers, ovqe_state_ptr taken).
    // the rotation list of the previous ovqe_set_program call, as given: a program that EXTENDS it (an ADAPT ansatz one macro-iteration
    // later: same rotations, new ones behind them) starts from what was learnt about its predecessor (sector_prepare: probe mode)
    std::vector<uint64_t> prev_x, prev_z;
    std::vector<double> prev_coeff, prev_phi0;
    std::vector<int32_t> prev_pidx;
    uint64_t prev_hf = 0;
    bool prog_extends_prev = false;
    // which kernel forms of the sector path served this handle since the last ovqe_set_program / ovqe_set_gate_program (ovqe_last_support
    // which = 6; the tests name the geometry that selects each form): bit 0 first sweep form (pair words), 1 second (64-bit words + rounds),
    // 2 third (per-wave streams), 3 regular supports (bit arithmetic); 4..7 the backward sweeps of ovqe_energy_gradient in the same order;
    // 8 / 9 the first / second form of the pair-table builder
    uint32_t forms_used = 0;
    bool nz_super = false, nz_super_prev = false, state_exposed = false;
    uint64_t nz_super_count = 0;
    DevBuf d_tile_smasks, d_tile_lists, d_tile_counts;   // non-empty tiles per sweep of H psi on a listed state (k_tile_lists)  // support list of the screened state (k_pool_grad_nz)
    int opt_screen_sparse = 16;   // the ADAPT screen walks the support of psi when it is at most 1/this of the register (0 = never)
    int64_t last_exp_support = -1;     // amplitudes the last ovqe_apply_exp_pauli_sum call's Taylor steps ran over (-1: the register)
    int64_t last_screen_sector = 0;    // determinants of the symmetry sector whose materialised Hamiltonian gave the last screen's sigma (0: register / tile cover)
    int64_t last_screen_support = -1;  // support size seen by the last ovqe_pool_gradients call (-1: register walked)
    std::vector<int64_t> pg_off;
    std::vector<uint64_t> pg_xs;
    std::vector<HTerm> pg_terms;
    bool pg_valid = false;
    int opt_sparse = 1;           // allow the support-compacted path
    int opt_sparse_spw = 0;       // evaluations per wave (0 = automatic)
    int opt_sparse_dbg = 0;       // measurement: k_sparse_vqe_rows without one of its phases (SparseArgs::dbg)
    int opt_clifford_phase_host = 1;   // global phase of a closed Clifford frame from a sparse host simulation (0: the gates run on the device)
    int opt_sparse_dealias = 1;   // arrange the restricted-Hamiltonian entries against LDS bank conflicts
    int opt_sparse_renumber = 1;  // number the compact support against LDS bank conflicts of the circuit's pairs
    int opt_sparse_rows = 1;      // support-compacted evaluation, large batches: flat rows of padded 64-bit pair words (k_sparse_vqe_rows)
    int opt_sparse_wg = 1;        // small batches (<= 1024): one evaluation per 1024-thread workgroup (k_sparse_vqe_wg)
    int opt_sparse_grad = 1;      // ovqe_energy_gradient on the compact support in one launch (n <= 16)
    // pair-index-space expectation tables of the fused kernel, built per (thread bits, real mode)
    DevBuf d_egroups, d_eterms, d_echunks, d_eflat;
    int exp_lbits = -1, exp_real = -1, exp_ngroups = 0, exp_nchunks = 0, exp_nflat = 0;
    int exp_ham_version = -1;     // version of the Hamiltonian the fused kernel's expectation tables were built from
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream of the tiled <H>: the compute-heavy sweeps (many x-groups) and the bandwidth-bound ones (few groups,
    // one read of the state each) run side by side
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int opt_expect_streams = 2;
    int opt_expect_diag_wht = 1;  // dense registers of 25+ qubits: the diagonal group of a tiled <H> by a Walsh-Hadamard transform per tile (k_tile_diag)
    int opt_real_state = 0;       // option "real_state": the state buffer holds 2^n_local DOUBLES (a shard of the partitioned register while
                                  // every applied rotation has an odd number of Y): ovqe_apply_pauli_rotations, ovqe_init_basis, ovqe_norm2
                                  // and the ovqe_xsum_expect_* calls work on 8-byte amplitudes
    CompactCover cc;              // of (current program, ham_real)
    int opt_compact = 1;          // allow the compact cover (real-amplitude streaming energies, 18..28 qubits)
    int opt_compact_cpp = 1;      // host chunks (512 terms each) staged in LDS per pass of the compact-cover kernel
    int prog_version = 0;
    SectorEngine sec;             // of (current program, stored Hamiltonian)
    SectorEngine scr;             // ADAPT screens: <H> tables on a symmetry sector, no circuit (build_screen_sector)
    int scr_failed_version = -1;  // Hamiltonian version for which the screen engine was declined
    bool probe_independent = false;  // resolve_angles: one quasi-random angle per ROTATION (support probe, second attempt)
    int opt_sector = 1;           // allow the sector path (real-amplitude streaming energies on a sparse support)
    int opt_sector_bits = 0;      // index bits per tile (0 = automatic: n - 8, at most 16)
    int opt_sector_max_gb = 128;  // table budget (also capped at 60 % of the free device memory)
    int opt_sector_threads = 0;   // workgroup size of the circuit sweeps (0 = automatic; 64: one wave per tile, no barriers)
    int opt_sector_min_qubits = 18;
    int opt_sector_h = 1;         // materialise <H> on the support when it fits the budget
    int opt_sector_h_bits = 0;    // index bits per <H> tile (0 = automatic: 300 .. 600 amplitudes per tile)
    int opt_sector_dict = 1;      // dictionary coding of the double-excitation-like matrix elements
    int opt_lanczos_keep_gb = 160;    // ovqe_ground_state keeps its Lanczos vectors in HBM up to this many GB (one pass); 0 = always two passes
    int opt_sector_tile_cap = 6500;   // amplitudes per circuit tile (up to 14000 for energies; gradients on the tables hold two tiles in LDS: <= 6500)
    int opt_sector_sparsity = 4;  // the support must be at most 1/this of the register
    int opt_sector_profile = 0;   // 1: HIP events around the circuit and the <H> kernel of every sector evaluation (program_info)
    int opt_sector_debug = 0;     // measurements only (1: circuit sweeps without their ops — wrong results)
    int opt_sector_stream_arrange = 1; // third sweep form: lanes of a row chosen for the LDS banks (0: in list order; testing builds)
    int opt_sector_stream_waves = 0;   // third sweep form: waves that share a tile's rows (0: from the pairs per op of the sweep's largest tile); testing builds
    int opt_sector_sweep = 3;     // circuit sweep kernel: 3 = per-wave streams, barriers at run boundaries only (k_sector_sweep3; built on the tables of 2); 2 = scatter-on-write, pair words in registers (k_sector_sweep2); 1 = first form
    int opt_sector_chunk = 2048;  // k_sector_sweep2: pair words per chunk = threads x words per thread (1024, 2048, 4096)
    int opt_sector_sweep_dbg = 0; // measurements only, k_sector_sweep2: 1 no ops, 2 empty kernel, 3 loads only — wrong results
    int opt_sector_h_pack = 1;     // <H> tables: coded words of a sweep with at most 1023 magnitudes stored as 24-bit elements (0: 32-bit words; testing builds)
    int opt_sector_h_groups = 256; // workgroups per <H> sweep (they share the sweep's tiles round robin)
    int opt_sector_h_dbg = 0;     // measurements only, k_sector_expect: 1 tile loads only, 2 no tile loads, 3 metadata only — wrong results
    int opt_sector_adjoint = 3;   // backward sweeps of the gradient: 3 = on the per-wave streams (k_sector_adjoint3) where a sweep has them; 2 = on the 64-bit tables (k_sector_adjoint2) where they exist and fit; 1 = first form
    int opt_sector_apply_threads = 0; // threads per workgroup of k_sector_apply (0 = automatic, 512, 1024)
    int opt_sector_h_threads = 512; // threads per workgroup of k_sector_expect (512 or 1024)
    int opt_sector_batch_sweep_threads = 512;    // workgroup size of a batch's circuit sweeps (512, 1024)
    int opt_sector_batch_dst_lds = 0;            // their scatter indices staged in LDS (0: read when the tile is written — 44 instead of 64 KB
                                                 // per workgroup at 24 qubits: three 512-thread workgroups per CU; B = 64: 0.82 -> 0.70 ms per evaluation)
    int opt_sector_batch_zfast = 1;              // batched <H>: state group = fastest grid index (the groups share a tile's elements through the caches)
    int opt_sector_batch_nb = 2;      // states per tile of the batched <H> (2 or 3)
    int opt_sector_batch_threads = 1024;   // its workgroup size (512, 1024)
    int opt_screen_tables = 1;        // ADAPT screens over the support list: pattern tables for the pool's same-x runs (PoolRun)
    int opt_screen_sector = 1;        // ADAPT screens: sigma = H psi from the materialised Hamiltonian of psi's symmetry sector (real states)
    int opt_screen_sector_min = 1024; // ... once psi lists at least this many amplitudes
    bool prog_from_gates = false;     // the stored program came from ovqe_set_gate_program (frame form): sector tables at the first evaluation
    int opt_expect_dense = 1;         // tiled <H> of dense complex registers (25+ qubits): census of the first sweep, then two workgroups per CU
    DevBuf d_tile_cnt;
    int opt_tile_unsplit = 1;         // tiled <H> of complex states: groups of one or two terms as unsplit entries (see build_ham_tiles)
    int opt_tile_flat = 2;            // tiled <H>: entries of one or two merged terms as per-LANE items (1), per-wave entries (0), items for real
                                      // states only (2, default: on dense complex tiles the items' LDS reads conflict 16 ways — 78 % of the LDS cycles,
                                      // profiles/r5_tilexp — and the per-wave entries are 10 % faster once two workgroups share a CU)
    int opt_sector_eager_rots = 2048; // programs of at most this many rotations build their sector tables at the FIRST evaluation (else the second)
    int opt_sector_regular = 1;       // supports that are a full coset of the program's Z2 symmetries: sweeps from bit arithmetic, no pair words (k_sector_sweep_reg); 2: such engines build no pair tables at all (energies only)
    int opt_sector_reg_threads = 256; // workgroup size of those sweeps
    int opt_sector_reg_adjoint = 1;   // ovqe_energy_gradient on a regular support: backward sweeps from bit arithmetic too (0: pair-word sweeps)
    int opt_sector_reg_pairs = 1;     // two consecutive three-bit ops that share two bits run as one 16-slot block
    int opt_sector_apply_seq = 1;     // lambda = H psi on the sector tables: one launch per sweep in sequence, plain additions (0: one launch, global atomics)
    int opt_sector_coset_first = 1;   // gate lists in frame form: the coset of their Z2 symmetries as support, no probe run (checked afterwards)
    int opt_sector_reg_runs = 1;      // runs of consecutive ops whose waves stay inside their own slots: no barrier inside a run
    int opt_sector_depth2 = 1;        // first form of the sweeps: two chunks of pair words ahead where every op of a tile fits a staging buffer
    int opt_sector_many_tiles = 1;    // single evaluations with >= 768 tiles: the workgroup shape of the batches (512 threads, scatter indices from memory)
    int opt_sector_h_lpt = 1;         // <H> kernels take the tiles of a sweep largest first
    int opt_sector_batch = 1;     // ovqe_energy_batch on the sector tables: whole batches per pass (0: one evaluation at a time)
    float last_batch_ms = 0.f;
    const double *cur_theta = nullptr;  // device pointers of the batch being evaluated
    double *cur_energies = nullptr;
    // options
    int opt_force_path = 0;       // 0 auto, 1 small kernel, 2 streaming kernels
    int opt_small_max = 14;       // always-small up to this many qubits
    int opt_small_batch_max = 16; // small kernel for batches up to this many qubits
    int opt_unroll = 4;
    int opt_index_streams = 1;    // precompute the pair-index streams of OP_TAB ops on the host
    int opt_table_fusion = 1;     // turn commuting same-x runs into single sparse pair rotations (OP_TAB)
    int opt_rot_variant = 0;      // tuning variant of the streaming pair sweep (0 = default kernel)
    int64_t last_passes = 0;      // passes over the state buffer (kernel launches that stream it) of the last ovqe_apply_pauli_rotations /
    int64_t last_pass_bytes = 0;  // ovqe_bilinear call and the bytes they move by construction (bench.py: the sharded block's real traffic)
    int64_t last_fci_rounds = 0;  // matvec rounds the last ovqe_sector_ground_state needed to saturate the block of |hf>
    int fault_inject = 0;         // option "fault_inject" (tests of the ABI's exception barrier): 1 = the next term-list build throws std::bad_alloc
    int opt_persist_blocks = 2048;
    int opt_small_threads = 0;    // 0: automatic; 256/512/1024: workgroup size of the fused kernel
    int opt_real_mode = 1;        // allow the real-amplitude specialisation of the fused kernel
    // LDS-tiled multi-op sweeps of the streaming path (sv_tile.hpp)
    int opt_tile_bits = -1;       // -1: automatic (12 when the state streams from HBM, n >= 25; else 11);
                                  // 0: one sweep per op; 10..12: tile size 2^bits amplitudes
    double2 init_amp = make_double2(1.0, 0.0);  // amplitude of |hf> (global phase of a folded Clifford part)
    int opt_clifford_frame = 1;   // gate programs: 0 literal, 1 Clifford-frame form when the frame closes, 2 forced
    int opt_tile_low = 4;         // lowest index bits always inside the tile (contiguous 16 B << low chunks)
    int opt_ham_tile_low = 2;     // the same for the tile cover of the Hamiltonian (-1: opt_tile_low): fewer forced bits = fewer
                                  // sweeps per H psi / <H> (N2/cc-pVDZ at 24 qubits: 102 sweeps at 4; 25.4 ms per H psi at 2, 30.1 at 4)
    TilePlan tp;                  // of the stored program
    TilePlan tp_real;             // same program on a real-amplitude state (built on first use)
    bool tp_real_built = false;
    bool prog_real_ok = false;    // every rotation has an odd number of Y and there is no diagonal run
    int opt_apply_min_tiles = 256;  // H psi goes through the tile cover from this many tiles on
    int opt_real_stream = 1;      // streaming energies of such programs keep the state as 2^n doubles
    int opt_expect_sparse = 4;    // tiled <H>: a tile with at most 1/den of its amplitudes non-zero is evaluated over the
                                  // compacted list of those amplitudes (0 = always the dense entry walks)
    HamDev ham_real;              // tile cover of the stored Hamiltonian for the real-amplitude state
    TilePlan tp_adhoc;            // of the rotation list of the current ovqe_apply_pauli_rotations call
    std::vector<CrossSum *> xsums;   // ovqe_xsum_create (slots of destroyed sums are nullptr)
};

namespace {

int rebuild_small_program(ovqe_handle h);
int build_tile_program(ovqe_handle h);
bool mapped_io(ovqe_handle h, int64_t B);

int fail(ovqe_handle h, int code, const std::string &msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

// The exception barrier of the C ABI (include/ovqe_sv.h: "No C++ exception crosses the ABI"): every extern "C" entry
// point is a function-try-block that ends in OVQE_CATCH — a host std::bad_alloc in a table build, a std::length_error of
// an absurd size, anything a library throws becomes a negative status + text on the handle instead of std::terminate in
// the caller's process.
int translate_exception(ovqe_handle h) noexcept {
    int code = OVQE_ERR_INVALID;
    try {
        try {
            throw;
        } catch (const std::bad_alloc &) {
            code = OVQE_ERR_ALLOC;
            fail(h, code, "host allocation failed (std::bad_alloc)");
        } catch (const std::exception &e) {
            fail(h, code, std::string("C++ exception stopped at the ABI: ") + e.what());
        } catch (...) {
            fail(h, code, "unknown C++ exception stopped at the ABI");
        }
    } catch (...) {   // the message itself could not be stored
    }
    return code;
}
#define OVQE_CATCH(h) catch (...) { return translate_exception(h); }

// every entry point runs on its handle's device, whatever the caller's current device is
#define OVQE_ENTER(h)                        \
    do {                                     \
        if (h) {                             \
            (void)hipSetDevice((h)->device); \
            (h)->nz_super_prev = (h)->nz_super; \
            (h)->nz_super = false;           \
        }                                    \
    } while (0)

#define HIPC(h, call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(h, OVQE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

void free_hamdev(HamDev &H) {
    for (DevBuf *b : {&H.d_groups, &H.d_terms, &H.d_tchunks, &H.d_tgroups, &H.d_tterms, &H.d_tflats, &H.d_titems, &H.d_rest, &H.d_achunks,
                      &H.d_agroups, &H.d_aterms, &H.d_dzin, &H.d_doff, &H.d_dterms})
        if (b->p) {
            (void)hipFree(b->p);
            *b = DevBuf{};
        }
}

void free_cross_sum(CrossSum *X) {
    if (!X) return;
    free_hamdev(X->local);
    for (int f = 0; f < 2; ++f)
        for (CrossCover &C : X->partners[f])
            for (DevBuf *b : {&C.d_achunks, &C.d_agroups, &C.d_aterms, &C.d_groups, &C.d_terms})
                if (b->p) (void)hipFree(b->p);
    if (X->d_part.p) (void)hipFree(X->d_part.p);
    delete X;
}

void release_block(void *p, size_t cap) {
    if (!p) return;
    if (DevBlockCache::current && DevBlockCache::current->blocks.size() < 512) DevBlockCache::current->blocks.push_back(DevBuf{p, cap});
    else (void)hipFree(p);
}

// "poll_result" (default on): a lone evaluation's host watches the mapped slot its last kernel writes instead of waiting for the
// stream's completion signal — the store to host-coherent memory lands before the end-of-kernel processing the signal waits for
// (6 us per call: H2O 31 -> 25 us = 40 k evaluations/s).  The slot holds OVQE_POLL_SENTINEL (a NaN no evaluation produces) before the
// launch.  false: nothing arrived within 5 ms — the caller synchronises the stream after all and whatever went wrong surfaces there.
constexpr uint64_t OVQE_POLL_SENTINEL = 0x7ff8dead0badf00dull;
inline bool poll_mapped_slot(volatile uint64_t *slot) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0; *slot == OVQE_POLL_SENTINEL; ++spins) {
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) return false;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

// B result slots (a small batch through the mapped buffer): every slot written exactly once by its evaluation's kernel
inline void poll_arm(double *slots, int64_t B) {
    for (int64_t b = 0; b < B; ++b) std::memcpy(slots + b, &OVQE_POLL_SENTINEL, sizeof(uint64_t));
}
inline bool poll_mapped_slots(double *slots, int64_t B) {
    for (int64_t b = 0; b < B; ++b)
        if (!poll_mapped_slot(reinterpret_cast<volatile uint64_t *>(slots + b))) return false;
    return true;
}

int ensure(ovqe_handle h, DevBuf &b, size_t bytes) {
    if (b.cap >= bytes && b.p) return OVQE_OK;
    release_block(b.p, b.cap);
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes, 256);
    if (DevBlockCache::current) {
        // inside a build, blocks below 8 MB come in size classes (2^k x 1, 1.25, 1.5, 1.75): the per-sweep temporaries differ by a few
        // percent from sweep to sweep and would never meet a kept block of their exact size
        if (want < ((size_t)8 << 20)) {
            size_t base = 256;
            while (base * 2 <= want) base *= 2;
            const size_t step = base / 4;
            want = base + ((want - base + step - 1) / step) * step;
        }
        size_t cap = 0;
        if (void *p = DevBlockCache::current->take(want, &cap)) {
            b.p = p;
            b.cap = cap;
            return OVQE_OK;
        }
    }
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess && DevBlockCache::current && !DevBlockCache::current->blocks.empty()) {   // give the kept blocks back and try again
        (void)hipGetLastError();
        DevBlockCache::current->flush();
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) return fail(h, OVQE_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    b.cap = want;
    return OVQE_OK;
}

int ensure_scratch(ovqe_handle h, int k) {
    if (h->scratch[k]) return OVQE_OK;
    hipError_t e = hipMalloc((void **)&h->scratch[k], h->namps * sizeof(amp_t));
    if (e != hipSuccess) return fail(h, OVQE_ERR_ALLOC, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
    return OVQE_OK;
}

inline uint64_t local_mask(ovqe_handle h) { return h->namps - 1ull; }
inline int reduce_blocks(uint64_t namps) {
    return (int)std::min<uint64_t>(2048, std::max<uint64_t>(1, (namps + 255) / 256));
}

// sort terms by x (stable), fold i^ny into the coefficient, build per-x groups
int build_groups(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *cr, const double *ci,
                 bool allow_global_x, std::vector<HGroup> &groups, std::vector<HTerm> &terms,
                 std::vector<uint64_t> *xs_out = nullptr, std::vector<int64_t> *perm_out = nullptr) {
    const uint64_t lmask = local_mask(h);
    const int ntot = h->n_local + h->n_global;
    const uint64_t allmask = ntot >= 64 ? ~0ull : ((1ull << ntot) - 1ull);
    if (h->fault_inject == 1) {   // what a failed host allocation of the vectors below does
        h->fault_inject = 0;
        throw std::bad_alloc();
    }
    std::vector<int64_t> order(T);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return x[a] < x[b]; });
    groups.clear();
    terms.clear();
    terms.reserve(T);
    for (int64_t oi = 0; oi < T; ++oi) {
        const int64_t t = order[oi];
        if ((x[t] | z[t]) & ~allmask) return fail(h, OVQE_ERR_INVALID, "Pauli mask has bits beyond the register");
        if (!allow_global_x && (x[t] & ~lmask))
            return fail(h, OVQE_ERR_INVALID,
                        "x mask touches global (rank) bits: exchange shards first (openvqe_amd/distributed.py)");
        const int ny = __builtin_popcountll(x[t] & z[t]) & 3;
        const double a = cr[t], b = ci ? ci[t] : 0.0;
        HTerm ht;
        ht.z = z[t];
        switch (ny) {  // (a + ib) * i^ny
        case 0: ht.cr = a; ht.ci = b; break;
        case 1: ht.cr = -b; ht.ci = a; break;
        case 2: ht.cr = -a; ht.ci = -b; break;
        default: ht.cr = b; ht.ci = -a; break;
        }
        if (groups.empty() || x[order[oi - 1]] != x[t]) {
            HGroup g;
            g.x = x[t] & lmask;
            g.jbase = (h->base ^ x[t]) & ~lmask;
            g.t0 = (int32_t)terms.size();
            g.t1 = g.t0;
            g.tiny = 0.0;
            groups.push_back(g);
        }
        terms.push_back(ht);
        groups.back().t1 = (int32_t)terms.size();
        groups.back().tiny += 64.0 * 2.220446049250313e-16 * (std::fabs(ht.cr) + std::fabs(ht.ci));
        if (xs_out) xs_out->push_back(x[t] & lmask);
    }
    if (perm_out) *perm_out = order;
    return OVQE_OK;
}

int upload(ovqe_handle h, DevBuf &b, const void *src, size_t bytes) {
    int rc = ensure(h, b, bytes);
    if (rc) return rc;
    if (bytes) HIPC(h, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
}

// ---- streaming-path launches -------------------------------------------------------------------
// Launch geometry of the streaming sweeps, from measurements on MI355X (tools/exp_rot*.py, profiles/):
// one pair (resp. amplitude) per thread and small workgroups win — more independent workgroups in flight
// beat more loads in flight per thread — and non-temporal accesses help exactly when the state is far
// beyond the 256 MiB Infinity Cache:
//   n_local >= 25 : non-temporal; 64-thread groups when the pivot bit >= 7, else 128-thread groups
//   21..24        : 256-thread groups, cached accesses (the state lives in the Infinity Cache)
//   14..20        : 64-thread groups (L2-resident)
//   smaller       : legacy multi-pair kernels (launch-bound anyway)
// "rot_variant" > 0 forces one geometry (experiments); -1 forces the legacy kernel.
int launch_rot_run(ovqe_handle h, uint64_t x, const RotParam *d_rp, int nrot) {
    if (nrot <= 0) return OVQE_OK;
    const int nl = h->n_local;
    int variant = h->opt_rot_variant;
    if (x == 0) {
        const uint64_t n = h->namps;
#define OVQE_LAUNCH_D(NT, U, NTL)                                                                                \
    hipLaunchKernelGGL((k_rot_diag_v<NT, U, NTL>), dim3((unsigned)((n + (uint64_t)NT * U - 1) / ((uint64_t)NT * U))), \
                       dim3(NT), 0, h->stream, h->state, n, h->base, d_rp, nrot);
        if (variant == 0 && nl >= 14) variant = nl >= 32 ? 105 : (nl >= 25 ? 100 : (nl >= 21 ? 108 : 100));
        if (nl < 14 || variant < 100) variant = (variant == -1 || nl < 14) ? -1 : 100;
        switch (variant) {
        case 100: OVQE_LAUNCH_D(64, 1, true) break;
        case 101: OVQE_LAUNCH_D(128, 1, true) break;
        case 102: OVQE_LAUNCH_D(256, 1, true) break;
        case 104: OVQE_LAUNCH_D(128, 2, true) break;
        case 105: OVQE_LAUNCH_D(64, 4, true) break;  // n >= 32: a launch holds fewer than 2^32 threads
        case 108: OVQE_LAUNCH_D(256, 1, false) break;
        default:
            if (h->opt_unroll >= 4 && n >= 256u * 4u) {
                hipLaunchKernelGGL(k_rot_diag<4>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, h->stream, h->state,
                                   n, h->base, d_rp, nrot);
            } else {
                hipLaunchKernelGGL(k_rot_diag<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->state,
                                   n, h->base, d_rp, nrot);
            }
        }
#undef OVQE_LAUNCH_D
    } else {
        const uint64_t np = h->namps >> 1;
        const int pivot = 63 - __builtin_clzll(x);
#define OVQE_LAUNCH_V(NT, U, NTL, PERSIST)                                                                          \
    {                                                                                                              \
        const uint64_t ntiles = (np + (uint64_t)NT * U - 1) / ((uint64_t)NT * U);                                  \
        const unsigned grid = (unsigned)(PERSIST ? std::min<uint64_t>(ntiles, (uint64_t)h->opt_persist_blocks) : ntiles); \
        hipLaunchKernelGGL((k_rot_pairs_v<NT, U, NTL, PERSIST>), dim3(grid), dim3(NT), 0, h->stream, h->state, np,  \
                           pivot, x, h->base, d_rp, nrot);                                                         \
    }
        if (variant == 0 && nl >= 14)
            variant = nl >= 32 ? 21 : (nl >= 25 ? (pivot >= 7 ? 16 : 13) : (nl >= 21 ? 17 : 16));
        if (nl < 14 || variant >= 100) variant = -1;
        switch (variant) {
        case 1: OVQE_LAUNCH_V(256, 4, true, false) break;
        case 4: OVQE_LAUNCH_V(256, 4, false, true) break;
        case 8: OVQE_LAUNCH_V(256, 2, true, false) break;
        case 12: OVQE_LAUNCH_V(256, 1, true, false) break;
        case 13: OVQE_LAUNCH_V(128, 1, true, false) break;
        case 14: OVQE_LAUNCH_V(512, 1, true, false) break;
        case 16: OVQE_LAUNCH_V(64, 1, true, false) break;
        case 17: OVQE_LAUNCH_V(256, 1, false, false) break;
        case 19: OVQE_LAUNCH_V(64, 2, true, false) break;
        case 21: OVQE_LAUNCH_V(64, 4, true, false) break;  // n >= 32: a launch holds fewer than 2^32 threads
        default:
            if (h->opt_unroll >= 4 && np >= 256u * 4u) {
                hipLaunchKernelGGL(k_rot_pairs<4>, dim3((unsigned)((np + 1023) / 1024)), dim3(256), 0, h->stream,
                                   h->state, np, pivot, x, h->base, d_rp, nrot);
            } else if (h->opt_unroll >= 2 && np >= 256u * 2u) {
                hipLaunchKernelGGL(k_rot_pairs<2>, dim3((unsigned)((np + 511) / 512)), dim3(256), 0, h->stream, h->state,
                                   np, pivot, x, h->base, d_rp, nrot);
            } else {
                hipLaunchKernelGGL(k_rot_pairs<1>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, h->stream, h->state,
                                   np, pivot, x, h->base, d_rp, nrot);
            }
        }
#undef OVQE_LAUNCH_V
    }
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

int launch_gate(ovqe_handle h, int kind, int b0, int b1, amp_t *st = nullptr) {
    if (!st) st = h->state;
    const uint64_t nwork = kind == 2 ? (h->namps >> 2) : (h->namps >> 1);
    if (nwork == 0) return fail(h, OVQE_ERR_INVALID, "register too small for this gate");
    if (nwork >= 1024) {
        hipLaunchKernelGGL(k_gate<4>, dim3((unsigned)((nwork + 1023) / 1024)), dim3(256), 0, h->stream, st, nwork,
                           kind, b0, b1);
    } else {
        hipLaunchKernelGGL(k_gate<1>, dim3((unsigned)((nwork + 255) / 256)), dim3(256), 0, h->stream, st, nwork,
                           kind, b0, b1);
    }
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

inline RotParam make_rot(uint64_t x, uint64_t z, double phi) {
    RotParam r;
    const int ny = __builtin_popcountll(x & z) & 3;
    r.z = z;
    r.c = std::cos(phi);
    const double s = std::sin(phi);
    r.s = (ny & 2) ? -s : s;
    r.odd = ny & 1;
    r.pad = 0;
    return r;
}

int ensure_rp(ovqe_handle h, size_t n) {
    if (h->h_rp_cap < n) {
        if (h->h_rp) (void)hipHostFree(h->h_rp);
        h->h_rp = nullptr;
        size_t cap = std::max<size_t>(n, 1024);
        hipError_t e = hipHostMalloc((void **)&h->h_rp, cap * sizeof(RotParam), hipHostMallocDefault);
        if (e != hipSuccess) return fail(h, OVQE_ERR_ALLOC, "hipHostMalloc rotation table");
        h->h_rp_cap = cap;
    }
    return ensure(h, h->d_rp, n * sizeof(RotParam));
}

// sum over [bra|P|ket] groups -> complex result on host
int run_bilinear(ovqe_handle h, const amp_t *bra, const amp_t *ket, const std::vector<HGroup> &groups,
                 const HGroup *d_groups, const HTerm *d_terms, double2 *out, bool hermitian_expectation = false) {
    const int nb = reduce_blocks(h->namps);
    const int G = (int)groups.size();
    if (G == 0) {
        *out = make_double2(0.0, 0.0);
        return OVQE_OK;
    }
    // chunk the group loop so that one launch streams at most ~64 GiB
    hermitian_expectation = hermitian_expectation && bra == ket;  // x is local whenever bra == ket (same shard)
    const double bytes_per_group = 16.0 * (double)h->namps;   // (k_bilinear: + 16 B per amplitude and launch for the bra)
    int per_launch = (int)std::max(1.0, std::min((double)G, 6.4e10 / bytes_per_group));
    const int nchunks = (G + per_launch - 1) / per_launch;
    int rc = ensure(h, h->d_partials, (size_t)nchunks * nb * sizeof(double2));
    if (rc) return rc;
    rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    h->last_passes = nchunks;
    h->last_pass_bytes = (int64_t)(bytes_per_group * (double)G + (hermitian_expectation ? 0.0 : 16.0 * (double)h->namps * nchunks));
    for (int c = 0; c < nchunks; ++c) {
        const int g0 = c * per_launch, g1 = std::min(G, g0 + per_launch);
        if (hermitian_expectation) {
            hipLaunchKernelGGL(k_expect_pairs, dim3(nb), dim3(256), 0, h->stream, ket, h->namps, d_groups, g0, g1,
                               d_terms, (double2 *)h->d_partials.p + (size_t)c * nb);
        } else {
            hipLaunchKernelGGL(k_bilinear, dim3(nb), dim3(256), 0, h->stream, bra, ket, h->namps, d_groups, g0, g1,
                               d_terms, (double2 *)h->d_partials.p + (size_t)c * nb);
        }
    }
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)h->d_partials.p,
                       (int64_t)nchunks * nb, (double2 *)h->d_result.p, 0);
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    *out = h->h_result[0];
    return OVQE_OK;
}

inline int tile_bits(ovqe_handle h, bool real = false) {
    const int m = h->opt_tile_bits >= 0 ? h->opt_tile_bits : (h->n_local >= 25 ? 12 : 11);
    return (real && m >= 10) ? m + 1 : m;  // the same LDS bytes hold twice the real amplitudes
}
inline bool tile_ok(ovqe_handle h, bool real) {
    const int m = tile_bits(h, real);
    return m >= (real ? 11 : 10) && m <= (real ? 13 : 12) && h->n_local >= m + 2 && h->opt_tile_low >= (real ? 1 : 0) &&
           h->opt_tile_low <= 8;
}

// lowest index bits forced into every tile of the Hamiltonian's cover
inline int ham_tile_low(ovqe_handle h, bool real) {
    const int l = h->opt_ham_tile_low >= 0 ? h->opt_ham_tile_low : h->opt_tile_low;
    return std::min(8, std::max(l, real ? 1 : 0));   // (a real amplitude is 8 bytes: at least 16-byte chunks)
}

inline uint32_t extract_bits(uint64_t v, uint64_t mask) {  // pext
    uint32_t r = 0;
    int k = 0;
    for (uint64_t mk = mask; mk; mk &= mk - 1ull, ++k)
        if ((v >> __builtin_ctzll(mk)) & 1ull) r |= 1u << k;
    return r;
}

#include "tile_host.inc"


void tridiag_lowest(const std::vector<double> &a, const std::vector<double> &b, int m, double *lam, std::vector<double> &s);

#include "sector_host.inc"

// sector path: the tables of a (program, Hamiltonian) pair are built at its second evaluation (energy or gradient), so
// one-shot callers never pay for them
int sector_prepare(ovqe_handle h, bool eager = false) {
    if (!h->opt_sector || h->n_local < h->opt_sector_min_qubits) return OVQE_OK;
    SectorEngine &E = h->sec;
    if (E.prog_version != h->prog_version || E.ham_version != h->ham.version) {
        // an ansatz that extends its predecessor whose support needed the independent-angle probe (sector_orphaned) will need it too:
        // straight to that probe (N2 fermionic ADAPT from the 20th operator on: one build of ~10 ms and one dense <H> of 8 ms saved per
        // macro-iteration)
        const bool independent = E.valid && E.probe_mode == 1 && h->prog_extends_prev && E.ham_version == h->ham.version;
        free_sector_kept(h, E);
        E.disabled = false;
        E.seen = 0;
        E.probe_mode = independent ? 1 : 0;
        E.coset_rejected = false;
        E.prog_version = h->prog_version;
        E.ham_version = h->ham.version;
    }
    // eager: a gradient call — the dense-state adjoint pass costs more than building the tables (24 qubits: 0.6 s against
    // 0.26 s), and whoever asks for gradients evaluates many times
    // ... and a SHORT program (an ADAPT ansatz: a few hundred rotations) builds its tables in about the time of the one dense
    // evaluation they would wait for (24 qubits, 16 spin-adapted generators: 25 ms against 33 ms): at once
    // ... and a GATE LIST in frame form (ovqe_set_gate_program: the reference's QUCCSD templates, which only get_energy_qucc's two
    // minimisations ever submit — thousands of evaluations, ref:openvqe/ucc_family/get_energy_qucc.py:158-175): the dense evaluation the
    // tables would wait for costs 47 ms at 24 qubits and is saved (time to the first energy from the tables 229 -> 182 ms)
    const int wait = (h->srots.size() <= (size_t)h->opt_sector_eager_rots || h->prog_from_gates) ? 1 : 2;
    if (!E.valid && !E.disabled && (++E.seen >= wait || eager)) return build_sector(h);
    return OVQE_OK;
}

// An evaluation found a non-zero amplitude whose partner is outside the probed support.  The support was taken from the FINAL
// probe state; rotations that share a parameter (the strings of a spin-adapted generator) can pass through determinants whose
// amplitudes cancel again by the end of the program — absent from the list, present in between.  Second attempt: probe with an
// independent angle per rotation (nothing cancels: every determinant the program can touch is listed; a superset is still exact).
// A program that fails that too is left to the dense kernels.
void sector_orphaned(ovqe_handle h) {
    SectorEngine &E = h->sec;
    const int mode = E.probe_mode;
    free_sector_kept(h, E);
    if (mode == 0) {
        E.probe_mode = 1;
        E.disabled = false;
        E.seen = 1;   // rebuilt at the next evaluation
    } else {
        E.disabled = true;
    }
}

// commuting-run fusion analysis of one same-x run (see sv_small.hpp OP_TAB); returns false when the run
// does not have the structure (then it stays a sequential OP_PAIR)
bool try_table_op(ovqe_handle h, const SmallOp &op, SmallOp &out, std::vector<SmallRot> &entries) {
    if (op.kind != OP_PAIR || op.count < 1) return false;
    const uint64_t x = op.x;
    const int w = __builtin_popcountll(x);
    if (w > 7 || h->n_local > 32) return false;
    const SmallRot &r0 = h->rots[op.first];
    if (r0.pidx < 0) return false;
    const uint64_t zc = r0.z & ~x;
    for (int r = op.first; r < op.first + op.count; ++r) {
        const SmallRot &sr = h->rots[r];
        if (sr.pidx != r0.pidx || sr.phi0 != 0.0 || !(sr.ny & 1) || (sr.z & ~x) != zc) return false;
    }
    int pos[8], np = 0;
    for (int b = 0; b < 64; ++b)
        if ((x >> b) & 1) pos[np++] = b;
    out = op;
    out.kind = OP_TAB;
    out.zc = (uint32_t)zc;
    out.fixmask = (uint32_t)x;
    out.first = (int32_t)entries.size();
    out.count = 0;
    out.stream = -1;
    out.lognk = (uint32_t)(h->n_local - w);
    const bool want_stream = h->opt_index_streams && h->n_local <= 15 && h->n_local - w >= 0;
    const int32_t stream0 = (int32_t)h->idx_stream.size();
    // patterns over the non-pivot x bits (the pivot bit of i is 0)
    for (uint32_t e = 0; e < (1u << (w - 1)); ++e) {
        uint64_t ibits = 0;
        for (int f = 0; f < w - 1; ++f)
            if ((e >> f) & 1) ibits |= 1ull << pos[f];
        // rotation angle of the pair = chainsign(i) * theta * K_e,
        // K_e = sum_t -(-1)^{parity(ibits & z_t)} coeff_t (ny&2 ? -1 : 1)     [odd ny]
        double K = 0.0;
        for (int r = op.first; r < op.first + op.count; ++r) {
            const SmallRot &sr = h->rots[r];
            double c = (sr.ny & 2) ? -sr.coeff : sr.coeff;
            if (!(__builtin_popcountll(ibits & sr.z) & 1)) c = -c;
            K += c;
        }
        if (K == 0.0) continue;  // structurally untouched pairs
        SmallRot pe;
        pe.z = ibits;
        pe.coeff = K;
        pe.phi0 = 0.0;
        pe.pidx = r0.pidx;
        pe.ny = 1;
        entries.push_back(pe);
        out.count++;
        if (want_stream) {
            const uint64_t nk = 1ull << (h->n_local - w);
            for (uint64_t k = 0; k < nk; ++k) {
                uint64_t i = k;
                for (int f = 0; f < w; ++f) {  // deposit: zeros at the x positions (ascending)
                    const uint64_t low = (1ull << pos[f]) - 1ull;
                    i = ((i & ~low) << 1) | (i & low);
                }
                i |= ibits;
                const uint16_t sgn = (__builtin_popcountll(i & zc) & 1) ? 0x8000u : 0u;
                h->idx_stream.push_back((uint16_t)(i | sgn));
            }
        }
    }
    if (want_stream && out.count > 0) out.stream = stream0;
    return true;
}

int rebuild_small_program(ovqe_handle h) {
    const int cap = h->cs_capacity;
    h->sp_tried = false;
    h->sops.clear();
    h->srots.clear();
    h->sop_src.clear();
    h->sop_zc.clear();
    h->idx_stream.clear();
    auto push_sop = [&](const SmallOp &op, int src, uint64_t zc) {
        h->sops.push_back(op);
        h->sop_src.push_back(src);
        h->sop_zc.push_back(zc);
    };
    for (int oi = 0; oi < (int)h->ops.size(); ++oi) {
        const SmallOp &op = h->ops[oi];
        if (op.kind == OP_PAIR || op.kind == OP_DIAG) {
            SmallOp t;
            if (h->opt_table_fusion && try_table_op(h, op, t, h->srots)) {
                if (t.count > 0) push_sop(t, oi, h->rots[op.first].z & ~op.x);  // count == 0: the run is the identity
                continue;
            }
            for (int o = 0; o < op.count; o += cap) {  // split runs longer than the LDS table
                SmallOp p = op;
                p.first = (int32_t)h->srots.size();
                p.count = std::min(cap, op.count - o);
                for (int r = 0; r < p.count; ++r) h->srots.push_back(h->rots[op.first + o + r]);
                push_sop(p, oi, 0);
            }
        } else {
            push_sop(op, oi, 0);
        }
    }
    h->segs.clear();
    SmallSeg cur = {0, 0, 0, 0};
    bool rot_init = false;  // cur.rot0 is set by the first op of the segment that owns table entries
    for (int o = 0; o < (int)h->sops.size(); ++o) {
        const SmallOp &op = h->sops[o];
        const bool has = op.kind == OP_PAIR || op.kind == OP_DIAG || op.kind == OP_TAB;
        const int nent = op.count;
        bool split = (o - cur.op0 >= SMALL_OPS_CAP);  // ops of a segment are staged in LDS
        if (has && rot_init && op.first + nent - cur.rot0 > cap) split = true;
        if (split && cur.op1 > cur.op0) {
            h->segs.push_back(cur);
            cur = {o, o, 0, 0};
            rot_init = false;
        }
        if (has) {
            if (!rot_init) {
                cur.rot0 = op.first;
                cur.rot1 = op.first + nent;
                rot_init = true;
            } else {
                cur.rot1 = std::max(cur.rot1, op.first + nent);
            }
        }
        cur.op1 = o + 1;
    }
    if (cur.op1 > cur.op0) h->segs.push_back(cur);
    int rc = upload(h, h->d_ops, h->sops.data(), h->sops.size() * sizeof(SmallOp));
    if (rc) return rc;
    rc = upload(h, h->d_rots, h->srots.data(), h->srots.size() * sizeof(SmallRot));
    if (rc) return rc;
    rc = upload(h, h->d_rots_seq, h->rots.data(), h->rots.size() * sizeof(SmallRot));
    if (rc) return rc;
    rc = upload(h, h->d_stream, h->idx_stream.data(), h->idx_stream.size() * sizeof(uint16_t));
    if (rc) return rc;
    return upload(h, h->d_segs, h->segs.data(), h->segs.size() * sizeof(SmallSeg));
}

int finish_program(ovqe_handle h) {
    int rc = rebuild_small_program(h);
    if (rc) return rc;
    rc = build_tile_program(h);
    if (rc) return rc;
    h->tp_real_built = false;
    h->prog_version++;
    h->prog_real_ok = !h->ops.empty();
    for (const SmallOp &op : h->ops) h->prog_real_ok = h->prog_real_ok && op.kind != OP_DIAG;
    for (const SmallRot &sr : h->rots) h->prog_real_ok = h->prog_real_ok && (sr.ny & 1);
    h->prog_set = true;
    return OVQE_OK;
}

void push_rotation(ovqe_handle h, uint64_t x, uint64_t z, double coeff, double phi0, int32_t pidx) {
    SmallRot sr;
    sr.z = z;
    sr.coeff = coeff;
    sr.phi0 = phi0;
    sr.pidx = pidx;
    sr.ny = __builtin_popcountll(x & z) & 3;
    const int32_t idx = (int32_t)h->rots.size();
    h->rots.push_back(sr);
    const int32_t kind = x ? OP_PAIR : OP_DIAG;
    if (!h->ops.empty()) {
        SmallOp &last = h->ops.back();
        if (last.kind == kind && last.x == x && last.first + last.count == idx) {
            last.count++;
            return;
        }
    }
    SmallOp op = {};
    op.x = x;
    op.kind = kind;
    op.first = idx;
    op.count = 1;
    op.pivot = x ? 63 - __builtin_clzll(x) : 0;
    h->ops.push_back(op);
}

#include "gates_host.inc"

#include "sparse_host.inc"

int check_theta(ovqe_handle h, const double *theta, int32_t K) {
    if (!h->prog_set) return fail(h, OVQE_ERR_STATE, "no program set (ovqe_set_program / ovqe_set_gate_program)");
    if (K != h->K) return fail(h, OVQE_ERR_INVALID, "K does not match the program's parameter count");
    if (K > 0 && !theta) return fail(h, OVQE_ERR_INVALID, "theta is NULL");
    return OVQE_OK;
}

int create_common(int n_local, int n_global, uint64_t shard, int device, ovqe_handle *out, void *adopt = nullptr) {
    if (!out) return fail(nullptr, OVQE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // 33 local qubits (128 GiB) is what the sweep launchers cover (four pairs per thread keep a launch below 2^32
    // threads up to there) and what leaves room for any scratch in 288 GB; larger registers are sharded
    if (n_local < 1 || n_local > 33 || n_global < 0 || n_local + n_global > 64)
        return fail(nullptr, OVQE_ERR_INVALID,
                    "qubit count out of range (1 <= n_local <= 33 per device, total <= 64; shard larger registers)");
    if (n_global < 64 && shard >> n_global) return fail(nullptr, OVQE_ERR_INVALID, "shard index out of range");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, OVQE_ERR_NO_DEVICE, "no HIP device visible: libovqe_sv has no CPU fallback");
    if (device < 0 || device >= count) return fail(nullptr, OVQE_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, OVQE_ERR_NO_DEVICE,
                    std::string("device is not gfx950 (MI355X): ") + prop.gcnArchName + " — code objects are gfx950 only");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, OVQE_ERR_HIP, "hipSetDevice failed");
    ovqe_handle h = new (std::nothrow) ovqe_sv();
    if (!h) return fail(nullptr, OVQE_ERR_ALLOC, "host allocation failed");
    h->n_local = n_local;
    h->n_global = n_global;
    h->device = device;
    h->shard = shard;
    h->namps = 1ull << n_local;
    h->base = shard << n_local;
    if (adopt) {   // a view: the caller's buffer is the state, nothing of that size is allocated here
        h->state = (amp_t *)adopt;
        h->own_state = false;
    } else {
        e = hipMalloc((void **)&h->state, h->namps * sizeof(amp_t));
        if (e != hipSuccess) {
            delete h;
            return fail(nullptr, OVQE_ERR_ALLOC, std::string("hipMalloc state: ") + hipGetErrorString(e));
        }
    }
    if (hipHostMalloc((void **)&h->h_result, 64 * sizeof(double2), hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        if (h->own_state) (void)hipFree(h->state);
        delete h;
        return fail(nullptr, OVQE_ERR_ALLOC, "host staging / event creation failed");
    }
    *out = h;
    return OVQE_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

int ovqe_version(void) { return 100; }

const char *ovqe_last_error(ovqe_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ovqe_device_count(int *count) {
    if (!count) return OVQE_ERR_INVALID;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return OVQE_OK;
}

int ovqe_create(int n_qubits, int device, ovqe_handle *out) try {
    return create_common(n_qubits, 0, 0, device, out);
} OVQE_CATCH(nullptr)

int ovqe_create_shard(int n_local, int n_global, uint64_t shard_index, int device, ovqe_handle *out) try {
    return create_common(n_local, n_global, shard_index, device, out);
} OVQE_CATCH(nullptr)

int ovqe_create_view(int n_qubits, int device, void *dev_ptr, ovqe_handle *out) try {
    if (!dev_ptr) return fail(nullptr, OVQE_ERR_INVALID, "dev_ptr is NULL");
    return create_common(n_qubits, 0, 0, device, out, dev_ptr);
} OVQE_CATCH(nullptr)

int ovqe_destroy(ovqe_handle h) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->own_state && h->state) (void)hipFree(h->state);
    for (int k = 0; k < 2; ++k)
        if (h->scratch[k]) (void)hipFree(h->scratch[k]);
    std::vector<DevBuf *> bufs = {&h->d_tile_cnt, &h->d_partials, &h->d_result, &h->d_rp, &h->d_ops, &h->d_rots, &h->d_rots_seq, &h->d_segs,
                                  &h->d_stream,
                                  &h->d_theta, &h->d_energies, &h->d_workspace, &h->d_egroups, &h->d_eterms, &h->d_echunks,
                                  &h->d_eflat, &h->d_sp_ops, &h->d_sp_rows, &h->d_sp_rows64, &h->d_sp_prim, &h->d_sp_pairs, &h->d_sp_entries, &h->d_pg_off, &h->d_pg_xs, &h->d_pg_terms, &h->d_pg_runs, &h->d_pg_tabs,
                                  &h->d_pg_out, &h->d_pg_part, &h->d_nz_cnt, &h->d_nz_start, &h->d_nz_idx, &h->d_nz_val, &h->d_nz_bitmap, &h->d_exp_groups, &h->d_exp_terms, &h->d_tile_smasks, &h->d_tile_lists, &h->d_tile_counts, &h->cc.d_sup, &h->cc.d_psic, &h->cc.d_loc, &h->cc.d_cid, &h->cc.d_off, &h->cc.d_sweeps};
    for (TilePlan *tp : {&h->tp, &h->tp_adhoc, &h->tp_real}) bufs.insert(bufs.end(), {&tp->d_tops, &tp->d_trots});
    for (HamDev *H : {&h->ham, &h->ham_adhoc, &h->ham_real, &h->ham_conj})
        bufs.insert(bufs.end(), {&H->d_groups, &H->d_terms, &H->d_tchunks, &H->d_tgroups, &H->d_tterms, &H->d_tflats,
                                 &H->d_titems, &H->d_rest, &H->d_achunks, &H->d_agroups, &H->d_aterms, &H->d_dzin, &H->d_doff, &H->d_dterms});
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    free_sector(h->sec);
    free_sector(h->scr);
    for (CrossSum *X : h->xsums) free_cross_sum(X);
    h->kept_blocks.flush();
    if (h->h_rp) (void)hipHostFree(h->h_rp);
    if (h->h_result) (void)hipHostFree(h->h_result);
    if (h->h_io) (void)hipHostFree(h->h_io);
    if (h->h_fin) (void)hipHostFree(h->h_fin);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    delete h;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_set_stream(ovqe_handle h, void *hip_stream) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_ERR_INVALID;
    HIPC(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)hip_stream;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_set_option(ovqe_handle h, const char *name, int64_t value) try {
    OVQE_ENTER(h);
    if (!h || !name) return OVQE_ERR_INVALID;
    const std::string k(name);
    // Keys between `#ifdef OVQE_TESTING` lines exist only in the testing build of this same source (libovqe_sv_testing.so, built by
    // __graft_entry__.build() with -DOVQE_TESTING, loaded through OVQE_LIB by tests/test_gpu_abi.py and the measurement scripts under
    // tools/): fault injection, kernels with phases switched off, superseded kernel forms and launch geometries.  The product
    // library refuses them as unknown options and runs every one of them at its default.
    if (k == "force_path") h->opt_force_path = (int)value;
    else if (k == "real_state") h->opt_real_state = value ? 1 : 0;
    else if (k == "small_max_qubits") h->opt_small_max = (int)value;
    else if (k == "small_batch_max_qubits") h->opt_small_batch_max = (int)value;
#ifdef OVQE_TESTING
    else if (k == "unroll") h->opt_unroll = (int)value;
#endif
    else if (k == "real_mode") {
        h->opt_real_mode = (int)value;
        h->sp_tried = false;
    }
    else if (k == "sparse") {
        h->opt_sparse = (int)value;
        h->sp_tried = false;
    }
    else if (k == "sparse_grad") h->opt_sparse_grad = value ? 1 : 0;
    else if (k == "sparse_renumber") {
        h->opt_sparse_renumber = value ? 1 : 0;
        h->sp_tried = false;
    }
#ifdef OVQE_TESTING
    else if (k == "sparse_spw") h->opt_sparse_spw = (int)value;
    else if (k == "sparse_dbg") h->opt_sparse_dbg = (int)value;
#endif
    else if (k == "clifford_phase_host") h->opt_clifford_phase_host = (int)value;
    else if (k == "index_streams") {
        h->opt_index_streams = (int)value;
        if (h->prog_set) return finish_program(h);
    }
    else if (k == "sector" || k == "sector_bits" || k == "sector_max_gb" || k == "sector_min_qubits" || k == "sector_h" ||
             k == "sector_h_bits" || k == "sector_dict") {
        (k == "sector" ? h->opt_sector : k == "sector_bits" ? h->opt_sector_bits : k == "sector_max_gb" ? h->opt_sector_max_gb
         : k == "sector_h" ? h->opt_sector_h : k == "sector_h_bits" ? h->opt_sector_h_bits : k == "sector_dict" ? h->opt_sector_dict
                                                                                                              : h->opt_sector_min_qubits) = (int)value;
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    }
    else if (k == "sector_threads") h->opt_sector_threads = (value == 0 || value == 64 || value == 512 || value == 1024) ? (int)value : 256;
#ifdef OVQE_TESTING
    else if (k == "sector_h_groups") h->opt_sector_h_groups = (int)value;
    else if (k == "sector_h_dbg") h->opt_sector_h_dbg = (int)value;
#endif
    else if (k == "sector_adjoint") h->opt_sector_adjoint = value == 1 ? 1 : (value == 2 ? 2 : 3);
#ifdef OVQE_TESTING
    else if (k == "sector_apply_threads") h->opt_sector_apply_threads = value == 1024 ? 1024 : (value == 512 ? 512 : 0);
    else if (k == "sector_h_threads") h->opt_sector_h_threads = value == 1024 ? 1024 : 512;
#endif
    else if (k == "sector_batch") h->opt_sector_batch = (int)value;
#ifdef OVQE_TESTING
    else if (k == "sector_h_lpt") h->opt_sector_h_lpt = (int)value;
    else if (k == "sector_many_tiles") h->opt_sector_many_tiles = (int)value;
    else if (k == "sector_depth2") h->opt_sector_depth2 = (int)value;
#endif
    else if (k == "sector_regular") {
        h->opt_sector_regular = (int)value;
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    } else if (k == "sector_reg_pairs") {
        h->opt_sector_reg_pairs = (int)value;
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    } else if (k == "sector_reg_adjoint") {
        h->opt_sector_reg_adjoint = (int)value;   // (0 needs the pair words a large regular support does without: the tables are rebuilt)
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    }
    else if (k == "sector_reg_threads") h->opt_sector_reg_threads = value == 512 ? 512 : (value == 1024 ? 1024 : (value == 128 ? 128 : 256));
#ifdef OVQE_TESTING
    else if (k == "sector_eager_rots") h->opt_sector_eager_rots = (int)value;
#endif
    else if (k == "sector_fused_reduce") h->opt_sector_fused_reduce = (int)value;
    else if (k == "poll_result") h->opt_poll_result = (int)value;
    else if (k == "sector_pairs_form") h->opt_sector_pairs_form = (int)value;
    else if (k == "screen_sector") h->opt_screen_sector = (int)value;
#ifdef OVQE_TESTING
    else if (k == "screen_tables") h->opt_screen_tables = (int)value;
#endif
    else if (k == "screen_sector_min") h->opt_screen_sector_min = (int)value;
#ifdef OVQE_TESTING
    else if (k == "sector_batch_sweep_threads") h->opt_sector_batch_sweep_threads = value == 512 ? 512 : (value == 256 ? 256 : 1024);
    else if (k == "sector_batch_dst_lds") h->opt_sector_batch_dst_lds = (int)value;
    else if (k == "sector_batch_zfast") h->opt_sector_batch_zfast = (int)value;
    else if (k == "sector_batch_nb") h->opt_sector_batch_nb = value == 3 ? 3 : 2;
    else if (k == "sector_batch_threads") h->opt_sector_batch_threads = value == 512 ? 512 : 1024;
    else if (k == "sector_debug") h->opt_sector_debug = (int)value;
    else if (k == "sector_sweep") h->opt_sector_sweep = value == 1 ? 1 : (value == 2 ? 2 : (value == 4 ? 4 : 3));   // (2 on tables built under 3: the second form on the same tables; 4: the streams for batches too)
    else if (k == "sector_sweep_dbg") h->opt_sector_sweep_dbg = (int)value;
    else if (k == "sector_stream_waves") h->opt_sector_stream_waves = (int)value;
    else if (k == "sector_stream_arrange") h->opt_sector_stream_arrange = (int)value;
    else if (k == "sector_h_pack") h->opt_sector_h_pack = (int)value;
#endif
    else if (k == "sector_reg_runs") {   // runs of ops without barriers (planned at build time: the tables are rebuilt)
        h->opt_sector_reg_runs = (int)value;
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    }
#ifdef OVQE_TESTING
    else if (k == "sector_chunk") h->opt_sector_chunk = (value == 1024 || value == 4096) ? (int)value : 2048;
#endif
    else if (k == "sector_profile") h->opt_sector_profile = (int)value;
    else if (k == "sector_sparsity" || k == "sector_tile_cap") {
        (k == "sector_sparsity" ? h->opt_sector_sparsity : h->opt_sector_tile_cap) = (int)value;
        free_sector(h->sec);
        h->sec.disabled = false;
        h->sec.seen = 0;
        h->sec.prog_version = -1;
    }
    else if (k == "lanczos_keep_gb") h->opt_lanczos_keep_gb = (int)value;
    else if (k == "screen_sparse") h->opt_screen_sparse = (int)std::max<int64_t>(0, value);
#ifdef OVQE_TESTING
    else if (k == "rot_variant") h->opt_rot_variant = (int)value;
    else if (k == "tile_flat") h->opt_tile_flat = (int)value;
    else if (k == "tile_unsplit") h->opt_tile_unsplit = (int)value;
    else if (k == "sector_coset_first") h->opt_sector_coset_first = (int)value;
    else if (k == "sector_apply_seq") h->opt_sector_apply_seq = (int)value;
    else if (k == "expect_dense") h->opt_expect_dense = (int)value;
    else if (k == "expect_diag_wht") {
        h->opt_expect_diag_wht = (int)value;
        for (HamDev *H : {&h->ham, &h->ham_adhoc, &h->ham_real, &h->ham_conj}) H->tile_bits = -1;   // covers rebuilt on their next use
        for (CrossSum *X : h->xsums)
            if (X) X->local.tile_bits = -1;
    }
    else if (k == "fault_inject") h->fault_inject = (int)value;
#endif
    else if (k == "real_stream") h->opt_real_stream = (int)value;
    else if (k == "apply_min_tiles") h->opt_apply_min_tiles = (int)value;
    else if (k == "clifford_frame") h->opt_clifford_frame = (int)value;  // applies to the next ovqe_set_gate_program
#ifdef OVQE_TESTING
    else if (k == "ham_tile_low") h->opt_ham_tile_low = (int)value;
#endif
    else if (k == "tile_bits" || k == "tile_low") {
        (k == "tile_bits" ? h->opt_tile_bits : h->opt_tile_low) = (int)value;
        h->tp_real_built = false;  // the real-amplitude plan follows on its next use
        if (h->prog_set) return build_tile_program(h);
    }
#ifdef OVQE_TESTING
    else if (k == "sparse_dealias") {
        h->opt_sparse_dealias = value ? 1 : 0;
        h->sp_tried = false;
    }
    else if (k == "sparse_wg") h->opt_sparse_wg = value ? 1 : 0;
    else if (k == "sparse_rows") {
        h->opt_sparse_rows = value ? 1 : 0;
        h->sp_tried = false;
    }
    else if (k == "expect_sparse") h->opt_expect_sparse = (int)std::max<int64_t>(0, value);
    else if (k == "expect_streams") h->opt_expect_streams = value >= 2 ? 2 : 1;
    else if (k == "compact_cpp") h->opt_compact_cpp = (int)std::min<int64_t>(4, std::max<int64_t>(1, value));
#endif
    else if (k == "compact") {
        h->opt_compact = value ? 1 : 0;
        h->cc.valid = false;
        h->cc.disabled = false;
        h->cc.seen = 0;
    }
#ifdef OVQE_TESTING
    else if (k == "persist_blocks") h->opt_persist_blocks = (int)value;
    else if (k == "small_threads") {
        h->opt_small_threads = (int)value;
        h->exp_lbits = -1;
    }
#endif
    else if (k == "table_fusion") {
        if (h->opt_table_fusion != (int)value && h->prog_set) {
            h->opt_table_fusion = (int)value;
            h->exp_lbits = -1;
            return finish_program(h);
        }
        h->opt_table_fusion = (int)value;
        h->exp_lbits = -1;
    }
    else return fail(h, OVQE_ERR_INVALID, "unknown option " + k);
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_state_ptr(ovqe_handle h, void **dev_ptr) try {
    OVQE_ENTER(h);
    if (!h || !dev_ptr) return OVQE_ERR_INVALID;
    *dev_ptr = h->state;
    h->state_exposed = true;   // (the caller may write the amplitudes from now on: nothing about them is remembered between calls)
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_adopt_state(ovqe_handle h, void *dev_ptr) try {
    OVQE_ENTER(h);
    if (!h || !dev_ptr) return OVQE_ERR_INVALID;
    HIPC(h, hipStreamSynchronize(h->stream));
    if (h->own_state && h->state) (void)hipFree(h->state);
    h->state = (amp_t *)dev_ptr;
    h->own_state = false;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_init_basis(ovqe_handle h, uint64_t index) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_ERR_INVALID;
    const int ntot = h->n_local + h->n_global;
    if (ntot < 64 && (index >> ntot)) return fail(h, OVQE_ERR_INVALID, "basis index out of range");
    int rc = OVQE_OK;
    if (h->opt_real_state) {   // 2^n_local doubles; the 1.0 lives on the shard whose rank bits match
        const uint64_t lmask = local_mask(h);
        hipLaunchKernelGGL(k_init_basis_real, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, (double *)h->state, h->namps,
                           (index & ~lmask) == h->base ? (index & lmask) : ~0ull);
        HIPC(h, hipGetLastError());
    } else {
        rc = init_basis(h, index);
    }
    if (rc) return rc;
    // the support is this one index: the start of the list the exact exponentials extend (see nz_super)
    bool keep_list = !h->opt_real_state && h->own_state && !h->state_exposed && h->n_global == 0 && h->namps >= 4096 &&
                     h->n_local <= 28 && h->opt_screen_sparse > 0;   // (28 qubits: a list buffer of at most 2^24 indices = 128 MB)
    if (keep_list) {
        if (ensure(h, h->d_nz_idx, (size_t)(h->namps / (uint64_t)h->opt_screen_sparse) * sizeof(uint64_t)) != OVQE_OK) {
            (void)hipGetLastError();   // no room for the list: the exponentials scan the register as before
            keep_list = false;
        } else {
            HIPC(h, hipMemcpyAsync(h->d_nz_idx.p, &index, sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
        }
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    if (keep_list) {
        h->nz_super = true;
        h->nz_super_count = 1;
    }
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_set_state(ovqe_handle h, const double *amps) try {
    OVQE_ENTER(h);
    if (!h || !amps) return OVQE_ERR_INVALID;
    HIPC(h, hipMemcpyAsync(h->state, amps, h->namps * sizeof(amp_t), hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_get_state(ovqe_handle h, double *amps) try {
    OVQE_ENTER(h);
    if (!h || !amps) return OVQE_ERR_INVALID;
    HIPC(h, hipMemcpyAsync(amps, h->state, h->namps * sizeof(amp_t), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_get_amplitudes(ovqe_handle h, int64_t count, const uint64_t *idx, double *amps) try {
    OVQE_ENTER(h);
    if (!h || count < 0 || (count && (!idx || !amps))) return OVQE_ERR_INVALID;
    if (count == 0) return OVQE_OK;
    for (int64_t i = 0; i < count; ++i)
        if (idx[i] >= h->namps) return fail(h, OVQE_ERR_INVALID, "amplitude index out of range");
    DevBuf d_idx, d_out;
    int rc = ensure(h, d_idx, count * sizeof(uint64_t));
    if (!rc) rc = ensure(h, d_out, count * sizeof(amp_t));
    if (!rc) {
        hipError_t e = hipMemcpyAsync(d_idx.p, idx, count * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->stream, h->state, count,
                               (const uint64_t *)d_idx.p, (amp_t *)d_out.p);
            e = hipMemcpyAsync(amps, d_out.p, count * sizeof(amp_t), hipMemcpyDeviceToHost, h->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(h, OVQE_ERR_HIP, std::string("get_amplitudes: ") + hipGetErrorString(e));
    }
    if (d_idx.p) (void)hipFree(d_idx.p);
    if (d_out.p) (void)hipFree(d_out.p);
    return rc;
} OVQE_CATCH(h)

int ovqe_randomize(ovqe_handle h, uint64_t seed, double norm2_total, double *scale_out) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_ERR_INVALID;
    const int nb = reduce_blocks(h->namps);
    int rc = ensure(h, h->d_partials, (size_t)nb * sizeof(double2));
    if (!rc) rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    hipLaunchKernelGGL(k_randomize, dim3(nb), dim3(256), 0, h->stream, h->state, h->namps, h->base, seed, 1.0,
                       (double2 *)h->d_partials.p);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)h->d_partials.p, (int64_t)nb,
                       (double2 *)h->d_result.p, 0);
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    const double n2 = norm2_total > 0.0 ? norm2_total : h->h_result[0].x;
    const double scale = 1.0 / std::sqrt(n2);
    hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, h->stream, h->state, h->namps, scale);
    HIPC(h, hipGetLastError());
    HIPC(h, hipStreamSynchronize(h->stream));
    if (scale_out) *scale_out = scale;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_norm2(ovqe_handle h, double *out) try {
    OVQE_ENTER(h);
    if (!h || !out) return OVQE_ERR_INVALID;
    const int nb = reduce_blocks(h->namps);
    int rc = ensure(h, h->d_partials, (size_t)nb * sizeof(double2));
    if (!rc) rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    // (real state: the 2^n_local doubles read as 2^(n_local - 1) complex numbers have the same sum of squares)
    hipLaunchKernelGGL(k_norm2, dim3(nb), dim3(256), 0, h->stream, h->state, h->opt_real_state ? std::max<uint64_t>(h->namps >> 1, 1) : h->namps,
                       (double2 *)h->d_partials.p);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)h->d_partials.p, (int64_t)nb,
                       (double2 *)h->d_result.p, 0);
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    *out = h->h_result[0].x;
    return OVQE_OK;
} OVQE_CATCH(h)

#include "abi_unit.inc"

#include "abi_eval.inc"

#include "abi_adapt.inc"

#include "abi_solvers.inc"

// ---- measurement support ------------------------------------------------------------------------
int ovqe_time_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi, int warmup, int reps, double *avg_ms) try {
    OVQE_ENTER(h);
    if (!h || !avg_ms || reps <= 0 || warmup < 0) return OVQE_ERR_INVALID;
    if (x & ~local_mask(h)) return fail(h, OVQE_ERR_INVALID, "x mask touches global (rank) bits");
    int rc = ensure_rp(h, 1);
    if (rc) return rc;
    h->h_rp[0] = make_rot(x, z, phi);
    HIPC(h, hipMemcpyAsync(h->d_rp.p, h->h_rp, sizeof(RotParam), hipMemcpyHostToDevice, h->stream));
    for (int i = 0; i < warmup; ++i) {
        rc = launch_rot_run(h, x, (const RotParam *)h->d_rp.p, 1);
        if (rc) return rc;
    }
    HIPC(h, hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) {
        rc = launch_rot_run(h, x, (const RotParam *)h->d_rp.p, 1);
        if (rc) return rc;
    }
    HIPC(h, hipEventRecord(h->ev1, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPC(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *avg_ms = (double)ms / reps;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_program_info(ovqe_handle h, int64_t *info, int count) try {
    OVQE_ENTER(h);
    if (!h || !info || count < 0) return OVQE_ERR_INVALID;
    if (!h->prog_set) return fail(h, OVQE_ERR_STATE, "no program set");
    const bool real_on = h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true);
    const HamDev &HI = (real_on && h->ham_real.version == h->ham.version) ? h->ham_real : h->ham;  // the cover in use
    const TilePlan &TP = (real_on && h->tp_real_built) ? h->tp_real : h->tp;  // the plan the energies use
    int64_t v[16] = {(int64_t)h->ops.size(), (int64_t)h->rots.size(), 0, (int64_t)TP.plan.size(),
                     (int64_t)TP.tsegs.size(), (int64_t)h->sops.size(),
                     !h->sp_tried ? -1 : (h->sp_valid ? (int64_t)h->sp_m : 0),
                     (int64_t)HI.tsweeps.size(), (int64_t)HI.n_rest, HI.tile_entries, HI.tile_terms, HI.tile_work,
                     real_on ? 1 : 0, h->sp_valid ? (int64_t)h->sp_nops : 0, h->sp_valid ? (int64_t)h->sp_npairs : 0,
                     h->sp_valid ? (int64_t)h->sp_nent : 0};
    for (const SmallOp &op : h->ops) v[2] += (op.kind == OP_X || op.kind == OP_H || op.kind == OP_CNOT);
    for (int i = 0; i < count && i < 16; ++i) info[i] = v[i];
    const SectorEngine &E = h->sec;
    const int64_t sv[14] = {E.valid ? (int64_t)E.K : 0, E.valid ? (int64_t)E.segs.size() : 0, E.valid ? (int64_t)E.npairs : 0,
                           E.valid ? (int64_t)E.hs.size() : 0, E.valid ? (int64_t)E.nnz : 0, E.valid ? (int64_t)E.bytes : 0,
                           E.valid ? (int64_t)(1e3 * E.last_circuit_ms) : 0, E.valid ? (int64_t)(1e3 * E.last_expect_ms) : 0,
                           E.valid ? (int64_t)E.h_stream_bytes : 0, E.valid ? (int64_t)E.last_fci_block : 0,
                            h->sp_valid ? h->sp_conflicts_before : 0, h->sp_valid ? h->sp_conflicts_after : 0,
                            (E.valid && E.regular && h->opt_sector_regular) ? (int64_t)E.reg_m : 0,
                            (E.valid && E.regular) ? (int64_t)__builtin_popcount(E.freemask) : 0};
    for (int i = 16; i < count && i < 30; ++i) info[i] = sv[i - 16];
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_get_rotation_program(ovqe_handle h, int64_t capacity, uint64_t *x, uint64_t *z, double *coeff, double *phi0,
                              int32_t *pidx, int64_t *count) try {
    OVQE_ENTER(h);
    if (!h || !count || capacity < 0 || (capacity > 0 && (!x || !z || !coeff || !phi0 || !pidx))) return OVQE_ERR_INVALID;
    if (!h->prog_set) return fail(h, OVQE_ERR_STATE, "no program set");
    if (h->frame_open) return fail(h, OVQE_ERR_STATE, "the program's Clifford frame is open: the rotations alone are not the circuit");
    int64_t n = 0;
    for (const SmallOp &op : h->ops) {
        if (op.kind != OP_PAIR && op.kind != OP_DIAG) return fail(h, OVQE_ERR_STATE, "the program holds literal X / H / CNOT ops");
        for (int32_t r = op.first; r < op.first + op.count; ++r, ++n) {
            if (n >= capacity) continue;
            const SmallRot &sr = h->rots[r];
            x[n] = op.x;
            z[n] = sr.z;
            coeff[n] = sr.coeff;
            phi0[n] = sr.phi0;
            pidx[n] = sr.pidx;
        }
    }
    *count = n;
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_last_support(ovqe_handle h, int32_t which, int64_t *support) try {
    OVQE_ENTER(h);
    if (!h || !support || which < 0 || which > 6) return OVQE_ERR_INVALID;
    const int64_t v[7] = {h->last_screen_support, h->last_exp_support, h->last_screen_sector, h->last_fci_rounds, h->last_passes, h->last_pass_bytes,
                          (int64_t)h->forms_used};
    *support = v[which];
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_last_batch_ms(ovqe_handle h, double *ms) try {
    OVQE_ENTER(h);
    if (!h || !ms) return OVQE_ERR_INVALID;
    *ms = h->last_batch_ms;
    return OVQE_OK;
} OVQE_CATCH(h)

}  // extern "C"

#include "cross_host.inc"
