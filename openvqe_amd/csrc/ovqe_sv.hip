// ovqe_sv.hip — C ABI (include/ovqe_sv.h) + host-side engine of the MI355X statevector backend.
// gfx950 only; no CPU fallback: every entry point needs a live device.
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
// of one build and trims them to 256 MB / 64 blocks on the way out.
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
        DevBlockCache::current->trim((size_t)256 << 20, 64);
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
    int opt_sector_row_banks = 0; // materialised <H>: the elements of every row ordered against LDS bank conflicts (k_sec_row_banks)
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
        if (h) (void)hipSetDevice((h)->device); \
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

// ---- tiled expectation (sv_tile.hpp) -----------------------------------------------------------------------
// Greedy cover of the x-groups by tile bit sets: a set starts from the mandatory low bits and grows by the bit that
// brings the most still-uncovered groups within reach (groups that are nearly inside count more).
static int achunks_g0(const std::vector<ExChunkT> &a, int a0, const ExChunkT &cur) { return a0 < (int)a.size() ? a[a0].g0 : cur.g0; }
static int achunks_t0(const std::vector<ExChunkT> &a, int a0, const ExChunkT &cur) { return a0 < (int)a.size() ? a[a0].t0 : cur.t0; }

// the trip part of an entry's pair index, for tile_entry_pairs (sv_tile.hpp ExEntryT::pb / tsign): deposit of the trip counter's bits
// over the positions outside x, swizzled byte offsets, the terms' parities on the partner index
inline uint32_t host_deposit(uint32_t k, uint32_t fixmask) {
    for (uint32_t m = fixmask; m; m &= m - 1u) {
        const uint32_t low = (m & (0u - m)) - 1u;
        k = ((k & ~low) << 1) | (k & low);
    }
    return k;
}
inline uint32_t host_tile_swz(uint32_t e, bool real) { return real ? e ^ (((e >> 4) & 7u) << 1) : e ^ ((e >> 3) & 7u); }
void fill_entry_basis(ExEntryT &en, const std::vector<ExTermT> &tterms, bool real, int M) {
    const uint32_t nel = 1u << M, ab = real ? 8u : 16u;
    const uint32_t xf = en.pad ? (uint32_t)en.pad : en.x;
    const uint32_t iu0 = (host_deposit((uint32_t)en.k0, en.x) | en.ibits) & (nel - 1u);
    uint32_t d[4];
    for (int b = 0; b < 4; ++b) d[b] = host_deposit(64u << b, en.x) & (nel - 1u);
    en.tsign = 0;
    const int nt = std::min(2, en.t1 - en.t0);
    for (uint32_t t = 0; t < 16u; ++t) {
        uint32_t iu = iu0;                                  // trip t's share of the pair index: trip 0's XOR one basis value per set bit of t
        for (int b = 0; b < 4; ++b)
            if ((t >> b) & 1u) iu ^= d[b];
        en.ph[t] = host_tile_swz(iu, real) * ab;
        for (int j = 0; j < nt; ++j) {
            const uint32_t zin = tterms[(size_t)en.t0 + (j == 0 ? 0 : (en.t1 - en.t0 - 1))].zin;   // (the kernel takes the first and the LAST term)
            en.tsign |= ((uint32_t)__builtin_popcount((iu ^ xf) & zin) & 1u) << (16 * j + t);
        }
    }
}

int build_ham_tiles(ovqe_handle h, HamDev &H, bool real) {
    const int M = tile_bits(h, real), L = ham_tile_low(h, real);
    H.tile_bits = M;
    H.tile_low = L;
    H.tile_real = real;
    H.tsweeps.clear();
    H.tsweep_terms.clear();
    H.h_achunks.clear();
    H.cover_id++;
    H.n_rest = 0;
    H.tile_work = 0;
    H.diag_sweep = -1;
    H.diag_nu = 0;
    const bool tiled = tile_ok(h, real) && H.groups.size() >= 3;
    if (!tiled) return OVQE_OK;
    const int G = (int)H.groups.size();
    const uint64_t lowbits = (1ull << L) - 1ull;
    std::vector<char> covered(G, 0);
    std::vector<ExChunkT> chunks;
    std::vector<ExEntryT> tgroups;
    std::map<std::pair<uint64_t, uint64_t>, std::pair<double, double>> merged;
    std::vector<ExTermT> tterms;
    std::vector<ExFlatT> tflats;
    std::vector<ExItemT> titems;
    std::vector<ExChunkT> achunks;
    std::vector<ExAGroupT> agroups;
    std::vector<ExTermT> aterms;
    std::vector<HGroup> rest;
    int remaining = 0;
    for (int g = 0; g < G; ++g) {
        if (__builtin_popcountll(H.groups[g].x | lowbits) > M) {
            covered[g] = 2;
            rest.push_back(H.groups[g]);
        } else {
            ++remaining;
        }
    }
    const double wgt[8] = {1.0, 0.25, 0.0625, 0.015625, 0.00390625, 0.0009765625, 0.000244140625, 0.00006103515625};
    while (remaining > 0) {
        uint64_t S = lowbits;
        while (__builtin_popcountll(S) < M) {
            const int room = M - __builtin_popcountll(S);
            double score[64] = {0.0};
            bool any = false;
            for (int g = 0; g < G; ++g) {
                if (covered[g]) continue;
                const uint64_t miss = H.groups[g].x & ~S;
                const int nm = __builtin_popcountll(miss);
                if (nm == 0 || nm > room) continue;
                any = true;
                for (uint64_t mk = miss; mk; mk &= mk - 1ull) score[__builtin_ctzll(mk)] += wgt[std::min(nm - 1, 7)];
            }
            if (!any) break;
            int best = -1;
            for (int b = 0; b < h->n_local; ++b)
                if (!((S >> b) & 1ull) && (best < 0 || score[b] > score[best])) best = b;
            S |= 1ull << best;
        }
        for (int b = 0; __builtin_popcountll(S) < M; ++b) S |= 1ull << b;
        ExSweep sw = {};
        sw.smask = real ? S >> 1 : S;  // real state: masks in the index space of amplitude pairs (sv_tile.hpp)
        uint64_t lo = 0, mk = sw.smask;
        for (int k = 0; k < TILE_EXPECT_LOG_NT; ++k) {  // thread bits
            lo |= mk & (0ull - mk);
            mk &= mk - 1ull;
        }
        sw.mask_lo = lo;
        sw.mask_hi = sw.smask & ~lo;
        sw.c0 = (int32_t)chunks.size();
        sw.i0 = (int32_t)titems.size();
        ExChunkT ck = {(int32_t)tgroups.size(), (int32_t)tgroups.size(), (int32_t)tterms.size(), (int32_t)tterms.size()};
        int took = 0;
        sw.a0 = (int32_t)achunks.size();
        ExChunkT ak = {(int32_t)agroups.size(), (int32_t)agroups.size(), (int32_t)aterms.size(), (int32_t)aterms.size()};
        for (int g = 0; g < G; ++g) {
            if (covered[g] || (H.groups[g].x & ~S)) continue;
            covered[g] = 1;
            --remaining;
            ++took;
            const HGroup &gr = H.groups[g];
            if (gr.x == 0) H.diag_sweep = (int)H.tsweeps.size();   // (this sweep; pushed below)
            const uint32_t xl = extract_bits(gr.x, S);
            {   // operator-application form (k_tile_apply; sparse tiles of k_tile_expect): the group's raw terms, split
                // when they exceed a chunk.  Real state: strings with an imaginary folded coefficient (odd number of
                // Y) have <P> = 0 and are left out.
                std::vector<int> keep;
                for (int t = gr.t0; t < gr.t1; ++t)
                    if (!(real && H.terms[t].ci != 0.0)) keep.push_back(t);
                // a group enters in pieces of at most TILE_APPLY_TERMS terms (D_g is a sum over its terms, so the pieces
                // are independent work units): one wave serves a piece, and the diagonal group with its hundreds of
                // Z strings no longer keeps a single wave busy while the others idle
                for (size_t k0 = 0; k0 < keep.size(); k0 += TILE_APPLY_TERMS) {
                    const size_t k1 = std::min(keep.size(), k0 + TILE_APPLY_TERMS);
                    if ((int)aterms.size() - ak.t0 + (int)(k1 - k0) > TILE_TERM_CAP ||
                        (int)agroups.size() - ak.g0 + 1 > TILE_APPLY_GROUPS) {
                        ak.g1 = (int32_t)agroups.size();
                        ak.t1 = (int32_t)aterms.size();
                        achunks.push_back(ak);
                        ak = {ak.g1, ak.g1, ak.t1, ak.t1};
                    }
                    ExAGroupT ag = {xl, (int32_t)aterms.size(), 0, 0};
                    for (size_t k = k0; k < k1; ++k) {
                        const HTerm &ht = H.terms[keep[k]];
                        ExTermT et = {};
                        et.zin = extract_bits(ht.z, S);
                        et.zout = ht.z & ~S;
                        et.cr = ht.cr;
                        et.ci = ht.ci;
                        aterms.push_back(et);
                    }
                    ag.t1 = (int32_t)aterms.size();
                    ag.pad = 1;   // bit 0: every folded coefficient of the piece is real (a real-symmetric H: all of them)
                    for (int32_t t = ag.t0; t < ag.t1; ++t)
                        if (aterms[t].ci != 0.0) ag.pad = 0;
                    agroups.push_back(ag);
                }
            }
            const int w = __builtin_popcount(xl);
            int xpos[16], np = 0;
            for (uint32_t mk2 = xl; mk2; mk2 &= mk2 - 1u) xpos[np++] = __builtin_ctz(mk2);
            // UNSPLIT entries (round 5): a group of one or two raw terms on a complex state — no pattern of its x bits cancels, so
            // cutting it into 2^(w-1) pattern entries only multiplies the per-entry set-up (a quarter of the kernel's instructions at
            // w = 4: 25 per pair).  Its 2^(M-1) pairs (pivot bit of i clear) are walked in pieces of 512 with the x part of z left in
            // the terms' masks; en.x = the pivot bit alone (what the index walk skips), en.pad = the mask that leads to the partner.
            if (!real && w >= 2 && gr.t1 - gr.t0 <= 2 && h->opt_tile_unsplit && M - 1 >= 9) {
                if ((int)tterms.size() - ck.t0 + (gr.t1 - gr.t0) > TILE_TERM_CAP) {
                    ck.g1 = (int32_t)tgroups.size();
                    ck.t1 = (int32_t)tterms.size();
                    chunks.push_back(ck);
                    ck = {ck.g1, ck.g1, ck.t1, ck.t1};
                }
                const int32_t t0 = (int32_t)tterms.size();
                bool real_only = true;
                for (int t = gr.t0; t < gr.t1; ++t) {
                    const HTerm &ht = H.terms[t];
                    ExTermT et = {};
                    et.zin = extract_bits(ht.z, S);
                    et.zout = ht.z & ~S;
                    et.cr = ht.cr;
                    et.ci = ht.ci;
                    if (et.ci != 0.0) real_only = false;
                    tterms.push_back(et);
                }
                const int npairs = 1 << (M - 1), piece = std::min(npairs, TILE_UNSPLIT_PAIRS);
                for (int k0 = 0; k0 < npairs; k0 += piece) {
                    ExEntryT en = {};
                    en.x = 1u << xpos[w - 1];
                    en.ibits = 0;
                    en.t0 = t0;
                    en.t1 = (int32_t)tterms.size();
                    en.k0 = k0;
                    en.nk = piece;
                    en.real_only = real_only ? 1 : 0;
                    en.pad = (int32_t)xl;
                    fill_entry_basis(en, tterms, real, M);
                    tgroups.push_back(en);
                }
                continue;
            }
            const uint32_t npat = w ? (1u << (w - 1)) : 1u;
            for (uint32_t e = 0; e < npat; ++e) {
                uint32_t ibits = 0;
                for (int f = 0; f + 1 < w; ++f)
                    if ((e >> f) & 1u) ibits |= 1u << xpos[f];
                const uint32_t jx = ibits ^ xl;  // the partner's bits on the x positions
                // merge the terms that agree outside x: key = (z on the tile without x, z outside the tile)
                merged.clear();
                for (int t = gr.t0; t < gr.t1; ++t) {
                    const HTerm &ht = H.terms[t];
                    if (real && ht.ci != 0.0) continue;  // imaginary folded coefficient: <P> = 0 on a real state
                    const uint32_t zin = extract_bits(ht.z, S);
                    const double sg = (__builtin_popcount(jx & zin) & 1) ? -1.0 : 1.0;
                    auto &slot = merged[std::make_pair((uint64_t)(zin & ~xl), ht.z & ~S)];
                    slot.first += sg * ht.cr;
                    slot.second += sg * ht.ci;
                }
                std::vector<ExTermT> mt;
                bool real_only = true;
                for (const auto &kv : merged) {
                    if (kv.second.first == 0.0 && kv.second.second == 0.0) continue;  // exact cancellation
                    ExTermT et = {};
                    et.zin = (uint32_t)kv.first.first;
                    et.zout = kv.first.second;
                    et.cr = kv.second.first;
                    et.ci = kv.second.second;
                    if (et.ci != 0.0) real_only = false;
                    mt.push_back(et);
                }
                if (mt.empty()) continue;
                const int nk_total = 1 << (M - w);
                // (round 6: the float64 shards of the partitioned register are DENSE real states of 25+ qubits: per-wave entries there too —
                // 31 qubits, 1000-term <H>: 1.09 s with items, 0.94 s with entries, tools/exp_real_shard.py)
                if (mt.size() <= 2 && nk_total >= 2 && (h->opt_tile_flat == 1 || (h->opt_tile_flat == 2 && real && h->n_local < 25))) {  // one lane per TILE_ITEM_PAIRS pairs (sv_tile.hpp)
                    ExFlatT fe = {};
                    fe.x = xl;
                    fe.ibits = ibits;
                    fe.zin0 = mt[0].zin;
                    fe.zout0 = mt[0].zout;
                    fe.c0r = mt[0].cr;
                    fe.c0i = mt[0].ci;
                    if (mt.size() == 2) {
                        fe.zin1 = mt[1].zin;
                        fe.zout1 = mt[1].zout;
                        fe.c1r = mt[1].cr;
                        fe.c1i = mt[1].ci;
                    }
                    for (int k0 = 0; k0 < nk_total; k0 += TILE_ITEM_PAIRS) {
                        uint32_t istart = (uint32_t)k0;  // deposit k0 over the positions outside x, ascending
                        for (int f = 0; f < w; ++f) {
                            const uint32_t low = (1u << xpos[f]) - 1u;
                            istart = ((istart & ~low) << 1) | (istart & low);
                        }
                        titems.push_back(ExItemT{(uint32_t)tflats.size(), istart | ibits,
                                                 (uint32_t)std::min(TILE_ITEM_PAIRS, nk_total - k0), 0u});
                    }
                    tflats.push_back(fe);
                    H.tile_work += (int64_t)nk_total * (int64_t)mt.size();
                    continue;
                }
                for (size_t m0 = 0; m0 < mt.size(); m0 += TILE_TERM_CAP) {  // oversized lists are split (linear)
                    const size_t m1 = std::min(mt.size(), m0 + TILE_TERM_CAP);
                    if ((int)tterms.size() - ck.t0 + (int)(m1 - m0) > TILE_TERM_CAP) {
                        ck.g1 = (int32_t)tgroups.size();
                        ck.t1 = (int32_t)tterms.size();
                        chunks.push_back(ck);
                        ck = {ck.g1, ck.g1, ck.t1, ck.t1};
                    }
                    const int32_t t0 = (int32_t)tterms.size();
                    tterms.insert(tterms.end(), mt.begin() + m0, mt.begin() + m1);
                    for (int k0 = 0; k0 < nk_total; k0 += TILE_ENTRY_PAIRS) {
                        ExEntryT en = {};
                        en.x = xl;
                        en.ibits = ibits;
                        en.t0 = t0;
                        en.t1 = (int32_t)tterms.size();
                        en.k0 = k0;
                        en.nk = std::min(TILE_ENTRY_PAIRS, nk_total - k0);
                        en.real_only = real_only ? 1 : 0;
                        if (en.t1 - en.t0 <= 2) fill_entry_basis(en, tterms, real, M);
                        tgroups.push_back(en);
                    }
                }
            }
        }
        ck.g1 = (int32_t)tgroups.size();
        ck.t1 = (int32_t)tterms.size();
        if (ck.g1 > ck.g0) chunks.push_back(ck);
        sw.c1 = (int32_t)chunks.size();
        sw.i1 = (int32_t)titems.size();
        ak.g1 = (int32_t)agroups.size();
        ak.t1 = (int32_t)aterms.size();
        if (ak.g1 > ak.g0) achunks.push_back(ak);
        sw.a1 = (int32_t)achunks.size();
        if (took == 0) return fail(h, OVQE_ERR_INVALID, "internal: tile cover made no progress");
        if (std::getenv("OVQE_DEBUG_COVER"))
            std::fprintf(stderr, "cover sweep %zu: S=%llx groups=%d pieces=%d apply_terms=%d entries=%d items=%d\n",
                         H.tsweeps.size(), (unsigned long long)S, took, (int)agroups.size() - achunks_g0(achunks, sw.a0, ak),
                         (int)aterms.size() - achunks_t0(achunks, sw.a0, ak), (int)tgroups.size() - (int)(sw.c0 < (int)chunks.size() ? chunks[sw.c0].g0 : ck.g0),
                         (int)titems.size() - sw.i0);
        H.tsweeps.push_back(sw);
        H.tsweep_terms.push_back((int)aterms.size() - achunks_t0(achunks, sw.a0, ak));
    }
    H.n_rest = (int)rest.size();
    // the diagonal group in Walsh-Hadamard form (k_tile_diag): contiguous tiles of 2^12 amplitudes, the terms grouped by their z mask
    // on the tile bits (ascending mask, the group's term order inside: a fixed summation order)
    if (H.diag_sweep >= 0 && h->n_local >= 25 && h->opt_expect_diag_wht) {
        const HGroup *dg = nullptr;
        for (const HGroup &g : H.groups)
            if (g.x == 0) dg = &g;
        constexpr int DM = 12;
        if (dg && dg->t1 - dg->t0 >= 12) {
            std::map<uint32_t, std::vector<DiagTermT>> by_zin;
            for (int t = dg->t0; t < dg->t1; ++t)
                by_zin[(uint32_t)(H.terms[t].z & ((1ull << DM) - 1ull))].push_back(DiagTermT{H.terms[t].z >> DM << DM, H.terms[t].cr});
            std::vector<uint32_t> uz;
            std::vector<int32_t> uo = {0};
            std::vector<DiagTermT> dt;
            for (const auto &kv : by_zin) {
                uz.push_back(kv.first);
                dt.insert(dt.end(), kv.second.begin(), kv.second.end());
                uo.push_back((int32_t)dt.size());
            }
            int rc = upload(h, H.d_dzin, uz.data(), uz.size() * sizeof(uint32_t));
            if (!rc) rc = upload(h, H.d_doff, uo.data(), uo.size() * sizeof(int32_t));
            if (!rc) rc = upload(h, H.d_dterms, dt.data(), dt.size() * sizeof(DiagTermT));
            if (rc) return rc;
            H.diag_nu = (int)uz.size();
            H.diag_bits = DM;
        }
    }
    for (const ExEntryT &en : tgroups) H.tile_work += (int64_t)en.nk * (en.t1 - en.t0);
    H.tile_entries = (int64_t)(tgroups.size() + tflats.size());
    H.tile_terms = (int64_t)tterms.size();
    int rc = upload(h, H.d_tchunks, chunks.data(), chunks.size() * sizeof(ExChunkT));
    if (rc) return rc;
    rc = upload(h, H.d_tgroups, tgroups.data(), tgroups.size() * sizeof(ExEntryT));
    if (rc) return rc;
    rc = upload(h, H.d_tterms, tterms.data(), tterms.size() * sizeof(ExTermT));
    if (rc) return rc;
    rc = upload(h, H.d_tflats, tflats.data(), tflats.size() * sizeof(ExFlatT));
    if (rc) return rc;
    rc = upload(h, H.d_titems, titems.data(), titems.size() * sizeof(ExItemT));
    if (rc) return rc;
    H.h_achunks = achunks;
    rc = upload(h, H.d_achunks, achunks.data(), achunks.size() * sizeof(ExChunkT));
    if (!rc) rc = upload(h, H.d_agroups, agroups.data(), agroups.size() * sizeof(ExAGroupT));
    if (!rc) rc = upload(h, H.d_aterms, aterms.data(), aterms.size() * sizeof(ExTermT));
    if (rc) return rc;
    return upload(h, H.d_rest, rest.data(), rest.size() * sizeof(HGroup));
}

inline int expect_ysplit(ovqe_handle h, int M) {  // workgroups per tile: fill the chip when there are few tiles
    const uint64_t tiles = h->namps >> M;
    int y = 1;
    while (y < 8 && tiles * y < 1024) y *= 2;
    return y;
}

// dense_only: the caller knows that no tile of this state is sparse (census of the evaluation's first sweep, run_expectation_tiled):
// the staging area of the sparse-tile path and the tile's non-zero list (38 KB beside a 64-KB tile: ONE workgroup per CU) are left
// out of the launch, and two workgroups share a CU — one loads its tile while the other computes
template <int M, bool REAL>
int launch_tile_expect(ovqe_handle h, const HamDev &H, const ExSweep &sw, double2 *partials, int accumulate,
                       hipStream_t stream, bool dense_only = false, int *census = nullptr, bool skip_diag = false) {
    constexpr int NT = 1 << TILE_EXPECT_LOG_NT;
    static_assert(TILE_SPARSE_TERMS >= 2 * TILE_TERM_CAP && TILE_SPARSE_GROUPS >= 2 * TILE_APPLY_GROUPS, "two host chunks per pass");
    const size_t smem_full = ((size_t)(REAL ? 8 : 16) << M) + TILE_SPARSE_TERMS * sizeof(ExTermLds) +
                             TILE_SPARSE_GROUPS * sizeof(ExAGroupT) + (NT / 64) * sizeof(double2) + (NT / 64 + 2) * sizeof(int) +
                             ((size_t)2 << M);
    // (dense path: the term table of a chunk, TILE_TERM_CAP entries, lives at the start of the staging bytes; the reduction slots
    // sit behind the whole staging area in the kernel's layout, so the dense launch keeps the area's address range up to them)
    const size_t smem_dense = ((size_t)(REAL ? 8 : 16) << M) + TILE_TERM_CAP * sizeof(ExTermLds) + (NT / 64) * sizeof(double2) +
                              (NT / 64 + 2) * sizeof(int);
    const size_t smem = dense_only ? smem_dense : smem_full;
    const dim3 grid((unsigned)(h->namps >> M), (unsigned)expect_ysplit(h, M));
    const int sparse_den = dense_only ? (skip_diag ? -2 : -1) : ((H.d_agroups.p && sw.a1 > sw.a0) ? h->opt_expect_sparse : 0);
    static bool attr_done_dev[64] = {};  // function attributes are per device
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_expect<M, NT, true, REAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_full));
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_expect<M, NT, false, REAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_full));
        attr_done = true;
    }
    if (h->n_local >= 25) {
        hipLaunchKernelGGL((k_tile_expect<M, NT, true, REAL>), grid, dim3(NT), smem, stream, (const void *)h->state, h->base, sw,
                           (const ExChunkT *)H.d_tchunks.p, (const ExEntryT *)H.d_tgroups.p,
                           (const ExTermT *)H.d_tterms.p, (const ExFlatT *)H.d_tflats.p, (const ExItemT *)H.d_titems.p, partials,
                           accumulate, (const ExChunkT *)H.d_achunks.p, (const ExAGroupT *)H.d_agroups.p,
                           (const ExTermT *)H.d_aterms.p, sparse_den, census);
    } else {
        hipLaunchKernelGGL((k_tile_expect<M, NT, false, REAL>), grid, dim3(NT), smem, stream, (const void *)h->state, h->base, sw,
                           (const ExChunkT *)H.d_tchunks.p, (const ExEntryT *)H.d_tgroups.p,
                           (const ExTermT *)H.d_tterms.p, (const ExFlatT *)H.d_tflats.p, (const ExItemT *)H.d_titems.p, partials,
                           accumulate, (const ExChunkT *)H.d_achunks.p, (const ExAGroupT *)H.d_agroups.p,
                           (const ExTermT *)H.d_aterms.p, sparse_den, census);
    }
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// <state|H|state> of the stored Hamiltonian through the tile cover; *used = false when there is no cover
int run_expectation_tiled(ovqe_handle h, HamDev &H, double2 *out, bool *used, bool real = false) {
    *used = false;
    if (H.tile_bits != tile_bits(h, real) || H.tile_low != ham_tile_low(h, real) || H.tile_real != real) {
        int rc = build_ham_tiles(h, H, real);
        if (rc) return rc;
    }
    if (H.tsweeps.empty() && !(real && H.n_rest)) return OVQE_OK;
    const int M = H.tile_bits;
    // partial sums: one per workgroup of the tile sweeps (none when every group keeps its own sweep)
    const int64_t ntiles = H.tsweeps.empty() ? 0 : (int64_t)(h->namps >> M) * expect_ysplit(h, M);
    const int nb = reduce_blocks(h->namps);
    const int64_t ndiag = H.diag_nu > 0 ? (int64_t)(h->namps >> H.diag_bits) : 0;   // workgroups of k_tile_diag (dense registers)
    int rc = ensure(h, h->d_partials, (size_t)(2 * ntiles + nb + ndiag) * sizeof(double2));
    if (rc) return rc;
    rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    double2 *partials = (double2 *)h->d_partials.p;
    // two streams when the cover has enough sweeps: the sweeps are taken heaviest-first by the main stream and
    // lightest-first by the second one until they meet (estimated duration: one read of the state + compute per term)
    const int ns = (int)H.tsweeps.size();
    // (shards too: the second stream forks from and joins the handle's stream by events, so what the caller ordered behind that
    // stream — the RCCL transfers of the partitioned register — stays ordered behind both)
    const bool dual = h->opt_expect_streams >= 2 && ns >= 8 && h->n_local >= 20;
    if (dual && !h->stream2) {
        HIPC(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        HIPC(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPC(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    std::vector<char> on_second(ns, 0);
    int acc[2] = {0, 0};
    // Dense complex registers of 25+ qubits (a shard of the partitioned register, a random state): the LDS that the sparse-tile path
    // needs beside a 64-KB tile leaves ONE workgroup per CU, which then loads, waits and computes in turn (31 qubits: 62 sweeps at
    // 1.1 TB/s).  The first sweep of an evaluation counts the tiles that took the sparse path; when none did, the state is dense
    // under every tile bit set and the remaining sweeps run without that area: two workgroups per CU, of either stream.  (One stream
    // synchronisation per evaluation: only where there are at least four sweeps of at least half a gigabyte.)
    // (round 6: real states too — the float64 shards of the partitioned register: tiles of 2^13 doubles)
    const bool try_dense = ((!real && M == 12) || (real && M == 13)) && h->n_local >= 25 && ns >= 4 && h->opt_expect_sparse > 0 && h->opt_expect_dense;
    bool dense_only = false;
    int census_k = -1;      // the sweep that ran first, as the census
    if (try_dense) {
        rc = ensure(h, h->d_tile_cnt, sizeof(int));
        if (rc) return rc;
        // (the census takes a sweep WITHOUT the diagonal group when that group has a Walsh-Hadamard form: on a dense register the
        // group's hundreds of strings then never run term by term)
        census_k = (ndiag && H.diag_sweep == 0 && ns > 1) ? 1 : 0;
        HIPC(h, hipMemsetAsync(h->d_tile_cnt.p, 0, sizeof(int), h->stream));
        rc = real ? launch_tile_expect<13, true>(h, H, H.tsweeps[census_k], partials, 0, h->stream, false, (int *)h->d_tile_cnt.p)
                  : launch_tile_expect<12, false>(h, H, H.tsweeps[census_k], partials, 0, h->stream, false, (int *)h->d_tile_cnt.p);
        if (rc) return rc;
        int census = 1;
        HIPC(h, hipMemcpyAsync(&census, h->d_tile_cnt.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        dense_only = census == 0;
        acc[0] = 1;
    }
    const bool diag_wht = dense_only && ndiag > 0 && H.diag_sweep != census_k;
    if (dual) {
        const double mem = 16.0 * (double)h->namps * (real ? 0.5 : 1.0) / 2.8e6;        // us at ~2.8 TB/s
        const double per_term = 0.23 * (double)h->namps / (double)(1ull << 24);       // us, measured at 24 qubits
        double ta = 0.0, tb = 0.0;
        int i = 0, j = ns - 1;
        while (i <= j) {
            if (ta <= tb) ta += mem + per_term * H.tsweep_terms[i++];
            else {
                tb += mem + per_term * H.tsweep_terms[j];
                on_second[j--] = 1;
            }
        }
        HIPC(h, hipEventRecord(h->ev_fork, h->stream));
        HIPC(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
    }
    for (int k = 0; k < ns; ++k) {
        if (k == census_k) continue;
        const ExSweep &sw = H.tsweeps[k];
        const int which = on_second[k];
        hipStream_t strm = which ? h->stream2 : h->stream;
        double2 *part = partials + (which ? ntiles : 0);
        if (real) {
            switch (M) {
            case 11: rc = launch_tile_expect<11, true>(h, H, sw, part, acc[which], strm); break;
            case 12: rc = launch_tile_expect<12, true>(h, H, sw, part, acc[which], strm); break;
            default: rc = launch_tile_expect<13, true>(h, H, sw, part, acc[which], strm, dense_only, nullptr, diag_wht); break;
            }
        } else {
            switch (M) {
            case 10: rc = launch_tile_expect<10, false>(h, H, sw, part, acc[which], strm); break;
            case 11: rc = launch_tile_expect<11, false>(h, H, sw, part, acc[which], strm); break;
            default: rc = launch_tile_expect<12, false>(h, H, sw, part, acc[which], strm, dense_only, nullptr, diag_wht); break;
            }
        }
        if (rc) return rc;
        acc[which] = 1;
    }
    const int nparts = (dual && acc[1]) ? 2 : 1;
    if (dual) {
        HIPC(h, hipEventRecord(h->ev_join, h->stream2));
        HIPC(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
    int64_t count = ntiles * nparts;
    if (diag_wht) {   // the diagonal group: one pass over contiguous tiles, Walsh-Hadamard form (sv_tile.hpp k_tile_diag)
        constexpr int DM = 12, DNT = 512;
        const size_t dsm = ((size_t)8 << DM) + (DNT / 64) * sizeof(double2);
        const bool ntl = h->n_local >= 25;
        if (real) {
            if (ntl) hipLaunchKernelGGL((k_tile_diag<DM, DNT, true, true>), dim3((unsigned)ndiag), dim3(DNT), dsm, h->stream, (const void *)h->state, h->base,
                                        (const uint32_t *)H.d_dzin.p, (const int32_t *)H.d_doff.p, (const DiagTermT *)H.d_dterms.p, H.diag_nu, partials + count, 0);
            else hipLaunchKernelGGL((k_tile_diag<DM, DNT, false, true>), dim3((unsigned)ndiag), dim3(DNT), dsm, h->stream, (const void *)h->state, h->base,
                                    (const uint32_t *)H.d_dzin.p, (const int32_t *)H.d_doff.p, (const DiagTermT *)H.d_dterms.p, H.diag_nu, partials + count, 0);
        } else {
            if (ntl) hipLaunchKernelGGL((k_tile_diag<DM, DNT, true, false>), dim3((unsigned)ndiag), dim3(DNT), dsm, h->stream, (const void *)h->state, h->base,
                                        (const uint32_t *)H.d_dzin.p, (const int32_t *)H.d_doff.p, (const DiagTermT *)H.d_dterms.p, H.diag_nu, partials + count, 0);
            else hipLaunchKernelGGL((k_tile_diag<DM, DNT, false, false>), dim3((unsigned)ndiag), dim3(DNT), dsm, h->stream, (const void *)h->state, h->base,
                                    (const uint32_t *)H.d_dzin.p, (const int32_t *)H.d_doff.p, (const DiagTermT *)H.d_dterms.p, H.diag_nu, partials + count, 0);
        }
        count += ndiag;
    }
    if (H.n_rest) {
        if (real)
            hipLaunchKernelGGL(k_expect_pairs_real, dim3(nb), dim3(256), 0, h->stream, (const double *)h->state, h->namps,
                               (const HGroup *)H.d_rest.p, 0, H.n_rest, (const HTerm *)H.d_terms.p, partials + count);
        else
            hipLaunchKernelGGL(k_expect_pairs, dim3(nb), dim3(256), 0, h->stream, h->state, h->namps,
                               (const HGroup *)H.d_rest.p, 0, H.n_rest, (const HTerm *)H.d_terms.p, partials + count);
        count += nb;
    }
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)partials, count,
                       (double2 *)h->d_result.p, 0);
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    *out = h->h_result[0];
    *used = true;
    h->last_passes = ns + (H.n_rest ? 1 : 0);
    h->last_pass_bytes = (int64_t)((real ? 8.0 : 16.0) * (double)h->namps * (double)(ns + H.n_rest));
    return OVQE_OK;
}

// ---- compact cover -----------------------------------------------------------------------------------------------
int run_program_streaming(ovqe_handle h, const double *theta, bool real);

// Support of the program's states (non-zeros of the real state prepared at a generic parameter vector: structural zeros
// are exact zeros for every theta, and a structurally non-zero amplitude vanishes exactly only on a null set) and, for
// every sweep of the real cover, that support sorted by tile.  Clobbers the state buffer.
int build_compact_cover(ovqe_handle h, HamDev &H) {
    CompactCover &C = h->cc;
    C.valid = false;
    C.disabled = true;   // until everything below succeeded
    C.prog_version = h->prog_version;
    C.ham_version = H.version;
    C.cover_id = H.cover_id;
    const int M = H.tile_bits;
    if (H.tsweeps.empty() || H.n_rest || !H.tile_real || h->n_local > 28 || M > 13) return OVQE_OK;
    std::vector<double> theta((size_t)std::max(h->K, 1));
    for (int k = 0; k < h->K; ++k) {
        const double f = 0.6180339887498949 * (k + 1);
        theta[k] = 0.4 + 0.7 * (f - std::floor(f));
    }
    int rc = run_program_streaming(h, theta.data(), true);
    if (rc) return rc;
    std::vector<double> host(h->namps);
    HIPC(h, hipMemcpyAsync(host.data(), h->state, h->namps * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    std::vector<uint32_t> sup;
    for (uint64_t i = 0; i < h->namps; ++i)
        if (host[i] != 0.0) sup.push_back((uint32_t)i);
    std::vector<double>().swap(host);
    const uint32_t K = (uint32_t)sup.size();
    if (K == 0 || (uint64_t)K * 8ull > h->namps) return OVQE_OK;   // not sparse enough to pay off
    const int ns = (int)H.tsweeps.size();
    const uint64_t ntiles = h->namps >> M;
    std::vector<uint16_t> loc((size_t)ns * K);
    std::vector<uint32_t> cid((size_t)ns * K), off((size_t)ns * (ntiles + 1));
    std::vector<uint32_t> maxn(ns, 0);
    const uint64_t allbits = h->namps - 1ull;
    auto work = [&](int s0, int s1) {
        std::vector<uint32_t> cnt(ntiles + 1), tile_of(K);
        for (int s = s0; s < s1; ++s) {
            const uint64_t samp = (H.tsweeps[s].smask << 1) | 1ull;   // the tile's bits in amplitude-index space
            const uint64_t outside = allbits & ~samp;
            // byte tables of the two bit extractions
            uint32_t tl[4][256], tt[4][256];
            for (int b = 0; b < 4; ++b) {
                const uint64_t ms = (samp >> (8 * b)) & 0xffull, mo = (outside >> (8 * b)) & 0xffull;
                const int sh_s = __builtin_popcountll(samp & ((1ull << (8 * b)) - 1ull));
                const int sh_o = __builtin_popcountll(outside & ((1ull << (8 * b)) - 1ull));
                for (int v = 0; v < 256; ++v) {
                    tl[b][v] = extract_bits((uint64_t)v, ms) << sh_s;
                    tt[b][v] = extract_bits((uint64_t)v, mo) << sh_o;
                }
            }
            std::fill(cnt.begin(), cnt.end(), 0u);
            for (uint32_t k = 0; k < K; ++k) {
                const uint32_t i = sup[k];
                const uint32_t t = tt[0][i & 255u] | tt[1][(i >> 8) & 255u] | tt[2][(i >> 16) & 255u] | tt[3][i >> 24];
                tile_of[k] = t;
                cnt[t + 1]++;
            }
            uint32_t mx = 0;
            for (uint64_t t = 0; t < ntiles; ++t) {
                mx = std::max(mx, cnt[t + 1]);
                cnt[t + 1] += cnt[t];
            }
            maxn[s] = mx;
            uint32_t *o = off.data() + (size_t)s * (ntiles + 1);
            std::copy(cnt.begin(), cnt.end(), o);
            uint16_t *L = loc.data() + (size_t)s * K;
            uint32_t *Cd = cid.data() + (size_t)s * K;
            for (uint32_t k = 0; k < K; ++k) {   // ascending k inside a tile: a stable order
                const uint32_t i = sup[k];
                const uint32_t l = tl[0][i & 255u] | tl[1][(i >> 8) & 255u] | tl[2][(i >> 16) & 255u] | tl[3][i >> 24];
                const uint32_t pos = cnt[tile_of[k]]++;
                L[pos] = (uint16_t)l;
                Cd[pos] = k;   // compact id of the element at this tile-sorted position
            }
        }
    };
    {
        const int nthr = std::max(1, std::min<int>({ns, 16, (int)std::thread::hardware_concurrency()}));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t) pool.emplace_back(work, (int)((int64_t)ns * t / nthr), (int)((int64_t)ns * (t + 1) / nthr));
        for (std::thread &t : pool) t.join();
    }
    rc = upload(h, C.d_sup, sup.data(), (size_t)K * sizeof(uint32_t));
    if (!rc) rc = upload(h, C.d_loc, loc.data(), loc.size() * sizeof(uint16_t));
    if (!rc) rc = upload(h, C.d_cid, cid.data(), cid.size() * sizeof(uint32_t));
    if (!rc) rc = upload(h, C.d_off, off.data(), off.size() * sizeof(uint32_t));
    if (!rc) rc = ensure(h, C.d_psic, ((size_t)ns + 1) * K * sizeof(double));   // [compact state][per-sweep tile order]
    if (!rc) rc = upload(h, C.d_sweeps, H.tsweeps.data(), (size_t)ns * sizeof(ExSweep));
    if (rc) return rc;
    C.K = K;
    C.ntiles = ntiles;
    C.max_nnz = *std::max_element(maxn.begin(), maxn.end());
    C.valid = true;
    C.disabled = false;
    return OVQE_OK;
}

template <int M>
int launch_tile_expect_compact(ovqe_handle h, const HamDev &H, double2 *partials) {
    constexpr int NT = 1 << TILE_EXPECT_LOG_NT;
    const CompactCover &C = h->cc;
    const int cpp = h->opt_compact_cpp, term_cap = cpp * TILE_TERM_CAP, group_cap = cpp * TILE_APPLY_GROUPS;
    const size_t smem = ((size_t)8 << M) + (size_t)term_cap * sizeof(ExTermLds) + (size_t)group_cap * sizeof(ExAGroupT) +
                        (NT / 64) * sizeof(double2) + (((size_t)C.max_nnz * 2 + 15) & ~(size_t)15);
    static bool attr_done_dev[64] = {};
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_expect_compact<M, NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_tile_expect_compact<M, NT, true>), dim3((unsigned)C.ntiles, (unsigned)H.tsweeps.size()), dim3(NT),
                       smem, h->stream, (const double *)C.d_psic.p + C.K, (const uint16_t *)C.d_loc.p, (const uint32_t *)C.d_off.p,
                       h->base, (const ExSweep *)C.d_sweeps.p, C.K, (const ExChunkT *)H.d_achunks.p,
                       (const ExAGroupT *)H.d_agroups.p, (const ExTermT *)H.d_aterms.p, partials, term_cap, group_cap, cpp);
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// <state|H|state> of a real state through the compact cover; *ok = false when the guard failed (support not closed)
int run_expectation_compact(ovqe_handle h, HamDev &H, double2 *out, bool *ok, bool psic_ready = false) {
    CompactCover &C = h->cc;
    *ok = false;
    const int ns = (int)H.tsweeps.size();
    const uint64_t total = (uint64_t)ns * C.K;
    const int nbg = (int)std::min<uint32_t>(2048u, (C.K + 255u) / 256u);
    const size_t nslots = (size_t)ns * C.ntiles;
    int rc = ensure(h, h->d_partials, (nslots + (size_t)ns + nbg) * sizeof(double2));
    if (!rc) rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    double2 *partials = (double2 *)h->d_partials.p, *rows = partials + nslots, *pnorm = rows + ns;
    if (!psic_ready)   // (else the sector path's circuit left the compact state in C.d_psic: nothing to gather, norm 1 by construction)
        hipLaunchKernelGGL(k_compact_gather<true>, dim3(nbg), dim3(256), 0, h->stream, (const double *)h->state,
                           (const uint32_t *)C.d_sup.p, C.K, (double *)C.d_psic.p, pnorm);
    hipLaunchKernelGGL(k_compact_permute<true>, dim3((unsigned)std::min<uint64_t>(16384u, (total + 255u) / 256u)), dim3(256), 0,
                       h->stream, (const double *)C.d_psic.p, (const uint32_t *)C.d_cid.p, total, (double *)C.d_psic.p + C.K);
    switch (H.tile_bits) {
    case 11: rc = launch_tile_expect_compact<11>(h, H, partials); break;
    case 12: rc = launch_tile_expect_compact<12>(h, H, partials); break;
    default: rc = launch_tile_expect_compact<13>(h, H, partials); break;
    }
    if (rc) return rc;
    // fixed-order reduction: per sweep over its tiles, then over the sweeps
    hipLaunchKernelGGL(k_reduce_rows2, dim3((unsigned)ns), dim3(256), 0, h->stream, (const double2 *)partials, (int)C.ntiles, rows);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)rows, (int64_t)ns, (double2 *)h->d_result.p, 0);
    if (!psic_ready)
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)pnorm, (int64_t)nbg,
                           (double2 *)h->d_result.p, 1);
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, 2 * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    // guard: the circuit is unitary, so the amplitudes on the support must carry the whole norm
    if (!psic_ready && std::fabs(h->h_result[1].x - 1.0) > 1e-9) return OVQE_OK;
    *out = h->h_result[0];
    *ok = true;
    return OVQE_OK;
}

// out = ident * in + H in for the stored Hamiltonian: tile sweeps when the cover holds every group, else the gather
// kernel.  in / out: complex states of this handle's size, out != in.
template <int M>
int launch_tile_apply(ovqe_handle h, const HamDev &H, const ExSweep &sw, const amp_t *in, amp_t *out, int first,
                      double ident, const uint32_t *tile_list = nullptr, const uint32_t *tile_count = nullptr,
                      unsigned listed_grid = 0) {
    constexpr int NT = 1 << TILE_EXPECT_LOG_NT;  // the sweeps' thread / trip masks are laid out for this group size
    const size_t smem = ((size_t)16 << M) + TILE_TERM_CAP * sizeof(ExTermLds) + TILE_APPLY_GROUPS * sizeof(ExAGroupT);
    static bool attr_done_dev[64] = {};  // function attributes are per device
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_apply<M, NT, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_apply<M, NT, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    const unsigned grid = tile_list ? listed_grid : (unsigned)(h->namps >> M);
    if (h->n_local >= 25) {
        hipLaunchKernelGGL((k_tile_apply<M, NT, true>), dim3(grid), dim3(NT), smem, h->stream, in, out, h->base, sw,
                           (const ExChunkT *)H.d_achunks.p, (const ExAGroupT *)H.d_agroups.p,
                           (const ExTermT *)H.d_aterms.p, first, ident, tile_list, tile_count);
    } else {
        hipLaunchKernelGGL((k_tile_apply<M, NT, false>), dim3(grid), dim3(NT), smem, h->stream, in, out, h->base, sw,
                           (const ExChunkT *)H.d_achunks.p, (const ExAGroupT *)H.d_agroups.p,
                           (const ExTermT *)H.d_aterms.p, first, ident, tile_list, tile_count);
    }
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// in_idx / in_count: the ascending list of the non-zero amplitudes of `in` when the caller has it (the ADAPT screen of a
// state of a few determinants): every sweep then only visits the tiles that hold one of them
int apply_hamiltonian(ovqe_handle h, amp_t *out, const amp_t *in, double ident, const uint64_t *in_idx = nullptr,
                      uint64_t in_count = 0, HamDev *which = nullptr) {
    HamDev &H = which ? *which : h->ham;   // (which: the local part of a planned cross-shard sum, cross_host.inc)
    if (H.tile_bits != tile_bits(h, false) || H.tile_low != ham_tile_low(h, false) || H.tile_real) {
        int rc = build_ham_tiles(h, H, false);
        if (rc) return rc;
    }
    // below ~2^8 tiles the tile sweeps leave most of the chip idle: the gather kernel (state in L2) is faster there
    if (H.tsweeps.empty() || H.n_rest || (int64_t)(h->namps >> H.tile_bits) < h->opt_apply_min_tiles) {
        hipLaunchKernelGGL(k_apply_sum, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, out, in, (amp_t *)nullptr,
                           h->namps, h->base, (const HGroup *)H.d_groups.p, (int)H.groups.size(),
                           (const HTerm *)H.d_terms.p, 1.0, 0.0, ident, 0.0);
        HIPC(h, hipGetLastError());
        return OVQE_OK;
    }
    const uint32_t *lists = nullptr, *counts = nullptr;
    const uint64_t ntiles = h->namps >> H.tile_bits;
    uint32_t cap = 0;
    if (in_idx && in_count > 0 && ntiles <= (1u << 19) && in_count * 4 <= ntiles) {
        const size_t nsw = H.tsweeps.size();
        cap = (uint32_t)std::min<uint64_t>(in_count, ntiles);
        std::vector<uint64_t> smasks(nsw);
        for (size_t k = 0; k < nsw; ++k) smasks[k] = H.tsweeps[k].smask;
        int rc = upload(h, h->d_tile_smasks, smasks.data(), nsw * sizeof(uint64_t));
        if (!rc) rc = ensure(h, h->d_tile_lists, nsw * (size_t)cap * sizeof(uint32_t));
        if (!rc) rc = ensure(h, h->d_tile_counts, nsw * sizeof(uint32_t));
        if (rc) return rc;
        HIPC(h, hipMemsetAsync(out, 0, h->namps * sizeof(amp_t), h->stream));
        hipLaunchKernelGGL(k_tile_lists, dim3((unsigned)nsw), dim3(256), (size_t)((ntiles + 31) / 32) * sizeof(uint32_t), h->stream,
                           in_idx, in_count, (const uint64_t *)h->d_tile_smasks.p, h->n_local, (uint32_t)ntiles, cap,
                           (uint32_t *)h->d_tile_lists.p, (uint32_t *)h->d_tile_counts.p);
        HIPC(h, hipGetLastError());
        lists = (const uint32_t *)h->d_tile_lists.p;
        counts = (const uint32_t *)h->d_tile_counts.p;
    }
    int first = 1;
    size_t k = 0;
    for (const ExSweep &sw : H.tsweeps) {
        const uint32_t *tl = lists ? lists + k * cap : nullptr, *tc = lists ? counts + k : nullptr;
        int rc;
        switch (H.tile_bits) {
        case 10: rc = launch_tile_apply<10>(h, H, sw, in, out, first, ident, tl, tc, cap); break;
        case 11: rc = launch_tile_apply<11>(h, H, sw, in, out, first, ident, tl, tc, cap); break;
        default: rc = launch_tile_apply<12>(h, H, sw, in, out, first, ident, tl, tc, cap); break;
        }
        if (rc) return rc;
        first = 0;
        ++k;
    }
    return OVQE_OK;
}

int init_basis(ovqe_handle h, uint64_t index, double2 one = make_double2(1.0, 0.0)) {
    const uint64_t lmask = local_mask(h);
    const int has = ((index & ~lmask) == h->base) ? 1 : 0;
    hipLaunchKernelGGL(k_init_basis, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, h->state, h->namps,
                       index & lmask, has, one);
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// ---- LDS-tiled multi-op sweeps (sv_tile.hpp) --------------------------------------------------------
template <int M, bool REAL>
int launch_tile(ovqe_handle h, const TilePlan &tp, const TileSeg &sg) {
    constexpr int NT = 1 << TILE_SWEEP_LOG_NT;
    const size_t smem = ((size_t)(REAL ? 8 : 16) << M) + TILE_ROT_CAP * sizeof(RotLds);
    const unsigned grid = (unsigned)(h->namps >> M);
    const bool ntl = h->n_local >= 25;
    static bool attr_done_dev[64] = {};  // function attributes are per device
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sweep<M, NT, true, REAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sweep<M, NT, false, REAL>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    if (ntl) {
        hipLaunchKernelGGL((k_tile_sweep<M, NT, true, REAL>), dim3(grid), dim3(NT), smem, h->stream, (void *)h->state,
                           h->base, sg, (const TileOp *)tp.d_tops.p, (const TileRot *)tp.d_trots.p,
                           (const RotParam *)h->d_rp.p);
    } else {
        hipLaunchKernelGGL((k_tile_sweep<M, NT, false, REAL>), dim3(grid), dim3(NT), smem, h->stream, (void *)h->state,
                           h->base, sg, (const TileOp *)tp.d_tops.p, (const TileRot *)tp.d_trots.p,
                           (const RotParam *)h->d_rp.p);
    }
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// tile sizes: complex 2^10..2^12 amplitudes; real 2^11..2^13 (the segment's masks then hold one bit less, sv_tile.hpp)
int launch_tile_segment(ovqe_handle h, const TilePlan &tp, const TileSeg &sg, bool real = false) {
    const int bits = __builtin_popcountll(sg.smask) + (real ? 1 : 0);
    if (real) {
        switch (bits) {
        case 11: return launch_tile<11, true>(h, tp, sg);
        case 12: return launch_tile<12, true>(h, tp, sg);
        case 13: return launch_tile<13, true>(h, tp, sg);
        }
    } else {
        switch (bits) {
        case 10: return launch_tile<10, false>(h, tp, sg);
        case 11: return launch_tile<11, false>(h, tp, sg);
        case 12: return launch_tile<12, false>(h, tp, sg);
        }
    }
    return fail(h, OVQE_ERR_INVALID, "corrupt tile segment");
}

// Greedy segmentation of the (table-fused) program into tile sweeps: consecutive ops are taken while the union of
// their mixing bits (x mask / gate target) and the mandatory low bits fits the tile; ops that do not fit, and
// segments of a single op, keep their own full-bandwidth sweep.  Commuting runs enter a tile in their OP_TAB form
// (one rotation per active pair pattern, see try_table_op): inside a fused sweep the arithmetic, not HBM, is the
// cost, and the table form does 1/64 of it for a JW double excitation.
int build_tile_plan(ovqe_handle h, const std::vector<SmallOp> &sops, const std::vector<SmallRot> &srots,
                    const std::vector<uint64_t> &sop_zc, TilePlan &tp, bool real = false) {
    tp.tsegs.clear();
    tp.tops.clear();
    tp.trots.assign(srots.size(), TileRot{0, 0, 0});
    tp.plan.clear();
    const int M = tile_bits(h, real);
    const int nops = (int)sops.size();
    const bool tiled = tile_ok(h, real);
    if (!tiled) {
        for (int i = 0; i < nops; ++i) tp.plan.push_back(-1 - i);
        return OVQE_OK;
    }
    const uint64_t lowbits = (1ull << h->opt_tile_low) - 1ull;
    auto need = [&](const SmallOp &op) -> uint64_t {
        switch (op.kind) {
        case OP_PAIR:
        case OP_TAB: return op.x;
        case OP_DIAG: return 0ull;
        case OP_CNOT: return 1ull << op.count;
        default: return 1ull << op.pivot;
        }
    };
    auto is_rot = [](const SmallOp &op) { return op.kind == OP_PAIR || op.kind == OP_DIAG || op.kind == OP_TAB; };
    int i = 0;
    while (i < nops) {
        uint64_t S = lowbits;
        int j = i, nrot = 0;
        while (j < nops) {
            const SmallOp &op = sops[j];
            const uint64_t nb = S | need(op);
            if (__builtin_popcountll(nb) > M) break;
            if (is_rot(op) && nrot + op.count > TILE_ROT_CAP) break;
            if (is_rot(op)) nrot += op.count;
            S = nb;
            ++j;
        }
        if (j - i < 2) {
            tp.plan.push_back(-1 - i);
            ++i;
            continue;
        }
        for (int b = 0; __builtin_popcountll(S) < M; ++b) S |= 1ull << b;  // fill with the lowest free bits
        TileSeg sg = {};
        sg.smask = real ? S >> 1 : S;  // real state: masks in the index space of amplitude pairs
        uint64_t lo = 0, mk = sg.smask;
        for (int k = 0; k < TILE_SWEEP_LOG_NT; ++k) {  // thread bits
            lo |= mk & (0ull - mk);
            mk &= mk - 1ull;
        }
        sg.mask_lo = lo;
        sg.mask_hi = sg.smask & ~lo;
        sg.op0 = (int32_t)tp.tops.size();
        sg.rot0 = sg.rot1 = -1;
        for (int o = i; o < j; ++o) {
            const SmallOp &op = sops[o];
            TileOp t = {};
            t.kind = (int16_t)op.kind;
            if (is_rot(op)) {
                t.x = extract_bits(op.x, S);
                t.pivot = (int16_t)(t.x ? 31 - __builtin_clz(t.x) : 0);
                t.first = op.first;
                t.count = op.count;
                if (sg.rot0 < 0) sg.rot0 = op.first;
                sg.rot1 = op.first + op.count;
                const uint64_t zc = op.kind == OP_TAB ? sop_zc[o] : 0ull;
                t.zc = extract_bits(zc, S);
                for (int r = op.first; r < op.first + op.count; ++r) {
                    // OP_TAB entries: z = the pattern's bits (inside x, hence inside the tile); the run's common z
                    // part outside the tile is a per-tile sign of every entry
                    tp.trots[r].zin = extract_bits(srots[r].z, S);
                    tp.trots[r].zout = op.kind == OP_TAB ? (zc & ~S) : (srots[r].z & ~S);
                }
            } else if (op.kind == OP_CNOT) {
                const int cb = op.first, tbit = op.count;
                t.x = extract_bits(1ull << tbit, S);
                t.pivot = (int16_t)(31 - __builtin_clz(t.x));
                t.first = ((S >> cb) & 1ull) ? (31 - __builtin_clz(extract_bits(1ull << cb, S))) : (-1 - cb);
            } else {
                t.x = extract_bits(1ull << op.pivot, S);
                t.pivot = (int16_t)(31 - __builtin_clz(t.x));
            }
            tp.tops.push_back(t);
        }
        if (sg.rot0 < 0) sg.rot0 = sg.rot1 = 0;
        sg.op1 = (int32_t)tp.tops.size();
        tp.plan.push_back((int32_t)tp.tsegs.size());
        tp.tsegs.push_back(sg);
        i = j;
    }
    int rc = upload(h, tp.d_tops, tp.tops.data(), tp.tops.size() * sizeof(TileOp));
    if (rc) return rc;
    return upload(h, tp.d_trots, tp.trots.data(), tp.trots.size() * sizeof(TileRot));
}

int build_tile_program(ovqe_handle h) { return build_tile_plan(h, h->sops, h->srots, h->sop_zc, h->tp); }

inline RotParam resolve_rot(const SmallRot &sr, const double *theta) {
    const double phi = sr.phi0 + (sr.pidx >= 0 ? sr.coeff * theta[sr.pidx] : 0.0);
    RotParam rp;
    rp.z = sr.z;
    rp.c = std::cos(phi);
    const double s = std::sin(phi);
    rp.s = (sr.ny & 2) ? -s : s;
    rp.odd = sr.ny & 1;
    rp.pad = 0;
    return rp;
}

// Angle table of one evaluation in h->d_rp: [0, S) the entries of the table-fused program (tile sweeps, sequential runs),
// [S, S+R) the original rotations (commuting runs that keep their own sweep run in their sequential form).
int resolve_angles(ovqe_handle h, const double *theta, bool fused_only = false) {   // fused_only: [0, S) is all the caller reads (sector path)
    int rc = OVQE_OK;
    const size_t S = h->srots.size(), R = (fused_only && !h->probe_independent) ? 0 : h->rots.size();
    rc = ensure_rp(h, std::max<size_t>(S + R, 1));
    if (rc) return rc;
    static_assert(sizeof(RotSpec) == sizeof(SmallRot), "RotSpec mirrors SmallRot");
    if (h->probe_independent) {
        auto probe = [](const SmallRot &sr, size_t r) {
            const double f = 0.6180339887498949 * (double)(r + 1);
            const double phi = 0.4 + 0.7 * (f - std::floor(f));
            RotParam rp;
            rp.z = sr.z;
            rp.c = std::cos(phi);
            rp.s = (sr.ny & 2) ? -std::sin(phi) : std::sin(phi);
            rp.odd = sr.ny & 1;
            rp.pad = 0;
            return rp;
        };
        for (size_t r = 0; r < S; ++r) h->h_rp[r] = probe(h->srots[r], r);
        for (size_t r = 0; r < R; ++r) h->h_rp[S + r] = probe(h->rots[r], S + r);
        if (S + R)
            HIPC(h, hipMemcpyAsync(h->d_rp.p, h->h_rp, (S + R) * sizeof(RotParam), hipMemcpyHostToDevice, h->stream));
        return OVQE_OK;
    }
    if (S + h->rots.size() >= 256 && (size_t)h->K <= ovqe_sv::IO_DOUBLES && mapped_io(h, 1)) {
        // angles resolved on the device from the parameter vector (read through the pinned, mapped buffer): no host
        // trigonometry and no table upload on the evaluation path
        std::memcpy(h->h_io, theta, (size_t)h->K * sizeof(double));
        if (S)
            hipLaunchKernelGGL(k_resolve_rots, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, h->stream,
                               (const RotSpec *)h->d_rots.p, (int)S, (const double *)h->d_io, (RotParam *)h->d_rp.p);
        if (R)
            hipLaunchKernelGGL(k_resolve_rots, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, h->stream,
                               (const RotSpec *)h->d_rots_seq.p, (int)R, (const double *)h->d_io,
                               (RotParam *)h->d_rp.p + S);
    } else {
        for (size_t r = 0; r < S; ++r) h->h_rp[r] = resolve_rot(h->srots[r], theta);
        for (size_t r = 0; r < R; ++r) h->h_rp[S + r] = resolve_rot(h->rots[r], theta);
        if (S + R)
            HIPC(h, hipMemcpyAsync(h->d_rp.p, h->h_rp, (S + R) * sizeof(RotParam), hipMemcpyHostToDevice, h->stream));
    }
    return OVQE_OK;
}

// run the compiled program with the streaming kernels (state left in h->state).
// Angle table: [0, S) the entries of the table-fused program (tile sweeps, sequential runs), [S, S+R) the original
// rotations (commuting runs that keep their own sweep run in their sequential form).
int run_program_streaming(ovqe_handle h, const double *theta, bool real = false) {
    int rc = OVQE_OK;
    if (real) {
        if (!h->tp_real_built) {
            const auto t_plan = std::chrono::steady_clock::now();
            rc = build_tile_plan(h, h->sops, h->srots, h->sop_zc, h->tp_real, true);
            if (rc) return rc;
            h->tp_real_built = true;
            if (h->opt_sector_debug & 4)
                fprintf(stderr, "ovqe: tile plan of the real-amplitude program: %.2f ms\n",
                        1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_plan).count());
        }
        hipLaunchKernelGGL(k_init_basis_real, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, (double *)h->state,
                           h->namps, h->hf);
    } else {
        rc = init_basis(h, h->hf, h->init_amp);
    }
    if (rc) return rc;
    const size_t S = h->srots.size();
    rc = resolve_angles(h, theta);
    if (rc) return rc;
    const RotParam *d_rp = (const RotParam *)h->d_rp.p;
    const TilePlan &tp = real ? h->tp_real : h->tp;
    const int nbr = reduce_blocks(h->namps);
    for (const int32_t step : tp.plan) {
        if (step >= 0) {
            rc = launch_tile_segment(h, tp, tp.tsegs[step], real);
            if (rc) return rc;
            continue;
        }
        const SmallOp &op = h->sops[-1 - step];
        if (real) {  // one-op sweeps on the real state
            double *st = (double *)h->state;
            if (op.kind == OP_PAIR) {
                hipLaunchKernelGGL(k_rot_pairs_real, dim3(nbr), dim3(256), 0, h->stream, st, h->namps >> 1, op.pivot, op.x,
                                   h->base, d_rp + op.first, op.count);
            } else if (op.kind == OP_TAB) {
                const SmallOp &src = h->ops[h->sop_src[-1 - step]];
                hipLaunchKernelGGL(k_rot_pairs_real, dim3(nbr), dim3(256), 0, h->stream, st, h->namps >> 1, src.pivot, src.x,
                                   h->base, d_rp + S + src.first, src.count);
            } else if (op.kind == OP_X || op.kind == OP_H) {
                hipLaunchKernelGGL(k_gate_real, dim3(nbr), dim3(256), 0, h->stream, st, h->namps >> 1,
                                   op.kind == OP_X ? 0 : 1, op.pivot, 0);
            } else if (op.kind == OP_CNOT) {
                hipLaunchKernelGGL(k_gate_real, dim3(nbr), dim3(256), 0, h->stream, st, h->namps >> 2, 2, op.first, op.count);
            } else {
                return fail(h, OVQE_ERR_INVALID, "internal: complex op in a real program");
            }
            continue;
        }
        switch (op.kind) {
        case OP_PAIR:
        case OP_DIAG:
            rc = launch_rot_run(h, op.kind == OP_PAIR ? op.x : 0ull, d_rp + op.first, op.count);
            break;
        case OP_TAB: {
            const SmallOp &src = h->ops[h->sop_src[-1 - step]];
            rc = launch_rot_run(h, src.x, d_rp + S + src.first, src.count);
            break;
        }
        case OP_X: rc = launch_gate(h, 0, op.pivot, 0); break;
        case OP_H: rc = launch_gate(h, 1, op.pivot, 0); break;
        case OP_CNOT: rc = launch_gate(h, 2, op.first, op.count); break;
        default: rc = fail(h, OVQE_ERR_INVALID, "corrupt program");
        }
        if (rc) return rc;
    }
    // the pinned table may be rewritten by the next call: make sure the copy has been consumed
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
}

void tridiag_lowest(const std::vector<double> &a, const std::vector<double> &b, int m, double *lam, std::vector<double> &s);

#include "sector_host.inc"

// sector path: the tables of a (program, Hamiltonian) pair are built at its second evaluation (energy or gradient), so
// one-shot callers never pay for them
int sector_prepare(ovqe_handle h, bool eager = false) {
    if (!h->opt_sector || h->n_local < h->opt_sector_min_qubits) return OVQE_OK;
    SectorEngine &E = h->sec;
    if (E.prog_version != h->prog_version || E.ham_version != h->ham.version) {
        free_sector(E);
        E.disabled = false;
        E.seen = 0;
        E.probe_mode = 0;
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
    free_sector(E);
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

// ---- gate programs ---------------------------------------------------------------------------------
void push_gate(ovqe_handle h, int opcode, int b0, int b1) {
    SmallOp op = {};
    if (opcode == OVQE_GATE_CNOT) {
        op.x = 1ull << b1;
        op.kind = OP_CNOT;
        op.first = b0;
        op.count = b1;
        op.pivot = b1;
    } else {
        op.x = 1ull << b0;
        op.kind = opcode == OVQE_GATE_X ? OP_X : OP_H;
        op.pivot = b0;
    }
    h->ops.push_back(op);
}

void push_literal_gate(ovqe_handle h, int opcode, int b0, int b1, double ascale, double aconst, int32_t pidx) {
    const uint64_t bit = 1ull << b0;
    switch (opcode) {
    case OVQE_GATE_RX: push_rotation(h, bit, 0, 0.5 * ascale, 0.5 * aconst, pidx); break;
    case OVQE_GATE_RY: push_rotation(h, bit, bit, 0.5 * ascale, 0.5 * aconst, pidx); break;
    case OVQE_GATE_RZ: push_rotation(h, 0, bit, 0.5 * ascale, 0.5 * aconst, pidx); break;
    default: push_gate(h, opcode, b0, b1);
    }
}

// the literal gate list, one op per gate (consecutive rotations with equal x masks still share a sweep)
int compile_gate_program_literal(ovqe_handle h, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1,
                                 const double *ascale, const double *aconst, const int32_t *pidx) {
    h->prog_set = false;
    h->ops.clear();
    h->rots.clear();
    h->init_amp = make_double2(1.0, 0.0);
    for (int64_t g = 0; g < G; ++g) push_literal_gate(h, opcode[g], b0[g], b1[g], ascale[g], aconst[g], pidx[g]);
    return finish_program(h);
}

// Clifford-frame compilation.  X, H, CNOT and the fixed quarter-turn rotations RX/RY/RZ(+-pi/2) of the reference's
// templates (ref:openvqe/common_files/circuit.py:13-106) are Clifford gates: with C_<k the product of the Clifford
// gates before rotation k,   U = C_total * prod_k exp(-i phi_k C_<k^dagger P_k C_<k),   and the conjugated generator
// is again a Pauli string (Heisenberg picture; tracked as images of X_q, Z_q).  When C_total is the identity — every
// QUCCSD template returns the frame to the identity: its Clifford part only routes the excitation — the literal gate
// list is EXACTLY a sequence of Pauli rotations, which then take the fused same-x / commuting-run / real-mode /
// support-compacted paths.  C_total = e^{i alpha}: the phase is read from one execution of the Clifford part alone
// on |hf>; anything but alpha = 0 (or a frame that does not close) keeps the literal program.
// clifford_frame = 2 (tests): always use the frame form and append the Clifford part literally.

int install_hamdev(ovqe_handle h, HamDev &H, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff, double constant) {
    H.set = false;
    int rc = build_groups(h, T, x, z, coeff, nullptr, false, H.groups, H.terms);
    if (rc) return rc;
    rc = upload(h, H.d_groups, H.groups.data(), H.groups.size() * sizeof(HGroup));
    if (!rc) rc = upload(h, H.d_terms, H.terms.data(), H.terms.size() * sizeof(HTerm));
    if (rc) return rc;
    H.constant = constant;
    H.set = true;
    H.tile_bits = -1;  // tile cover rebuilt on first use
    H.version = ++h->ham_versions;
    return OVQE_OK;
}

// stored Hamiltonian through the open Clifford frame: P = i^{|x&z|} X^x Z^z -> product of the images of its X_q and Z_q
int install_conjugated_hamiltonian(ovqe_handle h) {
    h->ham_conj.set = false;
    if (!h->frame_open || !h->ham.set) return OVQE_OK;
    const int n = h->n_local;
    const size_t T = h->user_x.size();
    std::vector<uint64_t> cx(T), cz(T);
    std::vector<double> cc(T);
    auto img = [&](int idx) { return PauliRaw{h->frame_img[3 * idx], h->frame_img[3 * idx + 1], (int)h->frame_img[3 * idx + 2]}; };
    for (size_t t = 0; t < T; ++t) {
        const uint64_t x = h->user_x[t], z = h->user_z[t];
        PauliRaw acc{0, 0, __builtin_popcountll(x & z) & 3};   // the i^{|x&z|} of the Hermitian string
        for (int q = 0; q < n; ++q)
            if ((x >> q) & 1) acc = pauli_mul(acc, img(q));
        for (int q = 0; q < n; ++q)
            if ((z >> q) & 1) acc = pauli_mul(acc, img(n + q));
        const int rel = (acc.k - __builtin_popcountll(acc.x & acc.z)) & 3;
        if (rel & 1) return fail(h, OVQE_ERR_INVALID, "internal: non-Hermitian conjugated Hamiltonian term");
        cx[t] = acc.x;
        cz[t] = acc.z;
        cc[t] = rel ? -h->user_c[t] : h->user_c[t];
    }
    return install_hamdev(h, h->ham_conj, (int64_t)T, cx.data(), cz.data(), cc.data(), h->user_const);
}

int compile_gate_program_frame(ovqe_handle h, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1,
                               const double *ascale, const double *aconst, const int32_t *pidx, bool *done) {
    *done = false;
    const int n = h->n_local;
    FrameTrack F;
    if (!track_clifford_frame(n, G, opcode, b0, b1, ascale, aconst, pidx, F))
        return fail(h, OVQE_ERR_INVALID, "internal: non-Hermitian conjugated generator");
    std::vector<PauliRaw> &ix = F.ix, &iz = F.iz;
    std::vector<FrameEmit> &emitted = F.emitted;
    std::vector<int64_t> &tail = F.tail;  // gates folded into the frame (the Clifford part, original order)
    const bool closed = F.closed;
    const bool forced = h->opt_clifford_frame == 2;
    if (!closed && !forced) {
        // Open frame (e.g. interleaved CNOT ladders, which the reference's ladder code does not undo): the program is the
        // rotation sequence; the net Clifford operator goes into the Hamiltonian for energies (install_conjugated_hamiltonian)
        // and behind the rotations, gate by gate, for ovqe_prepare_state.  Option 3 keeps the literal program instead.
        if (h->opt_clifford_frame == 3 || n > 63) return OVQE_OK;
        h->prog_set = false;
        h->ops.clear();
        h->rots.clear();
        h->init_amp = make_double2(1.0, 0.0);
        for (const FrameEmit &e : emitted) push_rotation(h, e.x, e.z, e.coeff, e.phi0, e.pidx);
        h->frame_img.assign((size_t)6 * n, 0);
        for (int q = 0; q < n; ++q) {
            h->frame_img[3 * q] = ix[q].x;
            h->frame_img[3 * q + 1] = ix[q].z;
            h->frame_img[3 * q + 2] = (uint64_t)ix[q].k;
            h->frame_img[3 * (n + q)] = iz[q].x;
            h->frame_img[3 * (n + q) + 1] = iz[q].z;
            h->frame_img[3 * (n + q) + 2] = (uint64_t)iz[q].k;
        }
        h->tail_gates.clear();
        for (const int64_t g : tail) {
            h->tail_gates.push_back(opcode[g]);
            h->tail_gates.push_back(b0[g]);
            h->tail_gates.push_back(b1[g]);
            h->tail_gates.push_back(aconst[g] > 0 ? 1 : -1);
        }
        h->frame_open = true;
        *done = true;
        int rc = finish_program(h);
        if (!rc) rc = install_conjugated_hamiltonian(h);
        if (rc) h->frame_open = false;
        return rc;
    }
    bool drop_tail = false;
    double2 phase = make_double2(1.0, 0.0);
    h->init_amp = phase;
    if (closed && !forced) {
        // global phase of the Clifford part: execute it alone on |hf>
        h->prog_set = false;
        h->ops.clear();
        h->rots.clear();
        double2 amp = make_double2(0.0, 0.0);
        std::complex<double> amp_host(0.0, 0.0);
        const bool on_host = !tail.empty() && h->opt_clifford_phase_host && n <= 63 &&
                             clifford_amplitude_on_host(h->hf, tail, opcode, b0, b1, aconst, &amp_host);
        if (on_host) amp = make_double2(amp_host.real(), amp_host.imag());
        if (!tail.empty() && !on_host)
            for (const int64_t g : tail) push_literal_gate(h, opcode[g], b0[g], b1[g], ascale[g], aconst[g], pidx[g]);
        if (!tail.empty()) {
            // (device form) the Clifford part is installed as the handle's program only for this one execution: whatever happens
            // below, the handle must not keep reporting it as a valid program (prog_set stays false until the final
            // program is installed by the last finish_program)
            int rc = OVQE_OK;
            if (!on_host) {
                rc = finish_program(h);
                if (!rc) {
                    std::vector<double> zero((size_t)std::max(h->K, 1), 0.0);
                    rc = run_program_streaming(h, zero.data());
                }
                if (!rc) {
                    const hipError_t e = hipMemcpy(&amp, h->state + h->hf, sizeof(double2), hipMemcpyDeviceToHost);
                    if (e != hipSuccess) rc = fail(h, OVQE_ERR_HIP, hipGetErrorString(e));
                }
            }
            h->prog_set = false;
            if (rc) {
                h->ops.clear();
                h->rots.clear();
                return rc;
            }
            // the Clifford part maps |hf> to a phase times |hf>; the rounding of its quarter turns accumulates with
            // the length of the list (measured 5e-12 after 49 272 gates), hence a tolerance that grows with it
            const double nrm2 = amp.x * amp.x + amp.y * amp.y;
            if (std::fabs(nrm2 - 1.0) > 1e-12 + 1e-15 * (double)tail.size()) return OVQE_OK;  // literal
            const double inv = 1.0 / std::sqrt(nrm2);
            phase = make_double2(amp.x * inv, amp.y * inv);
        }
        drop_tail = true;
    }
    h->prog_set = false;
    h->ops.clear();
    h->rots.clear();
    for (const FrameEmit &e : emitted) push_rotation(h, e.x, e.z, e.coeff, e.phi0, e.pidx);
    if (!drop_tail)
        for (const int64_t g : tail) push_literal_gate(h, opcode[g], b0[g], b1[g], ascale[g], aconst[g], pidx[g]);
    *done = true;
    h->init_amp = phase;  // energies do not depend on it; ovqe_prepare_state reproduces the literal circuit's phase
    return finish_program(h);
}

bool use_small_path(ovqe_handle h, int64_t B) {
    if (h->n_global != 0) return false;
    if (h->opt_force_path == 1) return h->n_local <= 16;
    if (h->opt_force_path == 2) return false;
    if (h->n_local <= h->opt_small_max) return true;
    return h->n_local <= h->opt_small_batch_max && B >= 32;
}

template <bool REAL, bool LDS, int NT, int LBITS>
int launch_small(ovqe_handle h, const SmallArgs &A, int grid, size_t smem) {
    static bool attr_done_dev[64] = {};  // function attributes are per device
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_small_vqe<REAL, LDS, NT, LBITS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_small_vqe<REAL, LDS, NT, LBITS>), dim3(grid), dim3(NT), smem, h->stream, A,
                       h->cur_theta, (const SmallOp *)h->d_ops.p, (const SmallRot *)h->d_rots.p,
                       (const SmallSeg *)h->d_segs.p, (const ExpGroup *)h->d_egroups.p, (const ExpChunk *)h->d_echunks.p,
                       (const ExpTerm *)h->d_eterms.p, (const FlatItem *)h->d_eflat.p, (const uint16_t *)h->d_stream.p,
                       h->d_workspace.p,
                       h->cur_energies);
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// free-index-space expectation tables (sv_small.hpp): one entry per (x-group, pattern of the x-position
// bits) with a non-vanishing coefficient table; the fixed positions are squeezed out of the z masks, the
// sign of parity(x & z), i^ny and the pattern-dependent sign are folded into the coefficients, terms with
// equal outside masks are merged, terms bucketed by the 3 free-index bits above the thread bits
int build_exp_tables(ovqe_handle h, int lbits, bool real) {
    if (h->exp_lbits == lbits && h->exp_real == (int)real && h->exp_ham_version == h->ham.version) return OVQE_OK;
    std::vector<ExpGroup> eg;
    std::vector<ExpTerm> et;
    std::vector<FlatItem> flat;
    const int n = h->n_local;
    auto squeeze = [](uint64_t z, const int *pos, int np) {  // remove the bit positions pos[] (ascending)
        for (int f = np - 1; f >= 0; --f) {
            const uint64_t low = (1ull << pos[f]) - 1ull;
            z = ((z >> (pos[f] + 1)) << pos[f]) | (z & low);
        }
        return z;
    };
    auto emit = [&](uint32_t x, uint32_t ibits, const int *pos, int np, std::vector<ExpTerm> &list,
                    const std::vector<uint64_t> *zc_index_space) {
        if (list.empty()) return;
        if (zc_index_space && list.size() == 1 && list[0].ci == 0.0 && x != 0 && n <= 16) {
            // single real term: entry-per-lane flat item(s)
            const uint32_t nk = 1u << (n - np);
            const uint32_t nslices = nk >= 64 ? 4 : 1;
            for (uint32_t sl = 0; sl < nslices; ++sl) {
                FlatItem fi;
                fi.x = (uint16_t)x;
                fi.ibits = (uint16_t)ibits;
                fi.zc = (uint16_t)(*zc_index_space)[0];
                fi.slice = (uint16_t)sl;
                fi.count = nk / nslices;
                fi.stride = nslices;
                fi.c = 2.0 * list[0].cr;
                flat.push_back(fi);
            }
            return;
        }
        ExpGroup out = {};
        out.x = x;
        out.ibits = ibits;
        out.fixmask = 0;
        for (int f = 0; f < np; ++f) out.fixmask |= 1u << pos[f];
        out.t0 = (int32_t)et.size();
        std::vector<ExpTerm> bucket[8];
        for (const ExpTerm &e : list) bucket[(e.zk >> lbits) & 7].push_back(e);
        int off = 0;
        for (int b = 0; b < 8; ++b) {
            out.off[b] = off;
            for (const ExpTerm &e : bucket[b]) et.push_back(e);
            off += (int)bucket[b].size();
        }
        out.off[8] = off;
        eg.push_back(out);
    };
    for (const HGroup &g : h->ham.groups) {
        const uint64_t x = g.x;
        const int w = __builtin_popcountll(x);
        int pos[64], np = 0;
        for (int b = 0; b < 64; ++b)
            if ((x >> b) & 1) pos[np++] = b;
        // folded coefficients: D uses parity(j & z), j = i ^ x  ->  parity(i & z) ^ parity(x & z)
        std::vector<HTerm> ts;
        for (int t = g.t0; t < g.t1; ++t) {
            HTerm ht = h->ham.terms[t];
            const int ny = __builtin_popcountll(x & ht.z);
            if (real && (ny & 1)) continue;  // imaginary antisymmetric string: zero on a real state
            if (ny & 1) {
                ht.cr = -ht.cr;
                ht.ci = -ht.ci;
            }
            ts.push_back(ht);
        }
        if (ts.empty()) continue;
        if (w == 0 || w > 7 || !h->opt_table_fusion) {
            // dense form: only the pivot is fixed (diag group: nothing fixed)
            const int piv = w ? pos[np - 1] : 0;
            std::vector<ExpTerm> list;
            for (const HTerm &ht : ts) {
                ExpTerm e;
                e.zk = (uint32_t)(w ? squeeze(ht.z, &piv, 1) : ht.z);
                e.pad = 0;
                e.cr = ht.cr;
                e.ci = ht.ci;
                list.push_back(e);
            }
            emit((uint32_t)x, 0, &piv, w ? 1 : 0, list, nullptr);
            continue;
        }
        for (uint32_t e = 0; e < (1u << (w - 1)); ++e) {
            uint64_t ibits = 0;
            for (int f = 0; f < w - 1; ++f)
                if ((e >> f) & 1) ibits |= 1ull << pos[f];
            std::vector<ExpTerm> list;  // merged by outside mask
            std::vector<uint64_t> zcs;  // the outside masks in index space, parallel to list
            for (const HTerm &ht : ts) {
                const uint64_t zc = ht.z & ~x;
                const double sg = (__builtin_popcountll(ibits & ht.z) & 1) ? -1.0 : 1.0;
                const uint32_t zk = (uint32_t)squeeze(zc, pos, np);
                bool found = false;
                for (ExpTerm &q : list)
                    if (q.zk == zk) {
                        q.cr += sg * ht.cr;
                        q.ci += sg * ht.ci;
                        found = true;
                        break;
                    }
                if (!found) {
                    ExpTerm q;
                    q.zk = zk;
                    q.pad = 0;
                    q.cr = sg * ht.cr;
                    q.ci = sg * ht.ci;
                    list.push_back(q);
                    zcs.push_back(zc);
                }
            }
            std::vector<ExpTerm> nz;
            std::vector<uint64_t> nzc;
            for (size_t q = 0; q < list.size(); ++q)
                if (list[q].cr != 0.0 || list[q].ci != 0.0) {  // exact cancellations only
                    nz.push_back(list[q]);
                    nzc.push_back(zcs[q]);
                }
            emit((uint32_t)x, (uint32_t)ibits, pos, np, nz, &nzc);
        }
    }
    // chunks of general groups whose terms fit the LDS staging area (the idle rotation table)
    const int stage_cap = (int)((size_t)h->cs_capacity * sizeof(RotLds) / sizeof(ExpTerm));
    std::vector<ExpChunk> chunks;
    {
        ExpChunk cur = {0, 0, 0, 0};
        for (int g = 0; g < (int)eg.size(); ++g) {
            const int gt1 = eg[g].t0 + eg[g].off[8];
            if (eg[g].off[8] > stage_cap) return fail(h, OVQE_ERR_INVALID, "x-group with too many terms for the fused kernel");
            if (gt1 - cur.t0 > stage_cap) {
                if (cur.g1 > cur.g0) chunks.push_back(cur);
                cur = {g, g, eg[g].t0, eg[g].t0};
            }
            cur.g1 = g + 1;
            cur.t1 = gt1;
        }
        if (cur.g1 > cur.g0) chunks.push_back(cur);
    }
    int rc = upload(h, h->d_egroups, eg.data(), eg.size() * sizeof(ExpGroup));
    if (!rc) rc = upload(h, h->d_eterms, et.data(), et.size() * sizeof(ExpTerm));
    if (!rc) rc = upload(h, h->d_echunks, chunks.data(), chunks.size() * sizeof(ExpChunk));
    if (!rc) rc = upload(h, h->d_eflat, flat.data(), flat.size() * sizeof(FlatItem));
    if (rc) return rc;
    h->exp_lbits = lbits;
    h->exp_ham_version = h->ham.version;
    h->exp_real = (int)real;
    h->exp_ngroups = (int)eg.size();
    h->exp_nchunks = (int)chunks.size();
    h->exp_nflat = (int)flat.size();
    return OVQE_OK;
}

// B evaluations with the fused kernel; energies -> host
// host <-> device traffic of a small batch through the mapped buffer: [theta B x K][energies B]
bool mapped_io(ovqe_handle h, int64_t B) {
    // measured: zero-copy wins up to the 64-KiB buffer (H2O: 16 evaluations 63 us against 89 us through copies)
    if ((size_t)B * (size_t)(h->K + 1) > ovqe_sv::IO_DOUBLES || B > 1024) return false;
    if (!h->h_io) {
        if (hipHostMalloc((void **)&h->h_io, ovqe_sv::IO_DOUBLES * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            h->h_io = nullptr;
            (void)hipGetLastError();
            return false;
        }
        if (hipHostGetDevicePointer((void **)&h->d_io, h->h_io, 0) != hipSuccess) {
            (void)hipHostFree(h->h_io);
            h->h_io = nullptr;
            (void)hipGetLastError();
            return false;
        }
    }
    return true;
}

int run_small(ovqe_handle h, int64_t B, const double *theta, double *energies, bool on_device = false) {
    const int n = h->n_local;
    bool real = h->opt_real_mode != 0;
    for (const SmallRot &sr : h->rots) real = real && (sr.ny & 1);
    const size_t amp_bytes = real ? sizeof(double) : sizeof(amp_t);
    const size_t state_bytes = (size_t)h->namps * amp_bytes;
    const bool lds_state = state_bytes <= 128 * 1024;
    const uint64_t npairs = h->namps >> 1;
    int nt = (!lds_state || npairs >= 1024) ? 1024 : (npairs >= 256 ? 256 : 64);
    if (lds_state && h->opt_small_threads && npairs >= (uint64_t)h->opt_small_threads &&
        (h->opt_small_threads == 256 || h->opt_small_threads == 512 || h->opt_small_threads == 1024))
        nt = h->opt_small_threads;
    const int lbits = nt == 1024 ? 10 : (nt == 512 ? 9 : (nt == 256 ? 8 : 6));
    int rc = build_exp_tables(h, lbits, real);
    if (rc) return rc;
    int max_slices = (int)std::max<size_t>(1, std::min<size_t>(512, ((size_t)512 << 20) / state_bytes));
    if (lds_state) max_slices = 1024;
    const int grid = (int)std::min<int64_t>(B, max_slices);
    rc = ensure(h, h->d_workspace, lds_state ? 256 : (size_t)grid * state_bytes);
    if (rc) return rc;
    h->cur_theta = theta;
    h->cur_energies = energies;
    const bool zero_copy = !on_device && mapped_io(h, B);
    const bool poll = zero_copy && B <= 256 && h->opt_poll_result;
    if (zero_copy) {
        if (h->K > 0) std::memcpy(h->h_io, theta, (size_t)B * h->K * sizeof(double));
        if (poll) poll_arm(h->h_io + (size_t)B * h->K, B);
        h->cur_theta = h->d_io;
        h->cur_energies = h->d_io + (size_t)B * h->K;
    } else if (!on_device) {
        rc = ensure(h, h->d_theta, (size_t)B * std::max(1, h->K) * sizeof(double));
        if (rc) return rc;
        rc = ensure(h, h->d_energies, (size_t)B * sizeof(double));
        if (rc) return rc;
        if (h->K > 0)
            HIPC(h, hipMemcpyAsync(h->d_theta.p, theta, (size_t)B * h->K * sizeof(double), hipMemcpyHostToDevice,
                                   h->stream));
        h->cur_theta = (const double *)h->d_theta.p;
        h->cur_energies = (double *)h->d_energies.p;
    }
    SmallArgs A;
    A.n = n;
    A.K = h->K;
    A.nsegs = (int)h->segs.size();
    A.ngroups = h->exp_ngroups;
    A.nchunks = h->exp_nchunks;
    A.nflat = h->exp_nflat;
    A.cs_capacity = h->cs_capacity;
    A.B = B;
    A.constant = h->ham.constant;
    A.hf = h->hf;
    const size_t smem = (lds_state ? state_bytes : 0) + (size_t)h->cs_capacity * sizeof(RotLds) + SMALL_OPS_CAP * sizeof(SmallOp) + 16 * sizeof(double2);
    if (!zero_copy) HIPC(h, hipEventRecord(h->ev0, h->stream));
    if (real) {
        if (!lds_state) rc = launch_small<true, false, 1024, 10>(h, A, grid, smem);
        else if (nt == 1024) rc = launch_small<true, true, 1024, 10>(h, A, grid, smem);
        else if (nt == 512) rc = launch_small<true, true, 512, 9>(h, A, grid, smem);
        else if (nt == 256) rc = launch_small<true, true, 256, 8>(h, A, grid, smem);
        else rc = launch_small<true, true, 64, 6>(h, A, grid, smem);
    } else {
        if (!lds_state) rc = launch_small<false, false, 1024, 10>(h, A, grid, smem);
        else if (nt == 1024) rc = launch_small<false, true, 1024, 10>(h, A, grid, smem);
        else if (nt == 512) rc = launch_small<false, true, 512, 9>(h, A, grid, smem);
        else if (nt == 256) rc = launch_small<false, true, 256, 8>(h, A, grid, smem);
        else rc = launch_small<false, true, 64, 6>(h, A, grid, smem);
    }
    if (rc) return rc;
    if (zero_copy) {
        if (!poll || !poll_mapped_slots(h->h_io + (size_t)B * h->K, B)) HIPC(h, hipStreamSynchronize(h->stream));
        std::memcpy(energies, h->h_io + (size_t)B * h->K, (size_t)B * sizeof(double));
        h->last_batch_ms = 0.f;  // not timed: no events on the latency path
        return OVQE_OK;
    }
    HIPC(h, hipEventRecord(h->ev1, h->stream));
    if (!on_device)
        HIPC(h, hipMemcpyAsync(energies, h->cur_energies, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipEventElapsedTime(&h->last_batch_ms, h->ev0, h->ev1));
    return OVQE_OK;
}

// ---- support-compacted path (sv_sparse.hpp) -------------------------------------------------------------------
// Propagate the reachable support of |hf> through the OP_TAB ops and restate program and Hamiltonian on compact
// indices.  Returns with h->sp_valid = false when the structure is absent (then the dense kernels run).
int build_sparse_program(ovqe_handle h) {
    h->sp_tried = true;
    h->sp_valid = false;
    if (!h->opt_sparse || !h->opt_table_fusion || !h->opt_real_mode || h->n_global != 0 || h->n_local > 40) return OVQE_OK;
    if (!h->prog_set || !h->ham.set || h->K <= 0) return OVQE_OK;
    for (const SmallRot &sr : h->rots)
        if (!(sr.ny & 1)) return OVQE_OK;  // real mode only
    for (const SmallOp &op : h->sops)
        if (op.kind != OP_TAB || op.count > 127) return OVQE_OK;
    const int MAXM = 4096;
    std::unordered_map<uint64_t, int> id;
    std::vector<uint64_t> S;
    auto get = [&](uint64_t a) {
        auto it = id.find(a);
        if (it != id.end()) return it->second;
        const int k = (int)S.size();
        id.emplace(a, k);
        S.push_back(a);
        return k;
    };
    get(h->hf);
    std::vector<SpOp> ops;
    std::vector<uint32_t> pairs;
    for (const SmallOp &op : h->sops) {
        const uint64_t x = op.x, pbit = 1ull << (63 - __builtin_clzll(x));
        SpOp so;
        so.first = (int32_t)pairs.size();
        so.tab0 = op.first;
        so.pad = 0;
        const size_t s0 = S.size();
        std::unordered_set<uint64_t> seen;
        for (size_t k = 0; k < s0; ++k) {
            const uint64_t a = S[k];
            const uint64_t i0 = (a & pbit) ? (a ^ x) : a;
            if (!seen.insert(i0).second) continue;
            for (int p = 0; p < op.count; ++p) {
                if ((i0 & x) != h->srots[op.first + p].z) continue;
                const int ci = get(i0), cj = get(i0 ^ x);
                if ((int)S.size() > MAXM) return OVQE_OK;
                const uint32_t sgn = (__builtin_popcountll(i0 & (uint64_t)op.zc) & 1) ? 1u : 0u;
                pairs.push_back((uint32_t)ci | ((uint32_t)cj << 12) | (sgn << 24) | ((uint32_t)p << 25));
                break;
            }
        }
        so.npairs = (int32_t)pairs.size() - so.first;
        if (so.npairs > 0) ops.push_back(so);
    }
    const int m = (int)S.size();
    // Hamiltonian restricted to the support (real mode: even-ny terms; pair counted once -> factor 2)
    std::vector<SpEntry> entries;
    for (const HGroup &g : h->ham.groups) {
        const uint64_t x = g.x;
        const uint64_t pbit = x ? 1ull << (63 - __builtin_clzll(x)) : 0;
        for (int k = 0; k < m; ++k) {
            const uint64_t a = S[k];
            if (x && (a & pbit)) continue;
            const uint64_t b = a ^ x;
            int kb = k;
            if (x) {
                auto it = id.find(b);
                if (it == id.end()) continue;
                kb = it->second;
            }
            double d = 0.0;
            for (int t = g.t0; t < g.t1; ++t) {
                const HTerm &ht = h->ham.terms[t];
                if (__builtin_popcountll(x & ht.z) & 1) continue;  // odd #Y: zero on a real state
                d += (__builtin_popcountll(b & ht.z) & 1) ? -ht.cr : ht.cr;
            }
            if (d == 0.0) continue;
            SpEntry e;
            e.ij = (uint32_t)k | ((uint32_t)kb << 12);
            e.pad = 0;
            e.c = x ? 2.0 * d : d;
            entries.push_back(e);
            if (entries.size() > (size_t)16 << 20) return OVQE_OK;
        }
    }
    // ---- compact numbering against LDS bank conflicts in the circuit (round 3) ---------------------------------------------------
    // The lanes of an evaluation rotate the (up to 32) pairs of an op with ONE ds_read_b64 per member: 32 lanes against 32
    // bank pairs, bank = compact index mod 32.  In discovery order the members of an op collide (rocprofv3: 55 % of the LDS
    // cycles of k_sparse_vqe were conflict replays).  The numbering is the host's to choose: residues mod 32 are assigned by
    // a deterministic local search (swap two elements' residues, keep the swap when the sum over ops and sides of the
    // colliding lane pairs does not grow, with a little annealing) and the support is padded to a multiple of 32 slots.
    int mp = m, hf_slot = 0;
    h->sp_conflicts_before = h->sp_conflicts_after = 0;
    if (h->opt_sparse_renumber && m > 32 && m <= 4064 && !ops.empty()) {
        const int nh = 2 * (int)ops.size();
        const int rows = (m + 31) / 32;
        std::vector<std::array<uint16_t, 32>> hist((size_t)nh);
        for (auto &a : hist) a.fill(0);
        std::vector<int> cap((size_t)nh);
        std::vector<std::vector<int>> inc((size_t)m);
        for (size_t o = 0; o < ops.size(); ++o) {
            cap[2 * o] = cap[2 * o + 1] = (ops[o].npairs + 31) / 32;
            for (int k = 0; k < ops[o].npairs; ++k) {
                const uint32_t pw = pairs[(size_t)ops[o].first + k];
                inc[pw & 0xfffu].push_back((int)(2 * o));
                inc[(pw >> 12) & 0xfffu].push_back((int)(2 * o + 1));
            }
        }
        std::vector<int> res((size_t)m), count(32, 0);
        for (int k = 0; k < m; ++k) {
            res[k] = k & 31;
            ++count[k & 31];
            for (int hid : inc[k]) ++hist[hid][k & 31];
        }
        auto excess = [](int n, int c) { return n > c ? (int64_t)(n - c) * (n - c + 1) / 2 : (int64_t)0; };
        int64_t cost = 0;
        for (int hid = 0; hid < nh; ++hid)
            for (int r = 0; r < 32; ++r) cost += excess(hist[hid][r], cap[hid]);
        h->sp_conflicts_before = cost;
        auto move = [&](int e, int to) {   // -> change of the cost
            const int from = res[e];
            int64_t d = 0;
            for (int hid : inc[e]) {
                auto &hh = hist[hid];
                d += excess(hh[from] - 1, cap[hid]) - excess(hh[from], cap[hid]) + excess(hh[to] + 1, cap[hid]) - excess(hh[to], cap[hid]);
                --hh[from];
                ++hh[to];
            }
            res[e] = to;
            return d;
        };
        size_t total_inc = 0;
        for (const auto &v : inc) total_inc += v.size();
        const double avg_inc = std::max(1.0, (double)total_inc / m);
        const int64_t proposals = cost ? (int64_t)std::min(400.0 * m, 6e7 / avg_inc) : 0;
        uint64_t rng = 0x9e3779b97f4a7c15ull;
        auto next = [&]() {
            rng ^= rng << 13;
            rng ^= rng >> 7;
            rng ^= rng << 17;
            return rng;
        };
        for (int64_t it = 0; it < proposals && cost > 0; ++it) {
            const double temp = 0.6 * (1.0 - (double)it / (double)proposals) + 0.02;
            const int a = (int)(next() % (uint64_t)m);
            int64_t d;
            int b = -1, ra = res[a], rb;
            if ((next() & 3u) == 0) {                       // move into a residue class with a free slot
                rb = (int)(next() & 31u);
                if (rb == ra || count[rb] >= rows) continue;
                d = move(a, rb);
            } else {                                        // swap residues with another element
                b = (int)(next() % (uint64_t)m);
                rb = res[b];
                if (rb == ra) continue;
                d = move(a, rb);
                d += move(b, ra);
            }
            const bool accept = d <= 0 || (double)(next() >> 11) * (1.0 / 9007199254740992.0) < std::exp(-(double)d / temp);
            if (accept) {
                cost += d;
                if (b < 0) {
                    --count[ra];
                    ++count[rb];
                }
            } else {
                move(a, ra);
                if (b >= 0) move(b, rb);
            }
        }
        h->sp_conflicts_after = cost;
        // slots: residue + 32 * (rank inside the residue class, by discovery order)
        std::vector<int> slot((size_t)m), fill(32, 0);
        for (int k = 0; k < m; ++k) slot[k] = res[k] + 32 * fill[res[k]]++;
        mp = 32 * rows;
        hf_slot = slot[0];
        for (uint32_t &pw : pairs)
            pw = (pw & ~0xffffffu) | (uint32_t)slot[pw & 0xfffu] | ((uint32_t)slot[(pw >> 12) & 0xfffu] << 12);
        for (SpEntry &e : entries) e.ij = (uint32_t)slot[e.ij & 0xfffu] | ((uint32_t)slot[(e.ij >> 12) & 0xfffu] << 12);
        // (chunks of 32 are split into two halves of 16 below, for the stores)
        // ops with more than 32 pairs: chunks of 32 with distinct residues on both sides where the pairs allow it
        for (const SpOp &so : ops) {
            if (so.npairs <= 32) continue;
            std::vector<uint32_t> left(pairs.begin() + so.first, pairs.begin() + so.first + so.npairs), out;
            out.reserve(left.size());
            while (!left.empty()) {
                uint32_t ui = 0, uj = 0;
                std::vector<uint32_t> rest;
                size_t taken = 0;
                for (uint32_t pw : left) {
                    const uint32_t bi = pw & 31u, bj = (pw >> 12) & 31u;
                    if (taken < 32 && !((ui >> bi) & 1u) && !((uj >> bj) & 1u)) {
                        ui |= 1u << bi;
                        uj |= 1u << bj;
                        out.push_back(pw);
                        ++taken;
                    } else {
                        rest.push_back(pw);
                    }
                }
                while (taken < 32 && !rest.empty()) {   // pad the chunk so that later chunks stay aligned
                    out.push_back(rest.back());
                    rest.pop_back();
                    ++taken;
                }
                left.swap(rest);
            }
            std::copy(out.begin(), out.end(), pairs.begin() + so.first);
        }
        // The two ds_write_b64 of a rotation are served in groups of 16 lanes against 16 bank pairs (bank = slot mod 16): inside
        // every chunk of 32 pairs the two halves of 16 are chosen so that slots equal mod 16 — at most two per side once the
        // residues mod 32 are distinct — fall into different halves (twins on the first and on the second index form paths and
        // even cycles: two-colourable; greedy here, balanced halves).
        for (const SpOp &so : ops) {
            for (int c0 = 0; c0 < so.npairs; c0 += 32) {
                const int cn = std::min(32, so.npairs - c0);
                if (cn <= 1) continue;
                uint32_t *pw = pairs.data() + so.first + c0;
                std::vector<uint32_t> half[2];
                int cnt_i[2][16] = {}, cnt_j[2][16] = {};
                const int cap0 = std::min(16, cn), cap1 = cn - std::min(16, cn) < 0 ? 0 : 16;
                (void)cap1;
                for (int k = 0; k < cn; ++k) {
                    const uint32_t bi = pw[k] & 15u, bj = (pw[k] >> 12) & 15u;
                    const int c0s = cnt_i[0][bi] + cnt_j[0][bj], c1s = cnt_i[1][bi] + cnt_j[1][bj];
                    int side = c0s < c1s ? 0 : (c1s < c0s ? 1 : (half[0].size() <= half[1].size() ? 0 : 1));
                    if ((int)half[side].size() >= 16) side ^= 1;
                    if (side == 0 && (int)half[0].size() >= cap0) side = 1;
                    half[side].push_back(pw[k]);
                    ++cnt_i[side][bi];
                    ++cnt_j[side][bj];
                }
                // lanes 0..15 take half 0; when half 0 is short of 16 and half 1 not empty the chunk stays contiguous: pad from half 1
                while (half[0].size() < 16 && !half[1].empty()) {
                    half[0].push_back(half[1].back());
                    half[1].pop_back();
                }
                int k = 0;
                for (uint32_t w : half[0]) pw[k++] = w;
                for (uint32_t w : half[1]) pw[k++] = w;
            }
        }
    }
    // LDS bank conflicts: a wave reads the two amplitudes of 64 consecutive entries at once (ds_read_b64 is served in two
    // 32-lane groups, bank = double slot mod 32).  The entries are a plain sum, so their order is free: they are
    // re-arranged greedily so that inside every aligned group of 32 the first indices are distinct mod 32 and so are the
    // second ones — conflict-free reads wherever the entry set allows it (a fixed order: results stay reproducible).
    if (h->opt_sparse_dealias && entries.size() > 64) {
        std::vector<std::vector<uint32_t>> by_bank(32);
        for (uint32_t e = 0; e < (uint32_t)entries.size(); ++e) by_bank[entries[e].ij & 31u].push_back(e);
        std::vector<SpEntry> arranged;
        arranged.reserve(entries.size());
        size_t left = entries.size();
        std::vector<uint32_t> order(32);
        while (left) {
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(),
                             [&](uint32_t a, uint32_t b) { return by_bank[a].size() > by_bank[b].size(); });
            uint32_t used_j = 0;
            size_t taken = 0;
            for (uint32_t bi : order) {
                std::vector<uint32_t> &lst = by_bank[bi];
                if (lst.empty()) continue;
                size_t pick = lst.size();
                for (size_t k = lst.size(); k-- > 0;) {           // newest first: cheap erase
                    const uint32_t bj = (entries[lst[k]].ij >> 12) & 31u;
                    if (!((used_j >> bj) & 1u)) {
                        pick = k;
                        break;
                    }
                }
                if (pick == lst.size()) continue;                 // every candidate collides on the second index
                used_j |= 1u << ((entries[lst[pick]].ij >> 12) & 31u);
                arranged.push_back(entries[lst[pick]]);
                lst.erase(lst.begin() + (long)pick);
                ++taken;
                --left;
            }
            if (taken == 0) {                                     // only colliding entries remain: take one anyway
                for (auto &lst : by_bank)
                    if (!lst.empty()) {
                        arranged.push_back(entries[lst.back()]);
                        lst.pop_back();
                        --left;
                        break;
                    }
            }
            // pad the group to 32 with whatever is left so that later groups stay aligned
            while (taken && taken < 32 && left) {
                bool any = false;
                for (auto &lst : by_bank)
                    if (!lst.empty() && taken < 32) {
                        arranged.push_back(entries[lst.back()]);
                        lst.pop_back();
                        --left;
                        ++taken;
                        any = true;
                    }
                if (!any) break;
            }
        }
        entries.swap(arranged);
    }
    // rows of 32 padded 64-bit words for the throughput kernel (k_sparse_vqe_rows): only when every byte offset fits 16 bits
    std::vector<uint64_t> rows;
    h->sp_nrows4 = 0;
    const int ntab_all = (int)h->srots.size();
    // ... with ONE cos/sin entry per distinct angle: table entries of one parameter with coefficients +-c (the active patterns of
    // a JW excitation) share cos and differ in the sign of sin, which moves into the word's sign bit — fewer sincos per evaluation
    // and a smaller table per evaluation in LDS (more waves per CU)
    std::vector<SmallRot> prim;
    std::vector<int> prim_of((size_t)ntab_all, -1);
    std::vector<uint8_t> prim_neg((size_t)ntab_all, 0);
    {
        std::unordered_map<uint64_t, std::vector<int>> by_param;
        for (int e = 0; e < ntab_all; ++e) {
            const SmallRot &sr = h->srots[e];
            std::vector<int> &cand = by_param[(uint64_t)(uint32_t)sr.pidx];
            for (int p : cand)
                if (std::fabs(prim[p].coeff) == std::fabs(sr.coeff) && prim[p].phi0 == 0.0 && sr.phi0 == 0.0) {
                    prim_of[e] = p;
                    prim_neg[e] = (prim[p].coeff < 0) != (sr.coeff < 0);
                    break;
                }
            if (prim_of[e] < 0) {
                prim_of[e] = (int)prim.size();
                cand.push_back((int)prim.size());
                prim.push_back(sr);
            }
        }
    }
    const int nprim = (int)prim.size();
    h->sp_nprim = 0;
    if (h->opt_sparse_rows && (size_t)(mp + 64) * 8 < 65536 && (size_t)(nprim + 1) * 16 < 65536) {
        auto pad_word = [&](int lane) {
            return (uint64_t)((uint32_t)(mp + lane) * 8u) | ((uint64_t)((uint32_t)(mp + 32 + lane) * 8u) << 16) |
                   ((uint64_t)((uint32_t)nprim * 16u) << 32);
        };
        for (const SpOp &so : ops)
            for (int c0 = 0; c0 < so.npairs; c0 += 32)
                for (int lane = 0; lane < 32; ++lane) {
                    if (c0 + lane >= so.npairs) {
                        rows.push_back(pad_word(lane));
                        continue;
                    }
                    const uint32_t pw = pairs[(size_t)so.first + c0 + lane];
                    const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu, ent = (uint32_t)so.tab0 + (pw >> 25);
                    const bool neg = ((pw >> 24) & 1u) != (uint32_t)prim_neg[ent];
                    rows.push_back((uint64_t)(ci * 8u) | ((uint64_t)(cj * 8u) << 16) | ((uint64_t)((uint32_t)prim_of[ent] * 16u) << 32) |
                                   (neg ? (1ull << 63) : 0ull));
                }
        const int nrows = (int)(rows.size() / 32);
        const int nrows4 = (nrows + 3) & ~3;
        for (int r = nrows; r < nrows4 + 4; ++r)             // padding to a multiple of four + the four rows fetched ahead
            for (int lane = 0; lane < 32; ++lane) rows.push_back(pad_word(lane));
        h->sp_nrows4 = nrows4;
        h->sp_nprim = nprim;
    }
    // ... and rows of 64 for the latency kernel (one evaluation per workgroup, the circuit on its first wave; byte offsets
    // slot * 8, spare slots mp .. mp + 127)
    std::vector<uint64_t> rows64;
    h->sp_nrows8 = 0;
    if (h->opt_sparse_rows && (size_t)(mp + 128) * 8 < 65536 && (size_t)(nprim + 1) * 16 < 65536) {
        auto pad_word = [&](int lane) {
            return (uint64_t)((uint32_t)(mp + lane) * 8u) | ((uint64_t)((uint32_t)(mp + 64 + lane) * 8u) << 16) |
                   ((uint64_t)((uint32_t)nprim * 16u) << 32);
        };
        for (const SpOp &so : ops)
            for (int c0 = 0; c0 < so.npairs; c0 += 64)
                for (int lane = 0; lane < 64; ++lane) {
                    if (c0 + lane >= so.npairs) {
                        // padded lanes rotate their (zero) spare slots by the row's FIRST cos/sin entry: rows of one entry stay
                        // uniform for the gradient kernel's wave sums
                        const uint32_t pw0 = pairs[(size_t)so.first + c0];
                        const uint32_t ent0 = (uint32_t)prim_of[(uint32_t)so.tab0 + (pw0 >> 25)];
                        rows64.push_back((pad_word(lane) & 0xffffffffull) | ((uint64_t)(ent0 * 16u) << 32));
                        continue;
                    }
                    const uint32_t pw = pairs[(size_t)so.first + c0 + lane];
                    const uint32_t ci = pw & 0xfffu, cj = (pw >> 12) & 0xfffu, ent = (uint32_t)so.tab0 + (pw >> 25);
                    const bool neg = ((pw >> 24) & 1u) != (uint32_t)prim_neg[ent];
                    rows64.push_back((uint64_t)(ci * 8u) | ((uint64_t)(cj * 8u) << 16) | ((uint64_t)((uint32_t)prim_of[ent] * 16u) << 32) |
                                     (neg ? (1ull << 63) : 0ull));
                }
        const int nrows = (int)(rows64.size() / 64);
        const int nrows8 = (nrows + 7) & ~7;
        for (int r = nrows; r < nrows8 + 8; ++r)
            for (int lane = 0; lane < 64; ++lane) rows64.push_back(pad_word(lane));
        h->sp_nrows8 = nrows8;
    }
    int rc = upload(h, h->d_sp_ops, ops.data(), ops.size() * sizeof(SpOp));
    if (!rc && h->sp_nrows4) rc = upload(h, h->d_sp_rows, rows.data(), rows.size() * sizeof(uint64_t));
    if (!rc && (h->sp_nrows4 || h->sp_nrows8)) rc = upload(h, h->d_sp_prim, prim.data(), prim.size() * sizeof(SmallRot));
    if (!rc && h->sp_nrows8) rc = upload(h, h->d_sp_rows64, rows64.data(), rows64.size() * sizeof(uint64_t));
    if (!rc) rc = upload(h, h->d_sp_pairs, pairs.data(), pairs.size() * sizeof(uint32_t));
    if (!rc) rc = upload(h, h->d_sp_entries, entries.data(), entries.size() * sizeof(SpEntry));
    if (rc) return rc;
    h->sp_m = m;
    h->sp_mp = mp;
    h->sp_hf = hf_slot;
    h->sp_nops = (int)ops.size();
    h->sp_nent = (int)entries.size();
    h->sp_npairs = (int64_t)pairs.size();
    h->sp_valid = true;
    return OVQE_OK;
}

template <int SPW, bool STAGE = false>
int launch_sparse(ovqe_handle h, const SparseArgs &A, int grid, size_t smem) {
    static bool attr_done_dev[64] = {};  // function attributes are per device
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe<SPW, STAGE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_sparse_vqe<SPW, STAGE>), dim3(grid), dim3(64), smem, h->stream, A, h->cur_theta,
                       (const SmallRot *)h->d_rots.p, (const SpOp *)h->d_sp_ops.p, (const uint32_t *)h->d_sp_pairs.p,
                       (const SpEntry *)h->d_sp_entries.p, h->cur_energies);
    HIPC(h, hipGetLastError());
    return OVQE_OK;
}

// on_device: theta / energies are device pointers (inputs already resident in HBM, results left there)
int run_sparse(ovqe_handle h, int64_t B, const double *theta, double *energies, bool on_device = false) {
    int rc = OVQE_OK;
    const double *d_theta = theta;
    double *d_energies = energies;
    const bool zero_copy = !on_device && mapped_io(h, B);
    const bool poll = zero_copy && B <= 256 && h->opt_poll_result;   // (lone evaluations and finite-difference batches: microseconds)
    if (zero_copy) {
        std::memcpy(h->h_io, theta, (size_t)B * h->K * sizeof(double));
        if (poll) poll_arm(h->h_io + (size_t)B * h->K, B);
        d_theta = h->d_io;
        d_energies = h->d_io + (size_t)B * h->K;
    } else if (!on_device) {
        rc = ensure(h, h->d_theta, (size_t)B * std::max(1, h->K) * sizeof(double));
        if (!rc) rc = ensure(h, h->d_energies, (size_t)B * sizeof(double));
        if (rc) return rc;
        HIPC(h, hipMemcpyAsync(h->d_theta.p, theta, (size_t)B * h->K * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_theta = (const double *)h->d_theta.p;
        d_energies = (double *)h->d_energies.p;
    }
    h->cur_theta = d_theta;
    h->cur_energies = d_energies;
    SparseArgs A;
    A.m = h->sp_mp;
    A.mpad = (h->sp_mp + 1) & ~1;
    A.hf = h->sp_hf;
    A.K = h->K;
    A.nops = h->sp_nops;
    A.ntab = (int)h->srots.size();
    A.nent = h->sp_nent;
    A.npairs = (int)h->sp_npairs;
    A.B = B;
    A.constant = h->ham.constant;
    A.dbg = h->opt_sparse_dbg;
    const size_t per_eval = (size_t)A.mpad * sizeof(double) + (size_t)A.ntab * sizeof(double2);
    static_assert(sizeof(double2) == 16 && sizeof(SpOp) == 16, "LDS carve-up of k_sparse_vqe assumes 16-byte records");
    int spw = h->opt_sparse_spw;
    if (spw != 1 && spw != 2 && spw != 4) spw = B >= 2048 ? 2 : 1;  // measured: 2 evaluations per wave is the sweet spot
    if (B <= 1024) spw = 1;
    while (spw > 1 && per_eval * spw > 64 * 1024) spw >>= 1;
    if (per_eval * spw > 150 * 1024) return fail(h, OVQE_ERR_INVALID, "support too large for the compacted kernel");
    const int64_t nwork = (B + spw - 1) / spw;
    const int grid = (int)std::min<int64_t>(nwork, 256 * 32);
    if (!zero_copy) HIPC(h, hipEventRecord(h->ev0, h->stream));
    // latency path: op table + pair words staged in LDS (one wave per evaluation, occupancy does not matter)
    const size_t staged = per_eval + (size_t)A.nops * sizeof(SpOp) + (size_t)A.npairs * sizeof(uint32_t);
    if (B <= 256 && h->sp_nrows8 && h->opt_sparse_rows && h->opt_sparse_wg) {
        // latency path: one evaluation per 1024-thread workgroup, at most one workgroup per CU (k_sparse_vqe_wg; measured: H2O
        // B = 1 / 141 32 / 37 us against 60 / 61 us with one wave per evaluation, B = 1024 129 against 115 us)
        SparseArgs R = A;
        R.mpad = (h->sp_mp + 128 + 1) & ~1;
        R.ntab = h->sp_nprim;
        const size_t smem = (size_t)R.mpad * sizeof(double) + (size_t)(R.ntab + 1) * sizeof(double2);
        static bool attr_wg_dev[64] = {};
        bool &attr_wg = attr_wg_dev[h->device & 63];
        if (!attr_wg) {
            HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe_wg<1024, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_wg = true;
        }
        hipLaunchKernelGGL((k_sparse_vqe_wg<1024, 10>), dim3((unsigned)std::min<int64_t>(B, 1024)), dim3(1024), smem, h->stream, R, h->cur_theta,
                           (const SmallRot *)h->d_sp_prim.p, (const uint64_t *)h->d_sp_rows64.p, h->sp_nrows8, (const SpEntry *)h->d_sp_entries.p,
                           h->cur_energies);
        HIPC(h, hipGetLastError());
    }
    else if (B <= 1024 && staged <= 96 * 1024) rc = launch_sparse<1, true>(h, A, grid, staged);
    else if (spw == 4) rc = launch_sparse<4>(h, A, grid, per_eval * 4);
    else if (spw == 2 && h->sp_nrows4 && h->opt_sparse_rows) {
        SparseArgs R = A;
        R.mpad = (h->sp_mp + 64 + 1) & ~1;   // + the padded lanes' spare slots
        R.ntab = h->sp_nprim;                // one entry per distinct angle
        const size_t per_eval_r = (size_t)R.mpad * sizeof(double) + (size_t)(R.ntab + 1) * sizeof(double2);
        static bool attr_rows_dev[64] = {};
        bool &attr_rows = attr_rows_dev[h->device & 63];
        if (!attr_rows) {
            HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe_rows<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_rows = true;
        }
#define OVQE_ROWS(DBG_)                                                                                                                       \
    hipLaunchKernelGGL((k_sparse_vqe_rows<2, DBG_>), dim3(grid), dim3(64), per_eval_r * 2, h->stream, R, h->cur_theta, (const SmallRot *)h->d_sp_prim.p, \
                       (const uint64_t *)h->d_sp_rows.p, h->sp_nrows4, (const SpEntry *)h->d_sp_entries.p, h->cur_energies)
        switch (h->opt_sparse_dbg) {   // (measurement variants carry their own LDS attribute: set on the fly)
        case 1: HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe_rows<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); OVQE_ROWS(1); break;
        case 2: HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe_rows<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); OVQE_ROWS(2); break;
        case 3: HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_vqe_rows<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); OVQE_ROWS(3); break;
        default: OVQE_ROWS(0);
        }
#undef OVQE_ROWS
        HIPC(h, hipGetLastError());
    }
    else if (spw == 2) rc = launch_sparse<2>(h, A, grid, per_eval * 2);
    else rc = launch_sparse<1>(h, A, grid, per_eval);
    if (rc) return rc;
    if (zero_copy) {
        if (!poll || !poll_mapped_slots(h->h_io + (size_t)B * h->K, B)) HIPC(h, hipStreamSynchronize(h->stream));
        std::memcpy(energies, h->h_io + (size_t)B * h->K, (size_t)B * sizeof(double));
        h->last_batch_ms = 0.f;
        return OVQE_OK;
    }
    HIPC(h, hipEventRecord(h->ev1, h->stream));
    if (!on_device)
        HIPC(h, hipMemcpyAsync(energies, h->cur_energies, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipEventElapsedTime(&h->last_batch_ms, h->ev0, h->ev1));
    return OVQE_OK;
}

// E and all K derivatives of one parameter vector on the compact support, one launch (k_sparse_grad).  *done = false: the
// program has no compact support or its tables do not fit one workgroup's LDS (the caller takes the other paths).
int run_sparse_gradient(ovqe_handle h, const double *theta, double *energy, double *grad, bool *done) {
    *done = false;
    if (!(h->opt_force_path == 0 || h->opt_force_path == 3) || !h->opt_sparse_grad || h->n_local > 16) return OVQE_OK;
    int rc = OVQE_OK;
    if (!h->sp_tried) {
        rc = build_sparse_program(h);
        if (rc) return rc;
    }
    if (!h->sp_valid) return OVQE_OK;
    SparseArgs A;
    A.m = h->sp_mp;
    A.mpad = (h->sp_mp + 1) & ~1;
    A.hf = h->sp_hf;
    A.K = h->K;
    A.nops = h->sp_nops;
    A.ntab = (int)h->srots.size();
    A.nent = h->sp_nent;
    A.npairs = (int)h->sp_npairs;
    A.B = 1;
    A.constant = h->ham.constant;
    const size_t base = 2 * (size_t)A.mpad * sizeof(double) + (size_t)A.ntab * (sizeof(double2) + sizeof(double)) +
                        (size_t)((A.K + 1) & ~1) * sizeof(double);
    const size_t staged = base + (size_t)A.nops * sizeof(SpOp) + (size_t)A.npairs * sizeof(uint32_t);
    if (base > 150 * 1024) return OVQE_OK;
    const bool stage = staged <= 150 * 1024;
    static bool attr_done_dev[64] = {};
    bool &attr_done = attr_done_dev[h->device & 63];
    if (!attr_done) {
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_grad<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_grad<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const bool zero_copy = mapped_io(h, 2);   // theta [K] | energy | gradient [K] in the pinned, device-mapped buffer
    const double *d_theta;
    double *d_e, *d_g;
    if (zero_copy) {
        std::memcpy(h->h_io, theta, (size_t)h->K * sizeof(double));
        d_theta = h->d_io;
        d_e = h->d_io + h->K;
        d_g = h->d_io + h->K + 1;
    } else {
        rc = ensure(h, h->d_theta, (size_t)std::max(1, h->K) * sizeof(double));
        if (!rc) rc = ensure(h, h->d_energies, (size_t)(h->K + 1) * sizeof(double));
        if (rc) return rc;
        HIPC(h, hipMemcpyAsync(h->d_theta.p, theta, (size_t)h->K * sizeof(double), hipMemcpyHostToDevice, h->stream));
        d_theta = (const double *)h->d_theta.p;
        d_e = (double *)h->d_energies.p;
        d_g = d_e + 1;
    }
    if (h->sp_nrows8 && h->opt_sparse_rows && h->opt_sparse_wg) {
        SparseArgs R = A;
        R.mpad = (h->sp_mp + 128 + 1) & ~1;
        R.ntab = h->sp_nprim;
        const size_t smem = 2 * (size_t)R.mpad * sizeof(double) + (size_t)(R.ntab + 1) * sizeof(double2) + (size_t)((R.ntab + 2) & ~1) * sizeof(double) +
                            (size_t)((R.K + 1) & ~1) * sizeof(double);
        static bool attr_gwg_dev[64] = {};
        bool &attr_gwg = attr_gwg_dev[h->device & 63];
        if (!attr_gwg) {
            HIPC(h, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sparse_grad_wg<1024, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_gwg = true;
        }
        if (smem <= 150 * 1024) {
            hipLaunchKernelGGL((k_sparse_grad_wg<1024, 10>), dim3(1), dim3(1024), smem, h->stream, R, d_theta, (const SmallRot *)h->d_sp_prim.p,
                               (const uint64_t *)h->d_sp_rows64.p, h->sp_nrows8, (const SpEntry *)h->d_sp_entries.p, d_e, d_g);
        } else if (stage) {
            hipLaunchKernelGGL((k_sparse_grad<true>), dim3(1), dim3(64), staged, h->stream, A, d_theta, (const SmallRot *)h->d_rots.p,
                               (const SpOp *)h->d_sp_ops.p, (const uint32_t *)h->d_sp_pairs.p, (const SpEntry *)h->d_sp_entries.p, d_e, d_g);
        } else {
            hipLaunchKernelGGL((k_sparse_grad<false>), dim3(1), dim3(64), base, h->stream, A, d_theta, (const SmallRot *)h->d_rots.p,
                               (const SpOp *)h->d_sp_ops.p, (const uint32_t *)h->d_sp_pairs.p, (const SpEntry *)h->d_sp_entries.p, d_e, d_g);
        }
    } else if (stage)
        hipLaunchKernelGGL((k_sparse_grad<true>), dim3(1), dim3(64), staged, h->stream, A, d_theta, (const SmallRot *)h->d_rots.p,
                           (const SpOp *)h->d_sp_ops.p, (const uint32_t *)h->d_sp_pairs.p, (const SpEntry *)h->d_sp_entries.p, d_e, d_g);
    else
        hipLaunchKernelGGL((k_sparse_grad<false>), dim3(1), dim3(64), base, h->stream, A, d_theta, (const SmallRot *)h->d_rots.p,
                           (const SpOp *)h->d_sp_ops.p, (const uint32_t *)h->d_sp_pairs.p, (const SpEntry *)h->d_sp_entries.p, d_e, d_g);
    HIPC(h, hipGetLastError());
    if (zero_copy) {
        HIPC(h, hipStreamSynchronize(h->stream));
        *energy = h->h_io[h->K];
        std::memcpy(grad, h->h_io + h->K + 1, (size_t)h->K * sizeof(double));
    } else {
        HIPC(h, hipMemcpyAsync(energy, d_e, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipMemcpyAsync(grad, d_g, (size_t)h->K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
    }
    *done = true;
    return OVQE_OK;
}

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
                                  &h->d_pg_out, &h->d_pg_part, &h->d_nz_cnt, &h->d_nz_start, &h->d_nz_idx, &h->d_nz_val, &h->d_nz_bitmap, &h->d_tile_smasks, &h->d_tile_lists, &h->d_tile_counts, &h->cc.d_sup, &h->cc.d_psic, &h->cc.d_loc, &h->cc.d_cid, &h->cc.d_off, &h->cc.d_sweeps};
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
    else if (k == "sector_row_banks") h->opt_sector_row_banks = (int)value;
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
    HIPC(h, hipStreamSynchronize(h->stream));
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

// ---- unit operations ----------------------------------------------------------------------------
int ovqe_apply_pauli_rotations(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *phi) try {
    OVQE_ENTER(h);
    if (!h || R < 0 || (R && (!x || !z || !phi))) return OVQE_ERR_INVALID;
    if (R == 0) return OVQE_OK;
    const uint64_t lmask = local_mask(h);
    const int ntot = h->n_local + h->n_global;
    const uint64_t allmask = ntot >= 64 ? ~0ull : ((1ull << ntot) - 1ull);
    for (int64_t r = 0; r < R; ++r) {
        if ((x[r] | z[r]) & ~allmask) return fail(h, OVQE_ERR_INVALID, "Pauli mask has bits beyond the register");
        if (x[r] & ~lmask)
            return fail(h, OVQE_ERR_INVALID,
                        "x mask touches global (rank) bits: exchange shards first (openvqe_amd/distributed.py)");
    }
    int rc = ensure_rp(h, (size_t)R);
    if (rc) return rc;
    for (int64_t r = 0; r < R; ++r) h->h_rp[r] = make_rot(x[r], z[r], phi[r]);
    HIPC(h, hipMemcpyAsync(h->d_rp.p, h->h_rp, (size_t)R * sizeof(RotParam), hipMemcpyHostToDevice, h->stream));
    // same-x runs, then LDS-tiled multi-run sweeps where consecutive runs fit a tile (sharded registers: the local
    // sweeps between two exchanges arrive here as one list)
    std::vector<SmallOp> ops;
    std::vector<SmallRot> rots((size_t)R);
    for (int64_t r0 = 0; r0 < R;) {
        int64_t r1 = r0 + 1;
        while (r1 < R && x[r1] == x[r0]) ++r1;
        SmallOp op = {};
        op.x = x[r0];
        op.kind = x[r0] ? OP_PAIR : OP_DIAG;
        op.first = (int32_t)r0;
        op.count = (int32_t)(r1 - r0);
        op.pivot = x[r0] ? 63 - __builtin_clzll(x[r0]) : 0;
        ops.push_back(op);
        for (int64_t r = r0; r < r1; ++r) rots[(size_t)r].z = z[r];
        r0 = r1;
    }
    // option "real_state": the buffer holds doubles — every string must keep a real state real (odd number of Y, x != 0): the
    // real-amplitude tile sweeps (k_tile_sweep<REAL>: 2^13 amplitudes per 64-KB tile, one more mixing bit per sweep) and pair sweeps
    const bool real = h->opt_real_state != 0;
    if (real) {
        if (h->n_local < 2) return fail(h, OVQE_ERR_INVALID, "real_state needs at least two local qubits");
        for (int64_t r = 0; r < R; ++r)
            if (x[r] == 0 || !(__builtin_popcountll(x[r] & z[r]) & 1))
                return fail(h, OVQE_ERR_INVALID, "real_state: a rotation with an even number of Y (or a diagonal one) makes the amplitudes "
                                                 "complex — clear the option and widen the buffer first");
    }
    TilePlan &tp = h->tp_adhoc;
    if (ops.size() >= 2 && R < (1ll << 30)) {
        rc = build_tile_plan(h, ops, rots, std::vector<uint64_t>(ops.size(), 0), tp, real);
        if (rc) return rc;
    } else {
        tp.plan.assign(ops.size(), 0);
        for (size_t i = 0; i < ops.size(); ++i) tp.plan[i] = -1 - (int32_t)i;
    }
    for (const int32_t step : tp.plan) {
        if (step >= 0) {
            rc = launch_tile_segment(h, tp, tp.tsegs[step], real);
        } else {
            const SmallOp &op = ops[-1 - step];
            if (real) {
                hipLaunchKernelGGL(k_rot_pairs_real, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, (double *)h->state, h->namps >> 1,
                                   op.pivot, op.x, h->base, (const RotParam *)h->d_rp.p + op.first, op.count);
                HIPC(h, hipGetLastError());
                rc = OVQE_OK;
            } else {
                rc = launch_rot_run(h, op.x, (const RotParam *)h->d_rp.p + op.first, op.count);
            }
        }
        if (rc) return rc;
    }
    HIPC(h, hipStreamSynchronize(h->stream));
    h->last_passes = (int64_t)tp.plan.size();               // every step reads and writes the shard once
    h->last_pass_bytes = (int64_t)((real ? 16.0 : 32.0) * (double)h->namps * (double)tp.plan.size());
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_apply_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi) try {
    OVQE_ENTER(h);
    return ovqe_apply_pauli_rotations(h, 1, &x, &z, &phi);
} OVQE_CATCH(h)

int ovqe_apply_gate(ovqe_handle h, int opcode, int b0, int b1, double angle) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_ERR_INVALID;
    if (b0 < 0 || b0 >= h->n_local) return fail(h, OVQE_ERR_INVALID, "gate bit out of (local) range");
    const uint64_t bit = 1ull << b0;
    int rc;
    switch (opcode) {
    case OVQE_GATE_X: rc = launch_gate(h, 0, b0, 0); break;
    case OVQE_GATE_H: rc = launch_gate(h, 1, b0, 0); break;
    case OVQE_GATE_RX: return ovqe_apply_pauli_rotation(h, bit, 0, 0.5 * angle);
    case OVQE_GATE_RY: return ovqe_apply_pauli_rotation(h, bit, bit, 0.5 * angle);
    case OVQE_GATE_RZ: return ovqe_apply_pauli_rotation(h, 0, bit, 0.5 * angle);
    case OVQE_GATE_CNOT:
        if (b1 < 0 || b1 >= h->n_local || b1 == b0) return fail(h, OVQE_ERR_INVALID, "CNOT target bit invalid");
        rc = launch_gate(h, 2, b0, b1);
        break;
    default: return fail(h, OVQE_ERR_INVALID, "unknown gate opcode");
    }
    if (rc) return rc;
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_bilinear(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t T, const uint64_t *x,
                  const uint64_t *z, const double *coeff_re, const double *coeff_im, double *out_re_im) try {
    OVQE_ENTER(h);
    if (!h || T < 0 || !out_re_im || (T && (!x || !z || !coeff_re))) return OVQE_ERR_INVALID;
    bool real_coeffs = true;  // real coefficients: every term is Hermitian -> pair-trick kernels when bra == ket
    if (coeff_im)
        for (int64_t t = 0; t < T; ++t) real_coeffs = real_coeffs && coeff_im[t] == 0.0;
    if (!bra_dev && !ket_dev && real_coeffs && T > 0) {
        // <state|H|state> of a Hermitian sum: kept (with its tile cover) until a different sum arrives — a sharded
        // register evaluates the same local term list once per energy
        HamDev &H = h->ham_adhoc;
        const bool same = H.set && (int64_t)h->adhoc_x.size() == T &&
                          std::equal(x, x + T, h->adhoc_x.begin()) && std::equal(z, z + T, h->adhoc_z.begin()) &&
                          std::equal(coeff_re, coeff_re + T, h->adhoc_c.begin());
        if (!same) {
            H.set = false;
            int rc = build_groups(h, T, x, z, coeff_re, nullptr, false, H.groups, H.terms);
            if (rc) return rc;
            rc = upload(h, H.d_groups, H.groups.data(), H.groups.size() * sizeof(HGroup));
            if (!rc) rc = upload(h, H.d_terms, H.terms.data(), H.terms.size() * sizeof(HTerm));
            if (rc) return rc;
            h->adhoc_x.assign(x, x + T);
            h->adhoc_z.assign(z, z + T);
            h->adhoc_c.assign(coeff_re, coeff_re + T);
            H.tile_bits = -1;
            H.set = true;
        }
        double2 res = make_double2(0.0, 0.0);
        bool tiled = false;
        int rc = run_expectation_tiled(h, H, &res, &tiled);
        if (!rc && !tiled)
            rc = run_bilinear(h, h->state, h->state, H.groups, (const HGroup *)H.d_groups.p, (const HTerm *)H.d_terms.p,
                              &res, true);
        out_re_im[0] = res.x;
        out_re_im[1] = res.y;
        return rc;
    }
    std::vector<HGroup> groups;
    std::vector<HTerm> terms;
    int rc = build_groups(h, T, x, z, coeff_re, coeff_im, /*allow_global_x=*/ket_dev != nullptr, groups, terms);
    if (rc) return rc;
    // (the handle's pool buffers carry the term list: a sharded <H> makes tens of thousands of these calls, one per partner chunk)
    rc = upload(h, h->d_pg_xs, groups.data(), std::max<size_t>(groups.size(), 1) * sizeof(HGroup));
    if (!rc) rc = upload(h, h->d_pg_terms, terms.data(), std::max<size_t>(terms.size(), 1) * sizeof(HTerm));
    h->pg_valid = false;  // the pool buffers were borrowed
    double2 res = make_double2(0.0, 0.0);
    if (!rc)
        rc = run_bilinear(h, bra_dev ? (const amp_t *)bra_dev : h->state, ket_dev ? (const amp_t *)ket_dev : h->state,
                          groups, (const HGroup *)h->d_pg_xs.p, (const HTerm *)h->d_pg_terms.p, &res, false);
    out_re_im[0] = res.x;
    out_re_im[1] = res.y;
    return rc;
} OVQE_CATCH(h)

int ovqe_apply_pauli_sum(ovqe_handle h, const void *ket_dev, void *out_dev, int64_t T, const uint64_t *x, const uint64_t *z,
                         const double *coeff_re, const double *coeff_im, int accumulate) try {
    OVQE_ENTER(h);
    if (!h || !out_dev || T < 0 || (T && (!x || !z || !coeff_re))) return OVQE_ERR_INVALID;
    const amp_t *ket = ket_dev ? (const amp_t *)ket_dev : h->state;
    if ((const void *)ket == out_dev) return fail(h, OVQE_ERR_INVALID, "ovqe_apply_pauli_sum: out must differ from the ket");
    if (T == 0) {   // the empty sum: out = 0, or out unchanged when accumulating
        if (!accumulate) HIPC(h, hipMemsetAsync(out_dev, 0, (size_t)h->namps * sizeof(amp_t), h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        return OVQE_OK;
    }
    std::vector<HGroup> groups;
    std::vector<HTerm> terms;
    int rc = build_groups(h, T, x, z, coeff_re, coeff_im, /*allow_global_x=*/ket_dev != nullptr, groups, terms);
    if (rc) return rc;
    rc = upload(h, h->d_pg_xs, groups.data(), std::max<size_t>(groups.size(), 1) * sizeof(HGroup));
    if (!rc) rc = upload(h, h->d_pg_terms, terms.data(), std::max<size_t>(terms.size(), 1) * sizeof(HTerm));
    if (rc) return rc;
    h->pg_valid = false;  // the pool buffers were borrowed
    hipLaunchKernelGGL(k_apply_terms, dim3(reduce_blocks(h->namps)), dim3(256), 0, h->stream, (amp_t *)out_dev, ket, h->namps,
                       (const HGroup *)h->d_pg_xs.p, (int)groups.size(), (const HTerm *)h->d_pg_terms.p, accumulate);
    HIPC(h, hipGetLastError());
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_bilinear_batch(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t n_ops, const int64_t *offsets,
                        const uint64_t *x, const uint64_t *z, const double *coeff_re, const double *coeff_im,
                        double *out_re_im) try {
    OVQE_ENTER(h);
    if (!h || n_ops < 0 || !offsets || (n_ops && !out_re_im)) return OVQE_ERR_INVALID;
    if (n_ops == 0) return OVQE_OK;
    const int64_t T = offsets[n_ops];
    if (T < 0 || (T && (!x || !z || !coeff_re))) return OVQE_ERR_INVALID;
    if (T == 0) {   // operators without terms: every bilinear form is 0
        std::fill(out_re_im, out_re_im + 2 * n_ops, 0.0);
        return OVQE_OK;
    }
    const uint64_t lmask = local_mask(h);
    std::vector<HTerm> terms(T);
    std::vector<uint64_t> xs(T);
    uint64_t xg = 0;
    for (int64_t t = 0; t < T; ++t) {
        if (t == 0) xg = x[t] & ~lmask;
        if ((x[t] & ~lmask) != xg) return fail(h, OVQE_ERR_INVALID, "ovqe_bilinear_batch: one global x part per call");
        const int ny = __builtin_popcountll(x[t] & z[t]) & 3;
        const double a = coeff_re[t], b = coeff_im ? coeff_im[t] : 0.0;
        HTerm ht;
        ht.z = z[t];
        switch (ny) {
        case 0: ht.cr = a; ht.ci = b; break;
        case 1: ht.cr = -b; ht.ci = a; break;
        case 2: ht.cr = -a; ht.ci = -b; break;
        default: ht.cr = b; ht.ci = -a; break;
        }
        terms[t] = ht;
        xs[t] = x[t] & lmask;
    }
    if (xg && !ket_dev) return fail(h, OVQE_ERR_INVALID, "x mask touches global (rank) bits: pass the partner's shard as ket");
    for (int64_t k = 0; k < n_ops; ++k)
        if (offsets[k] > offsets[k + 1] || offsets[k] < 0) return fail(h, OVQE_ERR_INVALID, "offsets not monotone");
    h->pg_valid = false;
    int rc = upload(h, h->d_pg_off, offsets, (n_ops + 1) * sizeof(int64_t));
    if (!rc) rc = upload(h, h->d_pg_xs, xs.data(), std::max<int64_t>(T, 1) * sizeof(uint64_t));
    if (!rc) rc = upload(h, h->d_pg_terms, terms.data(), std::max<int64_t>(T, 1) * sizeof(HTerm));
    const int nchunks = h->n_local <= 22 ? 1 : (int)(h->namps >> 16);
    const int64_t ops_per_launch = 32768;
    if (!rc) rc = ensure(h, h->d_pg_out, n_ops * sizeof(double2));
    if (!rc && nchunks > 1)
        rc = ensure(h, h->d_pg_part, (size_t)std::min<int64_t>(n_ops, ops_per_launch) * nchunks * sizeof(double2));
    if (rc) return rc;
    const amp_t *bra = bra_dev ? (const amp_t *)bra_dev : h->state, *ket = ket_dev ? (const amp_t *)ket_dev : h->state;
    const uint64_t ket_base = (h->base ^ xg) & ~lmask;
    for (int64_t op0 = 0; op0 < n_ops; op0 += ops_per_launch) {
        const int64_t cnt = std::min<int64_t>(ops_per_launch, n_ops - op0);
        double2 *out = (double2 *)h->d_pg_out.p + op0;
        double2 *part = nchunks > 1 ? (double2 *)h->d_pg_part.p : out;
        hipLaunchKernelGGL(k_pool_grad, dim3((unsigned)nchunks, (unsigned)cnt), dim3(256), 0, h->stream, bra, ket, h->namps,
                           ket_base, (const int64_t *)h->d_pg_off.p, (const uint64_t *)h->d_pg_xs.p,
                           (const HTerm *)h->d_pg_terms.p, op0, part);
        if (nchunks > 1)
            hipLaunchKernelGGL(k_reduce_rows2, dim3((unsigned)cnt), dim3(256), 0, h->stream, (const double2 *)part, nchunks, out);
    }
    HIPC(h, hipGetLastError());
    HIPC(h, hipMemcpyAsync(out_re_im, h->d_pg_out.p, n_ops * sizeof(double2), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_expectation(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                     double constant, double *out) try {
    OVQE_ENTER(h);
    if (!out) return OVQE_ERR_INVALID;
    double res[2] = {0.0, 0.0};
    int rc = ovqe_bilinear(h, nullptr, nullptr, T, x, z, coeff, nullptr, res);
    if (rc) return rc;
    *out = res[0] + constant;
    return OVQE_OK;
} OVQE_CATCH(h)

// ---- compiled evaluation ------------------------------------------------------------------------
int ovqe_set_hamiltonian(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                         double constant) try {
    OVQE_ENTER(h);
    if (!h || T < 0 || (T && (!x || !z || !coeff))) return OVQE_ERR_INVALID;
    int rc = install_hamdev(h, h->ham, T, x, z, coeff, constant);
    if (rc) return rc;
    free_sector(h->scr);   // the screen engine's tables belong to the Hamiltonian that was replaced
    h->scr_failed_version = -1;
    h->user_x.assign(x, x + T);
    h->user_z.assign(z, z + T);
    h->user_c.assign(coeff, coeff + T);
    h->user_const = constant;
    h->exp_lbits = -1;
    h->sp_tried = false;
    return install_conjugated_hamiltonian(h);
} OVQE_CATCH(h)

// energy-type entry points evaluate <phi|C^+ H C|phi> when the program's Clifford frame is open
struct FrameHamGuard {
    ovqe_handle h;
    bool on;
    explicit FrameHamGuard(ovqe_handle hh) : h(hh), on(hh && hh->frame_open && hh->ham_conj.set) {
        if (on) std::swap(h->ham, h->ham_conj);
    }
    ~FrameHamGuard() {
        if (on) std::swap(h->ham, h->ham_conj);
    }
};

int ovqe_set_program(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *coeff,
                     const double *phi0, const int32_t *pidx, int32_t K, uint64_t hf_index) try {
    OVQE_ENTER(h);
    if (h) h->prog_from_gates = false;
    if (!h || R < 0 || K < 0 || (R && (!x || !z || !coeff || !pidx))) return OVQE_ERR_INVALID;
    const int ntot = h->n_local + h->n_global;
    const uint64_t allmask = ntot >= 64 ? ~0ull : ((1ull << ntot) - 1ull);
    if (hf_index & ~allmask) return fail(h, OVQE_ERR_INVALID, "hf_index out of range");
    for (int64_t r = 0; r < R; ++r) {
        if ((x[r] | z[r]) & ~allmask) return fail(h, OVQE_ERR_INVALID, "Pauli mask has bits beyond the register");
        if (x[r] & ~local_mask(h)) return fail(h, OVQE_ERR_INVALID, "x mask touches global (rank) bits");
        if (pidx[r] >= K) return fail(h, OVQE_ERR_INVALID, "parameter index >= K");
    }
    h->prog_set = false;
    h->ops.clear();
    h->rots.clear();
    h->init_amp = make_double2(1.0, 0.0);
    h->K = K;
    h->hf = hf_index;
    h->frame_open = false;
    for (int64_t r = 0; r < R; ++r) push_rotation(h, x[r], z[r], coeff[r], phi0 ? phi0[r] : 0.0, pidx[r]);
    return finish_program(h);
} OVQE_CATCH(h)

int ovqe_set_gate_program(ovqe_handle h, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1,
                          const double *ascale, const double *aconst, const int32_t *pidx, int32_t K,
                          uint64_t hf_index) try {
    OVQE_ENTER(h);
    if (!h || G < 0 || K < 0 || (G && (!opcode || !b0 || !b1 || !ascale || !aconst || !pidx))) return OVQE_ERR_INVALID;
    if (h->n_global) return fail(h, OVQE_ERR_INVALID, "gate programs are single-device");
    if (hf_index >> h->n_local) return fail(h, OVQE_ERR_INVALID, "hf_index out of range");
    for (int64_t g = 0; g < G; ++g) {
        if (b0[g] < 0 || b0[g] >= h->n_local) return fail(h, OVQE_ERR_INVALID, "gate bit out of range");
        if (opcode[g] == OVQE_GATE_CNOT && (b1[g] < 0 || b1[g] >= h->n_local || b1[g] == b0[g]))
            return fail(h, OVQE_ERR_INVALID, "CNOT target bit invalid");
        if (opcode[g] < 0 || opcode[g] > OVQE_GATE_CNOT) return fail(h, OVQE_ERR_INVALID, "unknown gate opcode");
        if (pidx[g] >= K) return fail(h, OVQE_ERR_INVALID, "parameter index >= K");
    }
    h->K = K;
    h->hf = hf_index;
    h->frame_open = false;
    h->prog_from_gates = false;
    if (h->opt_clifford_frame) {
        bool done = false;
        int rc = compile_gate_program_frame(h, G, opcode, b0, b1, ascale, aconst, pidx, &done);
        if (rc) return rc;
        if (done) {
            h->prog_from_gates = true;
            return OVQE_OK;
        }
    }
    return compile_gate_program_literal(h, G, opcode, b0, b1, ascale, aconst, pidx);
} OVQE_CATCH(h)

// the Clifford part of an open frame, literally, on the state in the buffer: |psi> = C |phi>
static int apply_tail_gates(ovqe_handle h) {
    int rc = OVQE_OK;
    for (size_t g = 0; g + 3 < h->tail_gates.size() && !rc; g += 4) {
        const int op = h->tail_gates[g], t = h->tail_gates[g + 1], c = h->tail_gates[g + 2];
        const double a = h->tail_gates[g + 3] > 0 ? M_PI_2 : -M_PI_2;
        rc = ovqe_apply_gate(h, op, t, c, a);
    }
    return rc;
}

int ovqe_prepare_state(ovqe_handle h, const double *theta, int32_t K) try {
    OVQE_ENTER(h);
    if (!h) return OVQE_ERR_INVALID;
    int rc = check_theta(h, theta, K);
    if (rc) return rc;
    rc = run_program_streaming(h, theta);
    if (rc || !h->frame_open) return rc;
    return apply_tail_gates(h);
} OVQE_CATCH(h)

// batched sector evaluations need the tables of the second sweep kernel (64-bit pair words) and the materialised <H>
static bool sector_batch_ready(ovqe_handle h) {
    const SectorEngine &E = h->sec;
    const bool real = h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true) && h->ham.groups.size() >= 3;
    return real && h->opt_sector && h->opt_sector_batch && E.valid && E.h_tables && E.pad_elems && !E.segs.empty() && E.segs[0].d_wide.p &&
           E.prog_version == h->prog_version && E.ham_version == h->ham.version &&
           sector_h_smem(E, SEC_BATCH_NB) <= 156 * 1024;
}

int ovqe_energy_batch(ovqe_handle h, int64_t B, const double *theta, int32_t K, double *energies) try {
    OVQE_ENTER(h);
    if (!h || B < 0 || (B && !energies)) return OVQE_ERR_INVALID;
    FrameHamGuard frame_guard(h);
    int rc = check_theta(h, theta, K);
    if (rc) return rc;
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (B == 0) return OVQE_OK;
    if (h->opt_force_path == 0 || h->opt_force_path == 3) {
        // one wave owns an evaluation on the compact support: unbeatable for batches, but a lone evaluation of a
        // register beyond the LDS kernels is quicker on the whole chip (streaming + tiled sweeps) — and then the compact
        // program is not even built (an ADAPT ansatz at 24 qubits with a few thousand determinants: 20-100 ms of host work
        // per macro-iteration for tables no single evaluation would use)
        const bool wanted = h->opt_force_path == 3 || h->n_local <= 14 || B >= 16;
        if (wanted && !h->sp_tried) {
            rc = build_sparse_program(h);
            if (rc) return rc;
        }
        if (wanted && h->sp_valid) return run_sparse(h, B, theta, energies);
        if (h->opt_force_path == 3) return fail(h, OVQE_ERR_STATE, "program has no compact support (sparse path forced)");
    }
    if (use_small_path(h, B)) return run_small(h, B, theta, energies);
    // a lone evaluation is not timed with events (ovqe_last_batch_ms reads 0, as on the fused kernels' zero-copy path): every path below
    // ends with the host reading its result behind a synchronisation, and two event records + a second synchronisation + the
    // elapsed-time query are 6-8 us of the ~95 us of an ADAPT-sized evaluation
    const bool timed = B > 1;
    if (timed) HIPC(h, hipEventRecord(h->ev0, h->stream));
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {   // "sector_debug" bit 2 (value 4): stages of an evaluation that took more than 3 ms
        if (!(h->opt_sector_debug & 4)) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - lap_t).count();
        if (ms > 3.0) fprintf(stderr, "ovqe: evaluation, %s: %.2f ms\n", what, ms);
        lap_t = now;
    };
    int64_t b_first = 0;
    if (B >= 4 && h->opt_sector_batch && h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true) &&
        h->ham.groups.size() >= 3) {
        rc = sector_prepare(h, true);   // a batch is worth the tables at once (a lone evaluation builds them at its second call)
        if (rc) return rc;
    }
    if (B >= 2 && sector_batch_ready(h)) {   // the whole batch in one pass of the sector tables
        bool ok = false;
        rc = run_sector_energy_batch(h, B, theta, false, energies, &ok);
        if (rc) return rc;
        if (ok) b_first = B;
    }
    for (int64_t b = b_first; b < B; ++b) {
        // programs that keep the amplitudes real stream 8 bytes per amplitude (state left as 2^n doubles)
        const bool real = h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true) &&
                          h->ham.groups.size() >= 3;
        if (real && h->ham_real.version != h->ham.version) {
            HamDev &R = h->ham_real;
            R.groups = h->ham.groups;
            R.terms = h->ham.terms;
            rc = upload(h, R.d_terms, R.terms.data(), R.terms.size() * sizeof(HTerm));
            if (rc) return rc;
            R.tile_bits = -1;
            R.version = h->ham.version;
        }
        if (real) {
            lap("before the sector tables");
            rc = sector_prepare(h);
            if (rc) return rc;
            lap("sector_prepare");
            SectorEngine &E = h->sec;
            if (E.valid && E.h_tables) {
                double2 res;
                bool ok = false;
                rc = run_sector_energy(h, theta + b * (int64_t)K, &res, &ok);
                if (rc) return rc;
                lap("run_sector_energy");
                if (ok) {
                    energies[b] = res.x + h->ham.constant;
                    continue;
                }
                sector_orphaned(h);   // a structurally-zero amplitude was not: these tables do not describe this program
            }
        }
        bool use_cc = false;
        if (real && h->opt_compact && h->n_local >= 18) {
            // compact cover: built at the second evaluation of a (program, Hamiltonian) pair — one-shot callers never pay
            HamDev &R = h->ham_real;
            CompactCover &C = h->cc;
            if (R.tile_bits != tile_bits(h, true) || R.tile_low != ham_tile_low(h, true) || !R.tile_real) {
                rc = build_ham_tiles(h, R, true);
                if (rc) return rc;
            }
            if (C.prog_version != h->prog_version || C.ham_version != R.version || C.cover_id != R.cover_id) {
                C.valid = C.disabled = false;
                C.prog_version = h->prog_version;
                C.ham_version = R.version;
                C.cover_id = R.cover_id;
                C.seen = 0;
            }
            if (!C.valid && !C.disabled && ++C.seen >= 2) {
                rc = build_compact_cover(h, R);
                if (rc) return rc;
            }
            use_cc = C.valid;
        }
        if (use_cc && real && h->opt_sector && h->sec.valid && !h->sec.h_tables && h->sec.K == h->cc.K) {
            // circuit on the sector tables, <H> by the compact cover on the canonical compact state
            double2 res;
            bool ok = false;
            rc = run_sector_state(h, theta + b * (int64_t)K, (double *)h->cc.d_psic.p, &ok);
            if (rc) return rc;
            if (ok) {
                rc = run_expectation_compact(h, h->ham_real, &res, &ok, true);
                if (rc) return rc;
            }
            if (ok) {
                energies[b] = res.x + h->ham.constant;
                continue;
            }
            sector_orphaned(h);
        }
        lap("before the dense run");
        rc = run_program_streaming(h, theta + b * (int64_t)K, real);
        if (rc) return rc;
        lap("dense run");
        double2 res;
        bool tiled = false;
        if (use_cc) {
            rc = run_expectation_compact(h, h->ham_real, &res, &tiled);
            if (rc) return rc;
            if (!tiled) {  // the state left the recorded support: this program does not qualify
                h->cc.valid = false;
                h->cc.disabled = true;
            }
        }
        if (!tiled) rc = run_expectation_tiled(h, real ? h->ham_real : h->ham, &res, &tiled, real);
        if (rc) return rc;
        if (real && !tiled) return fail(h, OVQE_ERR_INVALID, "internal: real-amplitude path without a tile cover");
        if (!tiled)
            rc = run_bilinear(h, h->state, h->state, h->ham.groups, (const HGroup *)h->ham.d_groups.p,
                              (const HTerm *)h->ham.d_terms.p, &res, true);
        if (rc) return rc;
        lap("dense <H>");
        energies[b] = res.x + h->ham.constant;
    }
    if (!timed) {
        h->last_batch_ms = 0.f;
        return OVQE_OK;
    }
    HIPC(h, hipEventRecord(h->ev1, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    HIPC(h, hipEventElapsedTime(&h->last_batch_ms, h->ev0, h->ev1));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_energy_batch_device(ovqe_handle h, int64_t B, const void *theta_dev, int32_t K, void *energies_dev) try {
    OVQE_ENTER(h);
    if (!h || B < 0 || (B && (!theta_dev || !energies_dev))) return OVQE_ERR_INVALID;
    FrameHamGuard frame_guard(h);
    if (!h->prog_set) return fail(h, OVQE_ERR_STATE, "no program set (ovqe_set_program / ovqe_set_gate_program)");
    if (K != h->K || K <= 0) return fail(h, OVQE_ERR_INVALID, "K does not match the program's parameter count");
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (B == 0) return OVQE_OK;
    int rc;
    if (h->opt_force_path == 0 || h->opt_force_path == 3) {
        if (!h->sp_tried) {
            rc = build_sparse_program(h);
            if (rc) return rc;
        }
        if (h->sp_valid) return run_sparse(h, B, (const double *)theta_dev, (double *)energies_dev, true);
        if (h->opt_force_path == 3) return fail(h, OVQE_ERR_STATE, "program has no compact support (sparse path forced)");
    }
    if (h->n_global == 0 && h->n_local <= 16 && h->opt_force_path != 2)
        return run_small(h, B, (const double *)theta_dev, (double *)energies_dev, true);
    // larger registers: whole batches per pass of the sector tables, parameters and energies staying on the device
    if (h->opt_sector_batch && h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true) && h->ham.groups.size() >= 3) {
        rc = sector_prepare(h, true);
        if (rc) return rc;
        if (sector_batch_ready(h)) {
            bool ok = false;
            rc = run_sector_energy_batch(h, B, (const double *)theta_dev, true, (double *)energies_dev, &ok);
            if (rc) return rc;
            if (ok) return OVQE_OK;
        }
    }
    // anything else: through the host (one evaluation at a time on whatever path the program takes; B x K doubles down, B up)
    std::vector<double> th((size_t)B * K), en((size_t)B);
    HIPC(h, hipMemcpyAsync(th.data(), theta_dev, th.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    rc = ovqe_energy_batch(h, B, th.data(), K, en.data());
    if (rc) return rc;
    HIPC(h, hipMemcpyAsync(energies_dev, en.data(), en.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_energy(ovqe_handle h, const double *theta, int32_t K, double *energy) try {
    OVQE_ENTER(h);
    return ovqe_energy_batch(h, 1, theta, K, energy);
} OVQE_CATCH(h)

// Ascending list of the non-zero amplitudes of the state (d_nz_idx, d_nz_val) when they are at most 1/"screen_sparse" of the
// register; capacity: room for this many indices (0 = the support itself).
static int list_support(ovqe_handle h, uint64_t *support, bool *listed, uint64_t capacity) {
    *support = 0;
    *listed = false;
    if (h->opt_screen_sparse <= 0 || h->n_global != 0 || h->namps < 4096) return OVQE_OK;
    const uint64_t per_block = 256ull * NZ_PER_THREAD;
    const unsigned nbk = (unsigned)((h->namps + per_block - 1) / per_block);
    int rc = ensure(h, h->d_nz_cnt, (size_t)nbk * sizeof(uint32_t));
    if (!rc) rc = ensure(h, h->d_nz_start, (size_t)nbk * sizeof(uint64_t));
    if (rc) return rc;
    hipLaunchKernelGGL(k_nz_count, dim3(nbk), dim3(256), 0, h->stream, (const amp_t *)h->state, h->namps, (uint32_t *)h->d_nz_cnt.p);
    std::vector<uint32_t> cnt(nbk);
    hipError_t e = hipMemcpyAsync(cnt.data(), h->d_nz_cnt.p, (size_t)nbk * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, OVQE_ERR_HIP, std::string("support list: ") + hipGetErrorString(e));
    std::vector<uint64_t> start(nbk);
    uint64_t total = 0;
    for (unsigned b = 0; b < nbk; ++b) {
        start[b] = total;
        total += cnt[b];
    }
    *support = total;
    if (total == 0 || total * (uint64_t)h->opt_screen_sparse > h->namps) return OVQE_OK;
    rc = upload(h, h->d_nz_start, start.data(), (size_t)nbk * sizeof(uint64_t));
    if (!rc) rc = ensure(h, h->d_nz_idx, std::max(total, capacity) * sizeof(uint64_t));
    if (!rc) rc = ensure(h, h->d_nz_val, total * sizeof(amp_t));
    if (rc) return rc;
    hipLaunchKernelGGL(k_nz_fill, dim3(nbk), dim3(256), 0, h->stream, (const amp_t *)h->state, h->namps,
                       (const uint64_t *)h->d_nz_start.p, (uint64_t *)h->d_nz_idx.p, (amp_t *)h->d_nz_val.p);
    HIPC(h, hipGetLastError());
    *listed = true;
    return OVQE_OK;
}

// ---- sigma = H psi of the ADAPT screens from the materialised Hamiltonian of psi's symmetry sector ----------------------------
// (ref:openvqe/adapt/fermionic_adapt_vqe.py:114 `sig = hamiltonian_sparse.dot(curr_state)`.)  An ADAPT state of a dozen operators
// lists 10^5 amplitudes, spread over every tile of the register: the tile cover then costs its 61 sweeps of the whole register
// (24 qubits: 24 ms) for a vector of 600 k numbers.  The sector — the closure of psi's support under the Hamiltonian's x-groups —
// does not change while the ansatz grows, so the restricted Hamiltonian is materialised ONCE per Hamiltonian (the row-format tables
// of the sector path, sector_host.inc build_sector_h, without a circuit) and sigma is one pass over it (k_sector_apply).  Real
// Hamiltonians and real states only; anything else takes the register path.
static int build_screen_sector(ovqe_handle h, uint64_t support) {
    SectorEngine &E = h->scr;
    free_sector(E);   // (before the build's block cache opens: sector_host.inc build_sector)
    DevBlockScope kept_blocks(h->kept_blocks);
    E.ham_version = h->ham.version;
    h->scr_failed_version = h->ham.version;   // until everything below succeeded
    if (h->n_global != 0 || h->n_local > 32 || h->n_local < 12) return OVQE_OK;
    for (const HTerm &t : h->ham.terms)
        if (t.ci != 0.0) return OVQE_OK;       // an odd number of Y with a real coefficient (or a complex one): sigma is not real
    for (const HGroup &g : h->ham.groups)
        if (g.x > 0xffffffffull) return OVQE_OK;
    // closure of the listed support under the x-groups (D_g(j) != 0, residues snapped): the symmetry sector psi lives in
    const uint64_t cap = h->namps / (uint64_t)std::max(h->opt_sector_sparsity, 2);
    if (support > cap) return OVQE_OK;
    DevBuf list, bitmap, total_b;
    struct Scratch {   // released on every way out (the HIPC macro returns from the middle)
        DevBuf &a, &b, &c;
        ~Scratch() {
            free_buf(a);
            free_buf(b);
            free_buf(c);
        }
    } scratch{list, bitmap, total_b};
    auto done = [&](int code) { return code; };
    const size_t words = (size_t)std::max<uint64_t>(1, h->namps >> 5);
    int rc = ensure(h, list, cap * sizeof(uint64_t));
    if (!rc) rc = ensure(h, bitmap, words * sizeof(uint32_t) + 16);
    if (!rc) rc = ensure(h, total_b, 256);
    if (rc) return done(rc == OVQE_ERR_ALLOC ? ((void)hipGetLastError(), OVQE_OK) : rc);
    HIPC(h, hipMemcpyAsync(list.p, h->d_nz_idx.p, support * sizeof(uint64_t), hipMemcpyDeviceToDevice, h->stream));
    HIPC(h, hipMemsetAsync(bitmap.p, 0, words * sizeof(uint32_t), h->stream));
    hipLaunchKernelGGL(k_support_mark, dim3((unsigned)((support + 255) / 256)), dim3(256), 0, h->stream, (const uint64_t *)list.p, support,
                       (uint32_t *)bitmap.p);
    unsigned long long total = support;
    HIPC(h, hipMemcpyAsync(total_b.p, &total, sizeof(total), hipMemcpyHostToDevice, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    uint64_t first = 0, last = support;
    for (int round = 0; round < 256 && last > first; ++round) {
        hipLaunchKernelGGL(k_support_expand, dim3((unsigned)((last - first + 255) / 256)), dim3(256), 0, h->stream, (uint64_t *)list.p, first,
                           last, cap, h->base, (const HGroup *)h->ham.d_groups.p, (int)h->ham.groups.size(),
                           (const HTerm *)h->ham.d_terms.p, (uint32_t *)bitmap.p, (unsigned long long *)total_b.p);
        HIPC(h, hipMemcpyAsync(&total, total_b.p, sizeof(total), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        if (total > cap) return done(OVQE_OK);   // not a sparse sector
        first = last;
        last = total;
    }
    if (last > first || last > 0x7ffffff0ull) return done(OVQE_OK);
    const uint32_t K = (uint32_t)last;
    // ascending order (the sector path's canonical order), 32-bit indices
    DevBuf sorted;
    size_t tb = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(nullptr, tb, (const uint64_t *)list.p, (uint64_t *)nullptr, (int)K, 0, h->n_local, h->stream);
    if (e != hipSuccess) return done(fail(h, OVQE_ERR_HIP, std::string("screen sector: sort (size query): ") + hipGetErrorString(e)));
    DevBuf temp;
    rc = ensure(h, sorted, (size_t)K * sizeof(uint64_t));
    if (!rc) rc = ensure(h, temp, tb);
    if (!rc) rc = ensure(h, E.d_sup, (size_t)K * sizeof(uint32_t));
    if (rc) {
        free_buf(sorted);
        free_buf(temp);
        return done(rc == OVQE_ERR_ALLOC ? ((void)hipGetLastError(), OVQE_OK) : rc);
    }
    tb = temp.cap;
    e = hipcub::DeviceRadixSort::SortKeys(temp.p, tb, (const uint64_t *)list.p, (uint64_t *)sorted.p, (int)K, 0, h->n_local, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_scr_narrow, dim3((K + 255u) / 256u), dim3(256), 0, h->stream, (const uint64_t *)sorted.p, K, (uint32_t *)E.d_sup.p);
        e = hipStreamSynchronize(h->stream);
    }
    free_buf(sorted);
    free_buf(temp);
    if (e != hipSuccess) return done(fail(h, OVQE_ERR_HIP, std::string("screen sector: sort: ") + hipGetErrorString(e)));
    free_buf(list);   // (the closure list: the largest of the three, not needed while the tables are built)
    E.K = K;
    E.M = sector_tile_bits(h);
    E.chunk = (uint32_t)h->opt_sector_chunk;
    SectorScratch W;
    rc = ensure(h, W.inv_circ, (size_t)K * sizeof(uint32_t));
    if (rc) return rc == OVQE_ERR_ALLOC ? ((void)hipGetLastError(), OVQE_OK) : rc;
    hipLaunchKernelGGL(k_scr_iota, dim3((K + 255u) / 256u), dim3(256), 0, h->stream, (uint32_t *)W.inv_circ.p, K);
    size_t free_b = 0, total_mem = 0;
    HIPC(h, hipMemGetInfo(&free_b, &total_mem));
    const size_t budget = std::min<size_t>((size_t)std::max(h->opt_sector_max_gb, 0) << 30, free_b / 5 * 3);
    E.budget = budget;
    rc = build_sector_h(h, E, W, budget);
    if (rc == OVQE_ERR_ALLOC) {
        (void)hipGetLastError();
        free_sector(E);
        E.ham_version = h->ham.version;
        return OVQE_OK;
    }
    if (rc) return rc;
    if (!E.h_tables || E.hs.empty()) {
        free_sector(E);
        E.ham_version = h->ham.version;
        return OVQE_OK;
    }
    rc = ensure(h, E.d_buf[0], (size_t)K * sizeof(double));
    if (!rc) rc = ensure(h, E.d_buf[1], (size_t)K * sizeof(double));
    if (!rc) rc = ensure(h, E.d_flag, 256);
    if (rc) return rc;
    E.valid = true;
    h->scr_failed_version = -1;
    return OVQE_OK;
}
// sig (register) = (H + constant) psi through the screen engine; *used = false: the caller computes it on the register
static int screen_sector_sigma(ovqe_handle h, amp_t *sig, uint64_t support, bool *used) {
    *used = false;
    if (!h->opt_screen_sector || !h->opt_sector || h->n_global != 0 || support < (uint64_t)std::max(h->opt_screen_sector_min, 1)) return OVQE_OK;
    SectorEngine &E = h->scr;
    if (E.valid && E.ham_version != h->ham.version) free_sector(E);
    if (!E.valid) {
        if (h->scr_failed_version == h->ham.version) return OVQE_OK;
        int rc = build_screen_sector(h, support);
        if (rc) return rc;
        if (!E.valid) return OVQE_OK;
    }
    const uint32_t K = E.K;
    double *psic = (double *)E.d_buf[0].p, *sigc = (double *)E.d_buf[1].p;
    HIPC(h, hipMemsetAsync(psic, 0, (size_t)K * sizeof(double), h->stream));
    HIPC(h, hipMemsetAsync(E.d_flag.p, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_scr_compact, dim3((unsigned)((support + 255) / 256)), dim3(256), 0, h->stream, (const uint64_t *)h->d_nz_idx.p,
                       (const double2 *)h->d_nz_val.p, support, (const uint32_t *)E.d_sup.p, K, psic, (int *)E.d_flag.p);
    int flag = 0;
    HIPC(h, hipMemcpyAsync(&flag, E.d_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    if (flag) return OVQE_OK;   // psi has left the sector the tables were built for, or is complex: register path (tables kept)
    int rc = sector_matvec(h, E, psic, sigc);
    if (rc) return rc;
    HIPC(h, hipMemsetAsync(sig, 0, h->namps * sizeof(amp_t), h->stream));
    hipLaunchKernelGGL(k_scr_scatter, dim3((K + 255u) / 256u), dim3(256), 0, h->stream, (double2 *)sig, (const uint32_t *)E.d_sup.p, K,
                       (const double *)sigc, (const double *)psic, h->ham.constant);
    HIPC(h, hipGetLastError());
    h->last_screen_sector = (int64_t)K;
    *used = true;
    return OVQE_OK;
}

// ---- ADAPT --------------------------------------------------------------------------------------
int ovqe_pool_gradients(ovqe_handle h, int64_t n_ops, const int64_t *offsets, const uint64_t *x, const uint64_t *z,
                        const double *coeff_re, const double *coeff_im, int mode, double *grads) try {
    OVQE_ENTER(h);
    if (!h || n_ops < 0 || !offsets || (n_ops && !grads)) return OVQE_ERR_INVALID;
    if (mode != OVQE_GRAD_FERMIONIC && mode != OVQE_GRAD_QUBIT) return fail(h, OVQE_ERR_INVALID, "unknown gradient mode");
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (n_ops == 0) return OVQE_OK;
    const int64_t T = offsets[n_ops];
    if (T < 0 || (T && (!x || !z || !coeff_re))) return OVQE_ERR_INVALID;
    for (int64_t k = 0; k < n_ops; ++k)
        if (offsets[k] > offsets[k + 1] || offsets[k] < 0) return fail(h, OVQE_ERR_INVALID, "offsets not monotone");
    int rc = ensure_scratch(h, 0);
    if (rc) return rc;
    amp_t *sig = h->scratch[0];
    const int nb = reduce_blocks(h->namps);
    // the support of psi, when it is a small part of the register (exact zeros outside: ovqe_apply_exp_pauli_sum and the
    // rotation sweeps never write an amplitude they do not reach)
    uint64_t support = 0;
    bool on_support = false;
    h->last_screen_support = -1;
    rc = list_support(h, &support, &on_support, 0);
    if (rc) return rc;
    if (on_support) h->last_screen_support = (int64_t)support;
    // sigma = H psi (constant included)
    bool sector_sigma = false;
    h->last_screen_sector = 0;
    if (on_support) {
        rc = screen_sector_sigma(h, sig, support, &sector_sigma);
        if (rc) return rc;
    }
    if (!sector_sigma) rc = apply_hamiltonian(h, sig, h->state, h->ham.constant, on_support ? (const uint64_t *)h->d_nz_idx.p : nullptr, support);
    if (rc) return rc;
    std::vector<double2> vals(n_ops);
    {
        // terms in caller order with i^ny folded.  The pool is the same host data on every ADAPT iteration: its device
        // copy lives on the handle and is uploaded again only when the content changes.
        const uint64_t lmask = local_mask(h);
        std::vector<HTerm> terms(T);
        std::vector<uint64_t> xs(T);
        for (int64_t t = 0; t < T; ++t) {
            if (x[t] & ~lmask) return fail(h, OVQE_ERR_INVALID, "x mask touches global (rank) bits");
            const int ny = __builtin_popcountll(x[t] & z[t]) & 3;
            const double a = coeff_re[t], b = coeff_im ? coeff_im[t] : 0.0;
            HTerm ht;
            ht.z = z[t];
            switch (ny) {
            case 0: ht.cr = a; ht.ci = b; break;
            case 1: ht.cr = -b; ht.ci = a; break;
            case 2: ht.cr = -a; ht.ci = -b; break;
            default: ht.cr = b; ht.ci = -a; break;
            }
            terms[t] = ht;
            xs[t] = x[t];
        }
        const bool same = h->pg_valid && (int64_t)h->pg_off.size() == n_ops + 1 && (int64_t)h->pg_xs.size() == T &&
                          std::memcmp(h->pg_off.data(), offsets, (n_ops + 1) * sizeof(int64_t)) == 0 &&
                          (T == 0 || (std::memcmp(h->pg_xs.data(), xs.data(), T * sizeof(uint64_t)) == 0 &&
                                      std::memcmp(h->pg_terms.data(), terms.data(), T * sizeof(HTerm)) == 0));
        if (!same) {
            h->pg_valid = false;
            rc = upload(h, h->d_pg_off, offsets, (n_ops + 1) * sizeof(int64_t));
            if (!rc) rc = upload(h, h->d_pg_xs, xs.data(), std::max<int64_t>(T, 1) * sizeof(uint64_t));
            if (!rc) rc = upload(h, h->d_pg_terms, terms.data(), std::max<int64_t>(T, 1) * sizeof(HTerm));
            if (rc) return rc;
            h->pg_off.assign(offsets, offsets + n_ops + 1);
            h->pg_xs.swap(xs);
            h->pg_terms.swap(terms);
            h->pg_valid = true;
            // pattern tables of the same-x runs (see PoolRun)
            h->pg_tables = false;
            if (h->opt_screen_tables) {
                std::vector<PoolRun> runs((size_t)std::max<int64_t>(T, 1), PoolRun{0ull, 0u, -1, 0, 0});
                std::vector<double2> tabs;
                for (int64_t k = 0; k < n_ops; ++k)
                    for (int64_t t = offsets[k]; t < offsets[k + 1];) {
                        int64_t te = t + 1;
                        while (te < offsets[k + 1] && h->pg_xs[te] == h->pg_xs[t]) ++te;
                        const uint64_t z0 = h->pg_terms[t].z;
                        uint64_t V = 0;
                        for (int64_t u = t; u < te; ++u) V |= h->pg_terms[u].z ^ z0;
                        const int nv = __builtin_popcountll(V);
                        if (te - t >= 2 && nv <= 4) {
                            PoolRun r{z0, 0u, nv, (int32_t)tabs.size(), 0};
                            int pos[4] = {0, 0, 0, 0}, c = 0;
                            for (uint64_t mk = V; mk; mk &= mk - 1ull) pos[c++] = __builtin_ctzll(mk);
                            r.vpos = (uint32_t)pos[0] | ((uint32_t)pos[1] << 6) | ((uint32_t)pos[2] << 12) | ((uint32_t)pos[3] << 18);
                            for (int pat = 0; pat < (1 << nv); ++pat) {
                                uint64_t bits = 0;   // the pattern placed on V
                                for (int b = 0; b < nv; ++b)
                                    if ((pat >> b) & 1) bits |= 1ull << pos[b];
                                double dr = 0.0, di = 0.0;
                                for (int64_t u = t; u < te; ++u) {   // same terms, same order, fma: the device loop's doubles
                                    const double sg = (__builtin_popcountll(bits & (h->pg_terms[u].z ^ z0)) & 1) ? -1.0 : 1.0;
                                    dr = std::fma(h->pg_terms[u].cr, sg, dr);
                                    di = std::fma(h->pg_terms[u].ci, sg, di);
                                }
                                tabs.push_back(make_double2(dr, di));
                            }
                            runs[t] = r;
                        }
                        t = te;
                    }
                rc = upload(h, h->d_pg_runs, runs.data(), runs.size() * sizeof(PoolRun));
                if (tabs.empty()) tabs.push_back(make_double2(0.0, 0.0));
                if (!rc) rc = upload(h, h->d_pg_tabs, tabs.data(), tabs.size() * sizeof(double2));
                if (rc) return rc;
                h->pg_tables = true;
            }
        }
        // one workgroup per operator while the state re-streams from L2/MALL; above that 2^16 amplitudes per workgroup
        const int nchunks = on_support ? (int)std::min<uint64_t>(256, (support + 65535) >> 16)
                                       : h->n_local <= 22 ? 1 : (int)(h->namps >> 16);
        const int64_t ops_per_launch = 32768;
        rc = ensure(h, h->d_pg_out, n_ops * sizeof(double2));
        if (!rc && nchunks > 1)
            rc = ensure(h, h->d_pg_part, (size_t)std::min<int64_t>(n_ops, ops_per_launch) * nchunks * sizeof(double2));
        if (rc) return rc;
        for (int64_t op0 = 0; op0 < n_ops; op0 += ops_per_launch) {
            const int64_t cnt = std::min<int64_t>(ops_per_launch, n_ops - op0);
            double2 *out = (double2 *)h->d_pg_out.p + op0;
            double2 *part = nchunks > 1 ? (double2 *)h->d_pg_part.p : out;
            if (on_support)
                hipLaunchKernelGGL(k_pool_grad_nz, dim3((unsigned)nchunks, (unsigned)cnt), dim3(256), 0, h->stream,
                                   (const amp_t *)sig, (const uint64_t *)h->d_nz_idx.p, (const amp_t *)h->d_nz_val.p, support,
                                   h->base, (const int64_t *)h->d_pg_off.p, (const uint64_t *)h->d_pg_xs.p,
                                   (const HTerm *)h->d_pg_terms.p, op0, part,
                                   (h->pg_tables && h->opt_screen_tables) ? (const PoolRun *)h->d_pg_runs.p : (const PoolRun *)nullptr,
                                   (const double2 *)h->d_pg_tabs.p);
            else
                hipLaunchKernelGGL(k_pool_grad, dim3((unsigned)nchunks, (unsigned)cnt), dim3(256), 0, h->stream,
                                   (const amp_t *)sig, (const amp_t *)h->state, h->namps, h->base,
                                   (const int64_t *)h->d_pg_off.p, (const uint64_t *)h->d_pg_xs.p,
                                   (const HTerm *)h->d_pg_terms.p, op0, part);
            if (nchunks > 1)
                hipLaunchKernelGGL(k_reduce_rows2, dim3((unsigned)cnt), dim3(256), 0, h->stream, (const double2 *)part,
                                   nchunks, out);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(vals.data(), h->d_pg_out.p, n_ops * sizeof(double2), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) return fail(h, OVQE_ERR_HIP, std::string("pool_gradients: ") + hipGetErrorString(e));
    }
    for (int64_t k = 0; k < n_ops; ++k)
        grads[k] = mode == OVQE_GRAD_FERMIONIC ? 2.0 * vals[k].x : 2.0 * std::hypot(vals[k].x, vals[k].y);
    return OVQE_OK;
} OVQE_CATCH(h)

int ovqe_apply_exp_pauli_sum(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff_re,
                             const double *coeff_im, double theta) try {
    OVQE_ENTER(h);
    if (!h || T < 0 || (T && (!x || !z || !coeff_re))) return OVQE_ERR_INVALID;
    if (T == 0 || theta == 0.0) return OVQE_OK;
    std::vector<HGroup> groups;
    std::vector<HTerm> terms;
    int rc = build_groups(h, T, x, z, coeff_re, coeff_im, false, groups, terms);
    if (rc) return rc;
    double beta = 0.0;
    for (int64_t t = 0; t < T; ++t) beta += std::hypot(coeff_re[t], coeff_im ? coeff_im[t] : 0.0);
    beta *= std::fabs(theta);
    const int steps = std::max(1, (int)std::ceil(beta));
    const double tau = theta / steps, bstep = beta / steps;
    int M = 2;
    {
        double term = bstep * bstep / 2.0;
        while (term > 1e-19 && M < 64) {
            ++M;
            term *= bstep / M;
        }
    }
    rc = ensure_scratch(h, 0);
    if (!rc) rc = ensure_scratch(h, 1);
    if (rc) return rc;
    DevBuf dg, dt;
    rc = upload(h, dg, groups.data(), groups.size() * sizeof(HGroup));
    if (!rc) rc = upload(h, dt, terms.data(), terms.size() * sizeof(HTerm));
    const int nb = reduce_blocks(h->namps);
    // A state of a few determinants (the ADAPT state while the ansatz is short): the series only reaches the closure of its
    // support under the operator's x-groups.  The closure is listed once ("screen_sparse" bounds it) and every Taylor step
    // runs over the list; amplitudes equal the pass over the register bit for bit.
    uint64_t reach = 0;
    bool listed = false;
    h->last_exp_support = -1;
    if (!rc && h->opt_screen_sparse > 0) {
        const uint64_t cap = h->namps / (uint64_t)h->opt_screen_sparse;
        rc = list_support(h, &reach, &listed, cap);
        if (!rc && listed) {
            const size_t words = (size_t)std::max<uint64_t>(1, h->namps >> 5);
            rc = ensure(h, h->d_nz_bitmap, words * sizeof(uint32_t) + 16);
            unsigned long long *d_total = nullptr;
            if (!rc) {
                d_total = (unsigned long long *)((char *)h->d_nz_bitmap.p + words * sizeof(uint32_t) + (8 - (words * sizeof(uint32_t)) % 8) % 8);
                unsigned long long t0 = reach;
                hipError_t e = hipMemsetAsync(h->d_nz_bitmap.p, 0, words * sizeof(uint32_t), h->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(d_total, &t0, sizeof(t0), hipMemcpyHostToDevice, h->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "exp_pauli_sum: support bitmap");
            }
            if (!rc) {
                hipLaunchKernelGGL(k_support_mark, dim3((unsigned)((reach + 255) / 256)), dim3(256), 0, h->stream,
                                   (const uint64_t *)h->d_nz_idx.p, reach, (uint32_t *)h->d_nz_bitmap.p);
                uint64_t first = 0, last = reach;
                for (int round = 0; round < 4096 && !rc && listed && first < last; ++round) {
                    hipLaunchKernelGGL(k_support_expand, dim3((unsigned)((last - first + 255) / 256)), dim3(256), 0, h->stream,
                                       (uint64_t *)h->d_nz_idx.p, first, last, cap, h->base, (const HGroup *)dg.p, (int)groups.size(),
                                       (const HTerm *)dt.p, (uint32_t *)h->d_nz_bitmap.p, d_total);
                    unsigned long long total = 0;
                    hipError_t e = hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, h->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
                    if (e != hipSuccess) rc = fail(h, OVQE_ERR_HIP, std::string("exp_pauli_sum: ") + hipGetErrorString(e));
                    if (total > cap) listed = false;   // the series fills too much of the register: walk it
                    first = last;
                    last = total;
                }
                reach = last;
            }
        }
    }
    if (!rc && listed) h->last_exp_support = (int64_t)reach;
    for (int s = 0; s < steps && !rc; ++s) {
        amp_t *va = h->scratch[0], *vb = h->scratch[1];
        hipError_t e = hipMemcpyAsync(va, h->state, h->namps * sizeof(amp_t), hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess && listed && s == 0) e = hipMemsetAsync(vb, 0, h->namps * sizeof(amp_t), h->stream);
        if (e != hipSuccess) {
            rc = fail(h, OVQE_ERR_HIP, "exp_pauli_sum: copy failed");
            break;
        }
        for (int m = 1; m <= M; ++m) {
            if (listed)
                hipLaunchKernelGGL(k_apply_sum_list, dim3((unsigned)((reach + 255) / 256)), dim3(256), 0, h->stream, vb,
                                   (const amp_t *)va, h->state, (const uint64_t *)h->d_nz_idx.p, reach, h->base,
                                   (const HGroup *)dg.p, (int)groups.size(), (const HTerm *)dt.p, tau / m, 0.0);
            else
                hipLaunchKernelGGL(k_apply_sum, dim3(nb), dim3(256), 0, h->stream, vb, (const amp_t *)va, h->state, h->namps,
                                   h->base, (const HGroup *)dg.p, (int)groups.size(), (const HTerm *)dt.p, tau / m, 0.0, 0.0,
                                   0.0);
            std::swap(va, vb);
        }
        if (hipGetLastError() != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "exp_pauli_sum: launch failed");
    }
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "exp_pauli_sum: sync failed");
    if (dg.p) (void)hipFree(dg.p);
    if (dt.p) (void)hipFree(dt.p);
    return rc;
} OVQE_CATCH(h)

// ---- the state as a list of its non-zero amplitudes ---------------------------------------------------------------
int ovqe_get_support(ovqe_handle h, int64_t capacity, uint64_t *indices, double *amps, int64_t *count) try {
    OVQE_ENTER(h);
    if (!h || !count || capacity < 0 || (capacity && (!indices || !amps))) return OVQE_ERR_INVALID;
    *count = -1;
    if (h->n_global != 0 || h->namps < 4096) return OVQE_OK;   // small or sharded registers: ovqe_get_state
    const int keep = h->opt_screen_sparse;
    h->opt_screen_sparse = 1;   // list whatever the density
    uint64_t support = 0;
    bool listed = false;
    const int rc = list_support(h, &support, &listed, 0);
    h->opt_screen_sparse = keep;
    if (rc) return rc;
    *count = (int64_t)support;
    if (!listed || (int64_t)support > capacity) return OVQE_OK;   // (support = 0: nothing to copy)
    HIPC(h, hipMemcpyAsync(indices, h->d_nz_idx.p, support * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipMemcpyAsync(amps, h->d_nz_val.p, support * sizeof(amp_t), hipMemcpyDeviceToHost, h->stream));
    HIPC(h, hipStreamSynchronize(h->stream));
    return OVQE_OK;
} OVQE_CATCH(h)

// ---- ground state of the stored Hamiltonian: Lanczos on the device -------------------------------------------
namespace {

// lowest eigenpair of the symmetric tridiagonal matrix (a[0..m), b[0..m-1)): bisection on the Sturm count, then
// inverse iteration with a shift just below the eigenvalue (T - mu is positive definite: LDL^T without pivoting)
void tridiag_lowest(const std::vector<double> &a, const std::vector<double> &b, int m, double *lam,
                    std::vector<double> &s) {
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < m; ++i) {
        const double r = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i < m - 1 ? std::fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
    }
    const double scale = std::max({std::fabs(lo), std::fabs(hi), 1e-300});
    auto below = [&](double x) {  // number of eigenvalues < x
        int c = 0;
        double d = 1.0;
        for (int i = 0; i < m; ++i) {
            d = a[i] - x - (i > 0 ? b[i - 1] * b[i - 1] / d : 0.0);
            if (std::fabs(d) < 1e-300) d = -1e-300;
            if (d < 0.0) ++c;
        }
        return c;
    };
    for (int it = 0; it < 300 && hi - lo > 4e-16 * scale; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (below(mid) >= 1) hi = mid; else lo = mid;
    }
    *lam = 0.5 * (lo + hi);
    const double mu = *lam - 1e-9 * scale;
    std::vector<double> d(m), l(std::max(m - 1, 0));
    d[0] = a[0] - mu;
    for (int i = 0; i + 1 < m; ++i) {
        l[i] = b[i] / d[i];
        d[i + 1] = a[i + 1] - mu - l[i] * b[i];
    }
    s.assign(m, 1.0 / std::sqrt((double)m));
    for (int it = 0; it < 6; ++it) {
        for (int i = 1; i < m; ++i) s[i] -= l[i - 1] * s[i - 1];
        for (int i = 0; i < m; ++i) s[i] /= d[i];
        for (int i = m - 2; i >= 0; --i) s[i] -= l[i] * s[i + 1];
        double nrm = 0.0;
        for (int i = 0; i < m; ++i) nrm += s[i] * s[i];
        nrm = 1.0 / std::sqrt(nrm);
        for (int i = 0; i < m; ++i) s[i] *= nrm;
    }
}

struct Lanczos {
    ovqe_handle h;
    int nb;
    int reduce_to_host(double2 *out) {
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, h->stream, (const double2 *)h->d_partials.p, (int64_t)nb,
                           (double2 *)h->d_result.p, 0);
        HIPC(h, hipGetLastError());
        HIPC(h, hipMemcpyAsync(h->h_result, h->d_result.p, sizeof(double2), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        *out = h->h_result[0];
        return OVQE_OK;
    }
    int apply_h(amp_t *out, const amp_t *in) { return apply_hamiltonian(h, out, in, 0.0); }
    int dot(const amp_t *a, const amp_t *b, double2 *out) {
        hipLaunchKernelGGL(k_dot, dim3(nb), dim3(256), 0, h->stream, a, b, h->namps, (double2 *)h->d_partials.p);
        return reduce_to_host(out);
    }
    int update(amp_t *w, const amp_t *v, const amp_t *vprev, double alpha, double beta, double *norm) {
        hipLaunchKernelGGL(k_lanczos_update, dim3(nb), dim3(256), 0, h->stream, w, v, vprev, alpha, beta, h->namps,
                           (double2 *)h->d_partials.p);
        double2 r;
        int rc = reduce_to_host(&r);
        *norm = std::sqrt(r.x);
        return rc;
    }
    int start(amp_t *v, uint64_t seed) {
        hipLaunchKernelGGL(k_randomize, dim3(nb), dim3(256), 0, h->stream, v, h->namps, h->base, seed, 1.0,
                           (double2 *)h->d_partials.p);
        double2 r;
        int rc = reduce_to_host(&r);
        if (rc) return rc;
        hipLaunchKernelGGL(k_scale, dim3(nb), dim3(256), 0, h->stream, v, h->namps, 1.0 / std::sqrt(r.x));
        return OVQE_OK;
    }
};

}  // namespace

extern "C" int ovqe_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy,
                                 double *residual, int *iterations) try {
    OVQE_ENTER(h);
    if (!h || !energy || max_iter < 1 || !(tol > 0.0)) return OVQE_ERR_INVALID;
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (h->n_global) return fail(h, OVQE_ERR_INVALID, "ovqe_ground_state is single-device");
    int rc = ensure_scratch(h, 0);
    if (!rc) rc = ensure_scratch(h, 1);
    Lanczos L{h, reduce_blocks(h->namps)};
    if (!rc) rc = ensure(h, h->d_partials, (size_t)L.nb * sizeof(double2));
    if (!rc) rc = ensure(h, h->d_result, 64 * sizeof(double2));
    if (rc) return rc;
    max_iter = (int)std::min<uint64_t>((uint64_t)max_iter, h->namps);
    amp_t *tmp = nullptr;
    if (hipMalloc((void **)&tmp, h->namps * sizeof(amp_t)) != hipSuccess) return fail(h, OVQE_ERR_ALLOC, "hipMalloc Lanczos vector");
    std::vector<double> alpha, beta, s;
    double lam = 0.0, est = 0.0;
    int m = 0;
    // One pass while the Lanczos vectors fit in HBM (24 qubits: 256 MiB each, 150 of them = 40 GB of the 288): v_0..v_j stay
    // where they were written and the Ritz vector is their combination.  Beyond the budget ("lanczos_keep_gb", and never more
    // than 60 % of the free memory) the kept vectors are dropped and the recurrence is run a second time for the Ritz vector.
    std::vector<amp_t *> kept;
    size_t keep_budget = 0;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            keep_budget = std::min<size_t>((size_t)std::max(h->opt_lanczos_keep_gb, 0) << 30, free_b / 5 * 3);
    }
    const size_t vec_bytes = h->namps * sizeof(amp_t);
    bool keeping = keep_budget >= 8 * vec_bytes;
    auto drop_kept = [&]() {
        for (amp_t *v : kept) (void)hipFree(v);
        kept.clear();
        keeping = false;
    };
    auto recurrence = [&](bool accumulate) -> int {
        amp_t *A = h->scratch[0], *B = h->scratch[1], *C = tmp;  // v_{j-1}, v_j, w
        int r = L.start(B, seed);
        if (r) return r;
        if (accumulate)
            hipLaunchKernelGGL(k_axpy_real, dim3(L.nb), dim3(256), 0, h->stream, h->state, (const amp_t *)B, s[0], h->namps, 1);
        const int steps = accumulate ? m - 1 : max_iter;
        for (int j = 0; j < steps; ++j) {
            r = L.apply_h(C, B);
            if (r) return r;
            double bj;
            if (accumulate) {
                r = L.update(C, B, j ? A : nullptr, alpha[j], j ? beta[j - 1] : 0.0, &bj);
                if (r) return r;
                bj = beta[j];
            } else {
                double2 d;
                r = L.dot(B, C, &d);
                if (r) return r;
                alpha.push_back(d.x);
                r = L.update(C, B, j ? A : nullptr, d.x, j ? beta[j - 1] : 0.0, &bj);
                if (r) return r;
                m = j + 1;
                const bool last = j + 1 == steps || bj < 1e-13 * std::max(1.0, std::fabs(d.x));
                if (last || (j >= 4 && j % 5 == 4)) {
                    tridiag_lowest(alpha, beta, m, &lam, s);
                    est = std::fabs(bj * s[m - 1]);
                    if (last || est < tol * std::max(1.0, std::fabs(lam))) return OVQE_OK;
                }
                beta.push_back(bj);
            }
            hipLaunchKernelGGL(k_scale, dim3(L.nb), dim3(256), 0, h->stream, C, h->namps, 1.0 / bj);
            if (!accumulate && keeping) {
                // v_j (B) has served as v_{j-1}'s successor: it stays in `kept` from here on; the three working buffers are
                // re-filled from fresh allocations as long as the budget lasts.  A (v_{j-1}) is already kept or is a work buffer.
                amp_t *fresh = nullptr;
                if ((kept.size() + 1) * vec_bytes <= keep_budget && hipMalloc((void **)&fresh, vec_bytes) == hipSuccess) {
                    // keep a copy of v_j: device-to-device copy at HBM rate (0.1 ms at 24 qubits) next to a 40 ms H psi
                    if (hipMemcpyAsync(fresh, B, vec_bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess) {
                        (void)hipFree(fresh);
                        drop_kept();
                    } else {
                        kept.push_back(fresh);
                    }
                } else {
                    (void)hipGetLastError();
                    drop_kept();
                }
            }
            amp_t *t = A;
            A = B;
            B = C;
            C = t;
            if (accumulate)
                hipLaunchKernelGGL(k_axpy_real, dim3(L.nb), dim3(256), 0, h->stream, h->state, (const amp_t *)B, s[j + 1],
                                   h->namps, 0);
        }
        return OVQE_OK;
    };
    rc = recurrence(false);       // pass 1: the tridiagonal matrix
    if (!rc && keeping && (int)kept.size() == m - 1) {
        // kept = v_0..v_{m-2}; v_{m-1} is the vector the last step multiplied, still in its work buffer: scratch[(m-1) % 3 ...]
        // (the rotation A <- B <- C <- A moves one buffer per step, starting from B = scratch[1])
        amp_t *ring[3] = {h->scratch[1], tmp, h->scratch[0]};
        const amp_t *vlast = ring[(m - 1) % 3];
        for (int j = 0; j < m; ++j)
            hipLaunchKernelGGL(k_axpy_real, dim3(L.nb), dim3(256), 0, h->stream, h->state, j < m - 1 ? (const amp_t *)kept[j] : vlast,
                               s[j], h->namps, j == 0 ? 1 : 0);
        if (hipGetLastError() != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "ground_state: launch failed");
    } else if (!rc) {
        drop_kept();
        rc = recurrence(true);  // pass 2: the Ritz vector, same recurrence
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess && !rc) rc = fail(h, OVQE_ERR_HIP, "ground_state: sync failed");
    drop_kept();
    double true_res = 0.0;
    if (!rc) {
        // normalise, Rayleigh quotient and true residual |H y - lambda y|
        double2 d;
        rc = L.dot(h->state, h->state, &d);
        if (!rc) {
            hipLaunchKernelGGL(k_scale, dim3(L.nb), dim3(256), 0, h->stream, h->state, h->namps, 1.0 / std::sqrt(d.x));
            rc = L.apply_h(tmp, h->state);
            if (!rc) rc = L.dot(h->state, tmp, &d);
        }
        if (!rc) {
            lam = d.x;
            rc = L.update(tmp, h->state, nullptr, lam, 0.0, &true_res);
        }
    }
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "ground_state: launch failed");
    (void)hipFree(tmp);
    if (rc) return rc;
    *energy = lam + h->ham.constant;
    if (residual) *residual = true_res;
    if (iterations) *iterations = m;
    (void)est;
    return OVQE_OK;
} OVQE_CATCH(h)

// ---- lowest eigenpair inside the support of the stored program (sector tables) ------------------------------------
extern "C" int ovqe_sector_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy, double *residual,
                                        int *iterations) try {
    OVQE_ENTER(h);
    if (!h || !energy || max_iter < 1 || !(tol > 0.0)) return OVQE_ERR_INVALID;
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (!h->prog_set) {
        // no program: the sector of the state in the buffer — the closure of its support under the Hamiltonian's x-groups, e.g. of
        // the Hartree-Fock determinant after ovqe_init_basis: the (N_alpha, N_beta) sector, no UCCSD program needed to name it
        if (!h->opt_sector || h->n_global != 0) return fail(h, OVQE_ERR_STATE, "no program set, and the sector of the current state needs option sector on one device");
        uint64_t support = 0;
        bool listed = false;
        int rc = list_support(h, &support, &listed, 0);
        if (rc) return rc;
        if (!listed) return fail(h, OVQE_ERR_STATE, "no program set, and the state in the buffer is empty or not sparse (ovqe_init_basis first)");
        SectorEngine &S = h->scr;
        if (S.valid && S.ham_version != h->ham.version) free_sector(S);
        if (S.valid) {   // does the state live in the sector these tables were built for ?
            HIPC(h, hipMemsetAsync(S.d_buf[0].p, 0, (size_t)S.K * sizeof(double), h->stream));
            HIPC(h, hipMemsetAsync(S.d_flag.p, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(k_scr_compact, dim3((unsigned)((support + 255) / 256)), dim3(256), 0, h->stream, (const uint64_t *)h->d_nz_idx.p,
                               (const double2 *)h->d_nz_val.p, support, (const uint32_t *)S.d_sup.p, S.K, (double *)S.d_buf[0].p, (int *)S.d_flag.p);
            int flag = 0;
            HIPC(h, hipMemcpyAsync(&flag, S.d_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPC(h, hipStreamSynchronize(h->stream));
            if (flag & 1) free_sector(S);
        }
        if (!S.valid) {
            rc = build_screen_sector(h, support);
            if (rc) return rc;
        }
        if (!S.valid) return fail(h, OVQE_ERR_STATE, "no sector tables for the current state (complex Hamiltonian, closure denser than 1/sector_sparsity, or tables beyond sector_max_gb)");
        // the Lanczos block starts from the first listed determinant of the state
        uint64_t first_index = 0;
        HIPC(h, hipMemcpyAsync(&first_index, h->d_nz_idx.p, sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        std::vector<uint32_t> sup(S.K);
        HIPC(h, hipMemcpyAsync(sup.data(), S.d_sup.p, (size_t)S.K * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPC(h, hipStreamSynchronize(h->stream));
        const auto it = std::lower_bound(sup.begin(), sup.end(), (uint32_t)first_index);
        if (it == sup.end() || *it != (uint32_t)first_index) return fail(h, OVQE_ERR_STATE, "internal: state outside its own sector");
        S.hf_final = (uint32_t)(it - sup.begin());
        return run_sector_ground_state(h, S, tol, max_iter, seed, energy, residual, iterations);
    }
    FrameHamGuard frame_guard(h);
    const bool real = h->opt_real_stream && h->prog_real_ok && h->n_global == 0 && tile_ok(h, true) && h->ham.groups.size() >= 3;
    if (!real || !h->opt_sector) return fail(h, OVQE_ERR_STATE, "the stored program has no sector tables (real-amplitude program on one device needed)");
    SectorEngine &E = h->sec;
    if (E.prog_version != h->prog_version || E.ham_version != h->ham.version) {
        free_sector(E);
        E.disabled = false;
        E.seen = 0;
        E.probe_mode = 0;
        E.coset_rejected = false;
        E.prog_version = h->prog_version;
        E.ham_version = h->ham.version;
    }
    if (!E.valid && !E.disabled) {   // built on demand here: this call is what the tables are for
        int rc = build_sector(h);
        if (rc) return rc;
    }
    if (!E.valid || !E.h_tables)
        return fail(h, OVQE_ERR_STATE, "the stored program has no sector tables (support too dense, or the tables exceed sector_max_gb)");
    int rc = run_sector_ground_state(h, h->sec, tol, max_iter, seed, energy, residual, iterations);
    // Lanczos diagonalised C^+ H C on the rotation-only program's support: the vector in the buffer is C^+ |psi0>; the header
    // promises the eigenvector of the caller's H there (fidelities are taken against it), so the Clifford part goes on top
    if (!rc && h->frame_open) rc = apply_tail_gates(h);
    return rc;
} OVQE_CATCH(h)

// ---- exact gradient by the adjoint method -------------------------------------------------------------------
extern "C" int ovqe_energy_gradient(ovqe_handle h, const double *theta, int32_t K, double *energy, double *grad) try {
    OVQE_ENTER(h);
    if (!h || !energy || !grad) return OVQE_ERR_INVALID;
    int rc = check_theta(h, theta, K);
    if (rc) return rc;
    if (!h->ham.set) return fail(h, OVQE_ERR_STATE, "no Hamiltonian set (ovqe_set_hamiltonian)");
    if (h->n_global) return fail(h, OVQE_ERR_INVALID, "ovqe_energy_gradient is single-device");
    FrameHamGuard frame_guard(h);
    {   // small registers: forward, H psi and the backward pass in one launch on the compact support
        bool done = false;
        rc = run_sparse_gradient(h, theta, energy, grad, &done);
        if (rc || done) return rc;
    }
    if (h->opt_real_stream && h->prog_real_ok && tile_ok(h, true) && h->ham.groups.size() >= 3) {
        // real-amplitude program on a sparse support: the whole adjoint pass on the sector tables
        rc = sector_prepare(h, true);
        if (rc) return rc;
        if (h->sec.valid && h->sec.h_tables && sector_gradient_fits(h)) {   // (else: tiles sized for energies only, "sector_tile_cap")
            bool ok = false;
            rc = run_sector_gradient(h, theta, energy, grad, &ok);
            if (rc || ok) return rc;
            sector_orphaned(h);
            if (!h->sec.disabled) {   // second attempt: support probed with independent angles
                rc = sector_prepare(h, true);
                if (rc) return rc;
                if (h->sec.valid && h->sec.h_tables && sector_gradient_fits(h)) {
                    rc = run_sector_gradient(h, theta, energy, grad, &ok);
                    if (rc || ok) return rc;
                    sector_orphaned(h);
                }
            }
        }
    }
    rc = run_program_streaming(h, theta);  // psi = U(theta)|hf>; angle table: original rotations at offset S
    if (!rc) rc = ensure_scratch(h, 0);
    Lanczos L{h, reduce_blocks(h->namps)};
    // backward sweeps: one pair per thread while that stays below 65536 workgroups (a grid-stride loop of dependent
    // load -> rotate -> store trips exposes the memory latency of both states), partial sums per workgroup and rotation
    const int nb = (int)std::min<uint64_t>(65536, std::max<uint64_t>(1, (h->namps / 2 + 255) / 256));
    if (!rc) rc = ensure(h, h->d_partials, (size_t)std::max(L.nb, ADJ_MAX_ROT * nb) * sizeof(double2));
    if (!rc) rc = ensure(h, h->d_result, 64 * sizeof(double2));
    const size_t R = h->rots.size(), S = h->srots.size();
    DevBuf d_w;
    if (!rc) rc = ensure(h, d_w, std::max<size_t>(R, 1) * sizeof(double));
    if (rc) return rc;
    amp_t *lam = h->scratch[0];
    double2 e = make_double2(0.0, 0.0);
    rc = L.apply_h(lam, h->state);
    if (!rc) rc = L.dot(h->state, lam, &e);
    const RotParam *d_rp = (const RotParam *)h->d_rp.p + S;
    double *partials = (double *)h->d_partials.p;
    for (int oi = (int)h->ops.size() - 1; oi >= 0 && !rc; --oi) {
        const SmallOp &op = h->ops[oi];
        switch (op.kind) {
        case OP_PAIR:
        case OP_DIAG:
            for (int hi = op.count; hi > 0; hi -= ADJ_MAX_ROT) {  // chunks from the end of the run backwards
                const int lo = std::max(0, hi - ADJ_MAX_ROT), cnt = hi - lo;
                if (op.kind == OP_PAIR)
                    hipLaunchKernelGGL(k_adjoint_pairs, dim3(nb), dim3(256), 0, h->stream, h->state, lam, h->namps >> 1,
                                       op.pivot, op.x, h->base, d_rp + op.first + lo, cnt, partials);
                else
                    hipLaunchKernelGGL(k_adjoint_diag, dim3(nb), dim3(256), 0, h->stream, h->state, lam, h->namps,
                                       h->base, d_rp + op.first + lo, cnt, partials);
                hipLaunchKernelGGL(k_reduce_rows, dim3(cnt), dim3(256), 0, h->stream, (const double *)partials, nb,
                                   (double *)d_w.p + op.first + lo);
            }
            break;
        case OP_X:
            rc = launch_gate(h, 0, op.pivot, 0);
            if (!rc) rc = launch_gate(h, 0, op.pivot, 0, lam);
            break;
        case OP_H:
            rc = launch_gate(h, 1, op.pivot, 0);
            if (!rc) rc = launch_gate(h, 1, op.pivot, 0, lam);
            break;
        case OP_CNOT:
            rc = launch_gate(h, 2, op.first, op.count);
            if (!rc) rc = launch_gate(h, 2, op.first, op.count, lam);
            break;
        default: rc = fail(h, OVQE_ERR_INVALID, "corrupt program");
        }
    }
    std::vector<double> w(R, 0.0);
    if (!rc && hipGetLastError() != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "energy_gradient: launch failed");
    if (!rc && R && hipMemcpyAsync(w.data(), d_w.p, R * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess)
        rc = fail(h, OVQE_ERR_HIP, "energy_gradient: copy failed");
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, OVQE_ERR_HIP, "energy_gradient: sync failed");
    if (d_w.p) (void)hipFree(d_w.p);
    if (rc) return rc;
    for (int32_t p = 0; p < K; ++p) grad[p] = 0.0;
    for (size_t r = 0; r < R; ++r) {
        const SmallRot &sr = h->rots[r];
        if (sr.pidx >= 0) grad[sr.pidx] += 2.0 * sr.coeff * ((sr.ny & 2) ? -w[r] : w[r]);
    }
    *energy = e.x + h->ham.constant;
    return OVQE_OK;
} OVQE_CATCH(h)

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
    if (!h || !support || which < 0 || which > 5) return OVQE_ERR_INVALID;
    const int64_t v[6] = {h->last_screen_support, h->last_exp_support, h->last_screen_sector, h->last_fci_rounds, h->last_passes, h->last_pass_bytes};
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
