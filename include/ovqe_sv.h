/* ovqe_sv.h — C ABI of the MI355X-native statevector backend for OpenVQE's VQE / ADAPT-VQE
 * inner loop (libovqe_sv.so, gfx950 only).
 *
 * The reference (OpenVQE, pure Python) has no plugin ABI of its own: its hot path talks to the
 * third-party myQLM objects.  Each entry point below names the reference call
 * site(s) whose work it replaces ("ref:" = /root/reference/).
 *
 * Conventions
 *   - n-qubit state = 2^n complex<double> amplitudes (re,im interleaved, 16 B) resident in HBM.
 *   - reference qubit q  <->  basis-index bit (n-1-q)  (qubit 0 is the MSB:
 *     ref:openvqe/ucc_family/get_energy_qucc.py:40-45,
 *     ref:openvqe/common_files/molecule_factory_with_sparse.py:622-642,
 *     ref:openvqe/adapt/qubit_adapt_vqe.py:111-120).  ALL masks / bit numbers in this header are in
 *     basis-index bit space.
 *   - a Pauli string is two uint64 masks (x,z): I=(0,0) X=(1,0) Z=(0,1) Y=(1,1);
 *     P = i^{popcount(x&z)} X^x Z^z.
 *   - every function returns 0 on success or a negative OVQE_ERR_*; ovqe_last_error() gives text.
 *     No C++ exception crosses the ABI: every entry point is a function-try-block; std::bad_alloc -> OVQE_ERR_ALLOC,
 *     any other exception -> OVQE_ERR_INVALID with its what() in ovqe_last_error.  A handle is not thread-safe; distinct handles are
 *     independent.  Calls are synchronous unless stated (results are on the host on return).
 *   - host arrays are borrowed for the duration of the call only.
 *   - sharded states (multi-GPU): a handle may own one shard of 2^n_local amplitudes of a
 *     (n_local + n_global)-qubit state; shard s holds basis indices [s << n_local, (s+1) << n_local).
 *     Masks then span n_local+n_global bits; x bits must be local (the host layer makes them so by
 *     exchanging half shards, openvqe_amd/distributed.py), z bits may be global (rank-dependent sign).
 */
#ifndef OVQE_SV_H
#define OVQE_SV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVQE_OK 0
#define OVQE_ERR_INVALID (-1)   /* bad argument */
#define OVQE_ERR_NO_DEVICE (-2) /* no usable gfx950 device */
#define OVQE_ERR_HIP (-3)       /* HIP runtime error (text in ovqe_last_error) */
#define OVQE_ERR_ALLOC (-4)     /* device/host allocation failed */
#define OVQE_ERR_STATE (-5)     /* call order: program / Hamiltonian not set */

/* gate opcodes of ovqe_set_gate_program / ovqe_apply_gate
 * (gate set of ref:openvqe/common_files/circuit.py:1 and ref:openvqe/ucc_family/get_energy_qucc.py:4;
 *  RX(a)=exp(-iaX/2), RY(a)=exp(-iaY/2), RZ(a)=diag(e^{-ia/2},e^{ia/2}), CNOT(control,target)) */
#define OVQE_GATE_X 0
#define OVQE_GATE_H 1
#define OVQE_GATE_RX 2
#define OVQE_GATE_RY 3
#define OVQE_GATE_RZ 4
#define OVQE_GATE_CNOT 5

/* pool-gradient modes */
#define OVQE_GRAD_FERMIONIC 0 /* g = 2 Re <sigma|A|psi>   ref:openvqe/adapt/fermionic_adapt_vqe.py:67-73 */
#define OVQE_GRAD_QUBIT 1     /* g = 2 |<sigma|P|psi>|    ref:openvqe/adapt/qubit_adapt_vqe.py:147-150 */

typedef struct ovqe_sv *ovqe_handle;

int ovqe_version(void);
/* text of the last error on this handle (h may be NULL: last error of a failed create) */
const char *ovqe_last_error(ovqe_handle h);
int ovqe_device_count(int *count);

/* ---- lifecycle.  Replaces the per-submit state allocation of qat.qpus.get_default_qpu().submit
 * (ref:openvqe/ucc_family/get_energy_ucc.py:38-48): the library owns the device buffers.
 * 1 <= n_qubits (n_local) <= 33 per device (128 GiB state); larger registers are sharded. */
int ovqe_create(int n_qubits, int device, ovqe_handle *out);
/* one shard of a distributed state: n_local local bits, n_global rank bits, this shard's index */
int ovqe_create_shard(int n_local, int n_global, uint64_t shard_index, int device, ovqe_handle *out);
/* a handle whose state IS the caller's device buffer of 2^n_qubits amplitudes (16 bytes each) from the start: nothing of that
 * size is allocated (ovqe_create + ovqe_adopt_state would allocate a state and drop it).  The sharded register contracts CHUNKS of
 * partner shards through such views (the chunk contractions of <psi|H|psi> / sigma = H psi over a partitioned register, the
 * multi-device form of ref:openvqe/adapt/fermionic_adapt_vqe.py:114 and get_energy_ucc.py:47-48); the buffer must outlive the handle. */
int ovqe_create_view(int n_qubits, int device, void *dev_ptr, ovqe_handle *out);
int ovqe_destroy(ovqe_handle h);
/* run this handle's kernels on a caller-owned hipStream_t (NULL = default stream) */
int ovqe_set_stream(ovqe_handle h, void *hip_stream);
/* Options.  None of them changes a result beyond rounding (tests/test_gpu_tile.py, test_gpu_kernels.py, test_gpu_sector.py compare the
 * settings with one another and with the oracle); unknown names return OVQE_ERR_INVALID.  Three groups:
 *
 * (A) WHICH PATH RUNS — what a caller may reasonably set
 *   "force_path"      0 automatic (default), 1 fused small-register kernel, 2 streaming kernels, 3 support-compacted kernel
 *   "clifford_frame"  read by the NEXT ovqe_set_gate_program.  1 (default): a gate list whose Clifford part (X, H, CNOT, quarter-turn
 *                     rotations) multiplies to the identity runs as the algebraically identical sequence of Pauli rotations with
 *                     conjugated strings; when it does NOT close (interleaved CNOT ladders, stray gates) that rotation sequence runs
 *                     too, energies / gradients then use the stored Hamiltonian conjugated by the net Clifford operator
 *                     (<C phi|H|C phi> = <phi|C^+ H C|phi>, every term stays one Pauli string) and ovqe_prepare_state applies the
 *                     Clifford gates behind the rotations.  0: execute the literal list; 2: frame form always, Clifford part appended
 *                     literally; 3: frame form only when the frame closes
 *   "real_mode" / "real_stream" (1)  a program whose rotation strings all have an odd number of Y (every UCC / ADAPT generator, the
 *                     QUCCSD templates in frame form) keeps the amplitudes real: 8-byte amplitudes in the fused kernel / 2^n doubles in
 *                     streaming energies (n >= 15): half the bytes per sweep; ovqe_prepare_state always delivers the complex state
 *   "table_fusion" (1) commuting same-x runs become single sparse pair rotations in the fused kernel
 *   "clifford_phase_host" (1)  the global phase that the Clifford part of a closed gate list gives |hf> (ovqe_prepare_state reproduces the
 *                     literal circuit's phase) from a sparse simulation on the host; 0 or more than 4096 basis states: the gates run on
 *                     the device
 *   "sparse" (1)      support-compacted kernel when the program's reachable support fits one workgroup's LDS
 *   "compact" (1)     compact cover: from the second evaluation of a (program, Hamiltonian) pair, <H> of a real-amplitude streaming
 *                     energy runs over a compact copy of the state's support
 *   "sector" (1)      sector path: from the second energy evaluation (the first gradient call; the first evaluation for programs of at most
 *                     2048 rotations) a real-amplitude program whose states occupy at most 1/"sector_sparsity" (4) of
 *                     the register runs entirely on that support: circuit over compact tiles, <H> from the Hamiltonian materialised on
 *                     the support; tables in device memory up to "sector_max_gb" (128; and 60 % of the free memory) — beyond it <H> goes
 *                     through the compact cover; "sector_min_qubits" (18); "sector_h" 0: never materialise <H>.  When an evaluation meets
 *                     a non-zero amplitude whose partner is outside the probed support it is redone by the dense kernels and the tables are
 *                     rebuilt once from a probe with one angle per rotation.  The state buffer holds unspecified data after an energy
 *                     evaluation on this path.
 *   "sector_regular" (1)  supports that are the full coset of the program's Z2 symmetries (the spin-parity quarter that the reference's
 *                     QUCCSD templates populate): circuit sweeps — and with "sector_reg_adjoint" (1) the backward sweeps of
 *                     ovqe_energy_gradient — from bit arithmetic, no pair-word tables for supports of 2^20 amplitudes and more; 0: pair
 *                     words; 3: without the slot orders that make the gathers between sweeps run in runs (measurement)
 *   "sector_batch" (1) ovqe_energy_batch / _device run whole batches per pass of the sector tables
 *   "poll_result" (1)  the host of a lone evaluation — or of a batch of at most 256 on the fused kernels (a finite-difference gradient) —
 *                     watches the mapped slots its kernels write instead of waiting for the stream's completion signal (6 us per call;
 *                     evaluations of the sector tables that took less than 2 ms the time before, the fused kernels always); nothing
 *                     within 5 ms: the stream is synchronised after all
 *   "sector_fused_reduce" (1)  the final reduction of a lone evaluation on the sector tables writes energy and orphan flag into mapped
 *                     host memory; 0: reduction launch + two copies
 *   "screen_sparse" (16) / "screen_sector" (1) / "screen_sector_min" (1024)  ADAPT screens: bilinear forms summed over
 *                     the listed non-zero amplitudes of psi while they are at most 1/value of the register (0: never); sigma = H psi from the
 *                     materialised Hamiltonian of psi's symmetry sector once psi lists that many amplitudes; pattern tables for the pool's
 *                     same-x runs (see ovqe_pool_gradients, ovqe_last_support); ovqe_apply_exp_pauli_sum runs its Taylor steps over the
 *                     closure of the support under the operator's x-groups within the same bound (bit-identical amplitudes)
 *   "lanczos_keep_gb" (160) ovqe_ground_state keeps its Lanczos vectors in HBM up to this many GB (and 60 % of the free memory): one pass of
 *                     the recurrence gives the Ritz vector; 0 or vectors that do not fit: the recurrence runs twice
 *
 * (B) GEOMETRY — defaults are the measured optimum on MI355X
 *   streaming path: "tile_bits" (-1 automatic: 12 for n >= 25, else 11; 0 = one sweep per op), "tile_low" (4), "apply_min_tiles" (256),
 *                     "index_streams"
 *   fused / support-compacted kernels: "small_max_qubits", "small_batch_max_qubits", "sparse_grad" (1: all derivatives of a register of at most
 *                     16 qubits in one fused launch on the compact support; 0: the streaming adjoint pass), "sparse_renumber" (1)
 *   sector path: "sector_bits" / "sector_h_bits" (index bits per circuit / <H> tile, 0 automatic), "sector_tile_cap" (6500 amplitudes per
 *                     circuit tile so that gradients fit; up to 14000 for energies only), "sector_threads" (0 automatic, 64, 256, 512,
 *                     1024), "sector_adjoint" (backward sweeps of the gradient: 3 on the per-wave streams where a sweep has them, 2 on the 64-bit
 *                     pair tables, 1 first form), "sector_dict" (1: dictionary-coded matrix elements, dictionary from a sample of the
 *                     stream first; 2: from all values; 3: from a sample too thin to be complete — the fall-back, tests; 0: explicit values),
 *                     "sector_reg_threads" (256; 128, 512, 1024), "sector_reg_pairs" (1: blocks of two three-bit ops), "sector_reg_runs"
 *                     (1: runs of ops whose waves stay inside their own slots run without barriers), "sector_pairs_form" (2: pair-table
 *                     builder with the ops staged in LDS; 1: first form).  Combinations the second sweep form has no kernel for fall back to
 *                     the first form / to one evaluation at a time.
 *   "sector_profile"  HIP-event times of the two halves of a sector evaluation in ovqe_program_info
 *
 * (C) MEASUREMENT AND TESTS — accepted ONLY by the testing build of this same source (-DOVQE_TESTING: libovqe_sv_testing.so, loaded by
 *   tests/test_gpu_abi.py and, through OVQE_LIB=testing, by the scripts under tools/); the product library answers OVQE_ERR_INVALID
 *   "unknown option" and runs every one of them at its default:
 *   "fault_inject" (1: the next term-list build throws std::bad_alloc: the exception barrier's test), "sector_debug" (2: say on stderr why a
 *   program was left to the dense kernels; 4: wall time of the build's phases), "sector_sweep_dbg" / "sector_h_dbg" / "sparse_dbg" (kernels
 *   truncated after a given phase), "rot_variant", and the launch geometries and superseded forms kept for comparison: "unroll",
 *   "ham_tile_low", "expect_sparse", "expect_streams", "persist_blocks", "compact_cpp", "small_threads", "sparse_rows", "sparse_wg",
 *   "sparse_spw", "sparse_dealias", "sector_sweep" (circuit sweeps on an irregular support: 3 pair words in per-wave streams with barriers at
 *   run boundaries only — built on top of the tables of 2 —, 2 64-bit pair words in registers with a barrier per round, 1 first form; 2 on a
 *   handle built under 3 runs the second form on the same tables; 4: the streams for the states of a batch too — the product gives
 *   batches the second form: their workgroups hide each other's barriers and the streams gain them nothing), "sector_stream_waves" (0: waves that share a tile's rows from the pairs per
 *   op of the sweep's largest tile; 1, 2, 4, 8, 16), "sector_stream_arrange" (1: the lanes of a row chosen for the LDS banks), "sector_h_pack" (1: <H> sweeps with at most 1023
 *   magnitudes keep their coded words as 24-bit elements), "sector_chunk",
 *   "sector_depth2", "sector_many_tiles",
 *   "sector_h_lpt", "sector_h_threads", "sector_h_groups", "sector_apply_threads", "sector_eager_rots",
 *   "screen_tables", "sector_batch_threads" / "_nb" / "_sweep_threads" / "_dst_lds" / "_zfast", "tile_flat" (tiled <H>: entries of one
 *   or two merged terms as per-lane items 1 / per-wave entries 0 / items for real states only 2, the default), "sector_apply_seq" (1: lambda = H psi on the sector tables runs one launch per sweep in sequence with plain
 *   additions where one sweep fills the chip; 0: one launch, global atomics), "tile_unsplit" (1: tiled <H> of complex states takes groups
 *   of one or two terms as unsplit entries), "sector_coset_first" (1: a gate list in frame form takes the coset of its Z2 symmetries as
 *   its support without a probe run; the build checks afterwards that the coset is populated), "expect_dense" (1: the
 *   first sweep of a tiled <H> of a complex register of 25+ qubits counts its sparse tiles; none: the other sweeps run without the
 *   sparse path's LDS, two workgroups per CU) */
int ovqe_set_option(ovqe_handle h, const char *name, int64_t value);
/* device pointer to the 2^n_local amplitudes (for RCCL exchange by the host layer).  The caller may write them between calls: from
 * here on the handle keeps nothing it learnt about the state from one call to the next (the support list that a chain of
 * ovqe_apply_exp_pauli_sum calls otherwise carries over is listed afresh per call) */
int ovqe_state_ptr(ovqe_handle h, void **dev_ptr);
/* use caller-owned device memory (e.g. a torch tensor) of 2^n_local*16 bytes as the state buffer */
int ovqe_adopt_state(ovqe_handle h, void *dev_ptr);

/* ---- state set-up / read-back */
/* |index> (global index).  HF reference state: X on qubit q iff bit (n-1-q) of hf_init is set
 * (ref:openvqe/ucc_family/get_energy_ucc.py:43 `init`, get_energy_qucc.py:40-45;
 *  ref:openvqe/adapt/fermionic_adapt_vqe.py:183-213 prepare_hf_state) */
int ovqe_init_basis(ovqe_handle h, uint64_t index);
int ovqe_set_state(ovqe_handle h, const double *amps_re_im);       /* 2*2^n_local doubles */

/* The state as the list of its non-zero amplitudes, ascending index (what iterating a myQLM Result of a state-vector job yields,
 * ref:openvqe/adapt/fermionic_adapt_vqe.py:309-328 get_statevector): *count = number of non-zero amplitudes; indices[0..*count) and
 * amps[0..2 * *count) (re, im) are filled when *count <= capacity (else only the count is returned: call again, or ovqe_get_state).
 * *count = -1 on registers below 12 qubits and on shards of a distributed register (use ovqe_get_state there). */
int ovqe_get_support(ovqe_handle h, int64_t capacity, uint64_t *indices, double *amps, int64_t *count);
/* full-statevector read-back (ref:openvqe/adapt/fermionic_adapt_vqe.py:309-328 get_statevector) */
int ovqe_get_state(ovqe_handle h, double *amps_re_im);
int ovqe_get_amplitudes(ovqe_handle h, int64_t count, const uint64_t *local_indices, double *amps_re_im);
/* deterministic synthetic state: amp(i) = scale * (u1(i), u2(i)), u in [-1,1) from a counter-based
 * integer hash of (seed, global index) (bit-reproducible on the host, see openvqe_amd/synth.py);
 * scale normalises the FULL state when norm2_total > 0 is given, else this shard alone.
 * Returns the scale used. */
int ovqe_randomize(ovqe_handle h, uint64_t seed, double norm2_total, double *scale_out);
int ovqe_norm2(ovqe_handle h, double *out); /* sum |a_i|^2 over this shard */

/* ---- unit operations on the resident state (large-n, HBM-bound path) */
/* psi <- exp(-i phi P) psi : the unit of work of build_ucc_ansatz([op], init, n_steps=1)([theta])
 * (third-party; call sites ref:openvqe/ucc_family/get_energy_ucc.py:44,86,
 *  ref:openvqe/adapt/fermionic_adapt_vqe.py:158,302, ref:openvqe/adapt/qubit_adapt_vqe.py:181,303) */
int ovqe_apply_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi);
/* R rotations in order; consecutive rotations sharing an x mask are fused into one sweep */
int ovqe_apply_pauli_rotations(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *phi);
/* one literal gate (ref:openvqe/common_files/circuit.py:13-93 templates); b0 = target/control bit, b1 = CNOT target */
int ovqe_apply_gate(ovqe_handle h, int opcode, int b0, int b1, double angle);
/* Re sum_t coeff[t] <psi|P_t|psi> + constant  (shard-local partial sum when sharded):
 * circ.to_job(job_type="OBS", observable=H) + qpu.submit(job).value
 * (ref:openvqe/ucc_family/get_energy_ucc.py:46-48, get_energy_qucc.py:52-54,
 *  ref:openvqe/adapt/fermionic_adapt_vqe.py:161,237, ref:openvqe/adapt/qubit_adapt_vqe.py:209,267,306) */
int ovqe_expectation(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                     double constant, double *out);
/* sum_t coeff[t] <bra|P_t|ket> with explicit device buffers of 2^n_local amplitudes each
 * (bra/ket NULL = this handle's state).  Used for global-x Hamiltonian terms across shards. */
int ovqe_bilinear(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t T, const uint64_t *x,
                  const uint64_t *z, const double *coeff_re, const double *coeff_im, double *out_re_im);

/* out (+)= sum_t (coeff_re + i coeff_im)[t] P_t ket, explicit device buffers of 2^n_local amplitudes (ket NULL = this
 * handle's state; out != ket).  x masks may carry ONE global (rank) part when ket is the partner shard with that rank
 * difference: sigma = H psi of a sharded register, assembled group by group by the host layer
 * (ref:openvqe/adapt/fermionic_adapt_vqe.py:114 `sig = hamiltonian_sparse.dot(curr_state)`). */
int ovqe_apply_pauli_sum(ovqe_handle h, const void *ket_dev, void *out_dev, int64_t T, const uint64_t *x, const uint64_t *z,
                         const double *coeff_re, const double *coeff_im, int accumulate);
/* out[k] = sum_{j in [off[k], off[k+1])} c_j <bra|P_j|ket> (complex, interleaved re/im) for n_ops operators in one launch;
 * bra / ket as in ovqe_bilinear; all x masks of one call share their global part (the pool gradients of a sharded
 * register: bra = sigma shard, ket = psi shard of the partner; ref:openvqe/adapt/fermionic_adapt_vqe.py:67-73). */
int ovqe_bilinear_batch(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t n_ops, const int64_t *offsets,
                        const uint64_t *x, const uint64_t *z, const double *coeff_re, const double *coeff_im,
                        double *out_re_im);

/* ---- Pauli sums planned once for a shard of the partitioned register ("group terms by x_g ... exchange
 * (read-only) + local partial sums"; the observable of ref:openvqe/ucc_family/get_energy_ucc.py:46-48 and the sigma = H psi of
 * ref:openvqe/adapt/fermionic_adapt_vqe.py:114 on a register that no single device holds).  Masks live in the PHYSICAL index-bit space
 * of the whole register (n_local + n_global bits).  d = x >> n_local names the shard (this shard's index ^ d) a term reads its ket
 * amplitudes from; the host layer (openvqe_amd/distributed.py) receives that shard in chunks of 2^chunk_bits amplitudes and hands
 * every chunk to the calls below.  The plan — d = 0 terms as a tile cover of the shard, the others as LDS-tiled passes over
 * (chunk, own shard) pairs, csrc/sv_cross.hpp — is made once; an evaluation uploads nothing.  *_remote calls only enqueue work on the
 * handle's stream (ovqe_set_stream: order them behind the transfer of the chunk); expect_local / expect_finish synchronise. */
int ovqe_xsum_create(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff_re, const double *coeff_im,
                     int chunk_bits, int32_t *id);
int ovqe_xsum_destroy(ovqe_handle h, int32_t id);
/* the rank differences d != 0 the sum has terms for (ascending) and the passes one chunk of each costs; *count = how many */
int ovqe_xsum_partners(ovqe_handle h, int32_t id, int64_t capacity, uint64_t *d, int64_t *passes, int64_t *count);
/* info[0..10): local x-groups, local terms, local tile sweeps (0 until the first evaluation), local groups outside the cover,
 * partners, remote x-groups, remote terms, remote passes per chunk summed over the partners, tile bits of the passes, 1 when the
 * chunks are too small to tile (streaming kernel) */
int ovqe_xsum_info(ovqe_handle h, int32_t id, int64_t *info, int count);
/* Re <shard| H_0 |shard> of the d = 0 terms (real coefficients only) */
int ovqe_xsum_expect_local(ovqe_handle h, int32_t id, double *out);
/* accumulate <shard| H_d |ket> for chunk `chunk` of the shard of (this shard's index ^ d): ket_chunk = 2^chunk_bits amplitudes on
 * this device */
int ovqe_xsum_expect_remote(ovqe_handle h, int32_t id, uint64_t d, uint64_t chunk, const void *ket_chunk);
/* the accumulated remote contractions (re, im) since the last finish; resets the accumulator */
int ovqe_xsum_expect_finish(ovqe_handle h, int32_t id, double *out_re_im);
/* out = ident * psi + H_0 psi on this shard's state (out: 2^n_local amplitudes, != the state) */
int ovqe_xsum_apply_local(ovqe_handle h, int32_t id, void *out_dev, double ident);
/* out += H_d ket for one received chunk (out: the 2^n_local-amplitude buffer being accumulated) */
int ovqe_xsum_apply_remote(ovqe_handle h, int32_t id, uint64_t d, uint64_t chunk, const void *ket_chunk, void *out_dev);

/* ---- compiled evaluation: E(theta) of a whole ansatz circuit */
/* observable H = constant + sum_t coeff[t] P_t (real coefficients; ref:...get_energy_ucc.py:47) */
int ovqe_set_hamiltonian(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                         double constant);
/* Pauli-rotation program on |hf_index>: rotation r is exp(-i (coeff[r]*theta[pidx[r]] + phi0[r]) P_r)
 * (pidx[r] < 0: constant angle phi0[r]; phi0 may be NULL); K = number of parameters.
 * This is the flattened form of the loop ref:openvqe/ucc_family/get_energy_ucc.py:42-45
 * (and fermionic_adapt_vqe.py:156-159, qubit_adapt_vqe.py:301-304). */
int ovqe_set_program(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *coeff,
                     const double *phi0, const int32_t *pidx, int32_t K, uint64_t hf_index);
/* literal gate program on |hf_index>: gate g has angle ascale[g]*theta[pidx[g]] + aconst[g]
 * (ref:openvqe/ucc_family/get_energy_qucc.py:37-51 + circuit.py:95-106 efficient_fermionic_ansatz).
 * With option "clifford_frame" != 0 the call OVERWRITES the resident state buffer (the Clifford part of the list is
 * executed once on |hf_index> to read its global phase); on any error the handle is left with NO program set. */
int ovqe_set_gate_program(ovqe_handle h, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1,
                          const double *ascale, const double *aconst, const int32_t *pidx, int32_t K,
                          uint64_t hf_index);
/* E(theta): one call == one ucc_action / action_quccsd evaluation
 * (ref:openvqe/ucc_family/get_energy_ucc.py:8-50, get_energy_qucc.py:11-56).  The content of the state buffer after an
 * energy call is unspecified (the fused kernels never touch it, the streaming path may hold real amplitudes there):
 * use ovqe_prepare_state to obtain U(theta)|hf>. */
int ovqe_energy(ovqe_handle h, const double *theta, int32_t K, double *energy);
/* B parameter vectors (row-major B x K) in one launch — one finite-difference gradient of
 * scipy.optimize.minimize(jac=None) (ref:openvqe/ucc_family/get_energy_ucc.py:158-175) is B = K+1 */
int ovqe_energy_batch(ovqe_handle h, int64_t B, const double *theta, int32_t K, double *energies);
/* same with theta (B x K doubles) and energies (B doubles) RESIDENT ON THE DEVICE (e.g. theta batches produced on the GPU, or
 * uploaded once and re-used): nothing crosses PCIe on the fused kernels (n <= 16) and on the batched sector evaluations (real-amplitude
 * programs with sector tables: whole batches per pass of the tables); any other program is served through the host, one
 * evaluation at a time (B x K doubles down, B energies up) */
int ovqe_energy_batch_device(ovqe_handle h, int64_t B, const void *theta_dev, int32_t K, void *energies_dev);
/* run the program and leave U(theta)|hf> in the handle's state buffer
 * (prepare_state_ansatz + get_statevector, ref:openvqe/adapt/fermionic_adapt_vqe.py:273-328) */
int ovqe_prepare_state(ovqe_handle h, const double *theta, int32_t K);

/* ---- ADAPT gradient screen on the resident state psi (uses the stored Hamiltonian):
 * sigma = H psi once, then for pool operator k = sum_{j in [off[k],off[k+1])} c_j P_j
 *   mode FERMIONIC: g_k = 2 Re sum_j c_j <sigma|P_j|psi>   (c complex; A_k anti-Hermitian)
 *   mode QUBIT    : g_k = 2 | sum_j c_j <sigma|P_j|psi> |
 * (ref:openvqe/adapt/fermionic_adapt_vqe.py:41-122, ref:openvqe/adapt/qubit_adapt_vqe.py:126-150,462-471) */
int ovqe_pool_gradients(ovqe_handle h, int64_t n_ops, const int64_t *offsets, const uint64_t *x, const uint64_t *z,
                        const double *coeff_re, const double *coeff_im, int mode, double *grads);
/* psi <- exp(theta * A) psi with A = sum_t (coeff_re+i coeff_im)[t] P_t, exact (scaled Taylor series),
 * the state evolution of the gradient screens
 * (ref:openvqe/adapt/fermionic_adapt_vqe.py:12-38 expm_multiply; qubit_adapt_vqe.py:20-55 expm(-i theta P)) */
int ovqe_apply_exp_pauli_sum(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff_re,
                             const double *coeff_im, double theta);

/* ---- E(theta) and its EXACT gradient dE/dtheta[K] by the adjoint (reverse-mode) method: psi = U(theta)|hf>,
 * lambda = H psi, then one backward pass over the program that un-applies every rotation from both states and reads
 * dE/dtheta_p = sum_{r: pidx_r = p} 2 c_r Im <lambda_r|P_r|psi_r> on the way — about three circuit executions for
 * ALL K derivatives, where the reference's BFGS (jac=None, ref:openvqe/ucc_family/get_energy_ucc.py:158-175) spends K+1
 * evaluations on forward differences (opt-in, because exact derivatives change the
 * optimiser's iterates at the 1e-8 level).  Streaming kernels, any n; the state buffer is left in |hf>.  A real-amplitude
 * program with sector tables (option "sector"; built by the first gradient call when they do not exist yet) runs the whole
 * pass on them: forward circuit,
 * lambda = H psi from the materialised Hamiltonian, backward sweeps over the pair lists. */
int ovqe_energy_gradient(ovqe_handle h, const double *theta, int32_t K, double *energy, double *grad);

/* ---- lowest eigenpair of the stored Hamiltonian (Lanczos on the device, two-pass: tridiagonal matrix, then the Ritz
 * vector by the same recurrence; random start vector from `seed`, so every symmetry sector is reached — the global
 * minimum over the whole register, like column 0 of the reference's dense np.linalg.eigh,
 * ref:openvqe/adapt/fermionic_adapt_vqe.py:474 / qubit_adapt_vqe.py:424-426, which is O(8^n) and infeasible at the
 * H2O size).  Stops when the Lanczos residual estimate < tol * max(1, |lambda|) or after max_iter steps.  The
 * normalised eigenvector is left in the handle's state buffer (ovqe_get_state); *energy includes the constant,
 * *residual = |H y - lambda y| measured afterwards, *iterations = Lanczos steps taken. */
int ovqe_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy, double *residual,
                      int *iterations);
/* ... and of the Hamiltonian RESTRICTED TO THE SUPPORT of the stored program's states (sector tables, option "sector";
 * built by this call when they do not exist yet), inside the BLOCK OF THE REFERENCE DETERMINANT: the determinants that the
 * non-zero matrix elements connect to |hf> are found first and the Lanczos vectors live on them alone, so tables on a superset
 * of the symmetry sector (the one-angle-per-rotation probe of spin-adapted ansaetze lists several particle-number sectors)
 * give the same number as tables on the sector itself.  For a particle-number / spin conserving ansatz on a Hartree-Fock
 * determinant that is the full-CI energy of the determinant's symmetry sector — the `fci` argument the reference's
 * drivers take from PySCF (ref:openvqe/common_files/molecule_factory.py:120-125, `info["FCI"]`), here for active spaces the dense
 * routines cannot reach (N2/cc-pVDZ (10e,12o): 627 264 determinants, 538 M matrix elements).  Two-pass Lanczos on
 * vectors of |support| doubles, H v from the materialised matrix.  The normalised eigenvector is left in the handle's state
 * buffer (zeros outside the support: ovqe_get_state / ovqe_get_support).  OVQE_ERR_STATE when the program has no such tables
 * (not a real-amplitude program, support denser than 1/sector_sparsity, tables beyond sector_max_gb).
 * WITHOUT a stored program the sector is that of the state in the buffer: the closure of its support under the Hamiltonian's
 * x-groups — after ovqe_init_basis(hf) the (N_alpha, N_beta) sector of the reference determinant, with no UCCSD program to name it
 * (the tables are those of the ADAPT screens, option "screen_sector", and are kept for them). */
int ovqe_sector_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy, double *residual,
                             int *iterations);

/* ---- measurement support (bench.py): average device time in ms of `reps` back-to-back launches of
 * one Pauli-rotation sweep, bracketed by HIP events on the handle's stream */
int ovqe_time_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi, int warmup, int reps,
                             double *avg_ms);
/* device time in ms of the most recent ovqe_energy_batch launch (HIP events); 0 for a call that is not timed: a lone evaluation
 * beyond the fused kernels' register sizes, small batches through the mapped buffer */
int ovqe_last_batch_ms(ovqe_handle h, double *ms);

/* amplitudes the last ovqe_pool_gradients call (which = 0) or ovqe_apply_exp_pauli_sum call (which = 1) walked instead of the
 * register ("screen_sparse"): the support of psi, resp. its closure under the operator's x-groups; -1 when the call walked the
 * register (dense state, sharded register, option off).  which = 2: determinants of the symmetry sector whose materialised
 * Hamiltonian produced sigma = H psi of the last ovqe_pool_gradients call (option "screen_sector", default 1: real Hamiltonian,
 * real psi listing at least "screen_sector_min" = 1024 amplitudes; the tables are built once per Hamiltonian on the closure of
 * psi's support under its x-groups), 0 when sigma came from the register / the tile cover.  which = 3: matrix-vector rounds
 * the last ovqe_sector_ground_state took to saturate the block of H connected to the reference determinant (the call
 * returns OVQE_ERR_STATE instead of diagonalising a truncated block when the search does not saturate).  which = 4 / 5:
 * passes over the state (kernel launches that stream the shard) of the last ovqe_apply_pauli_rotations / ovqe_bilinear /
 * ovqe_expectation call, and the bytes those passes move by construction (fused runs and tile covers make both smaller
 * than one sweep per rotation / x-group: what bench.py's `sharded` block reports next to its formula rates).  which = 6: the kernel
 * forms of the sector path that served the handle since the program was set, as bits — 0 / 1 / 2 / 3 forward sweeps on pair words /
 * 64-bit words with rounds / per-wave streams / regular supports by bit arithmetic, 4..7 the backward sweeps of
 * ovqe_energy_gradient in the same order, 8 / 9 first / second form of the pair-table builder (the tests name the geometry behind
 * every form: DESIGN.md section 4) */
int ovqe_last_support(ovqe_handle h, int32_t which, int64_t *support);
/* shape of the compiled program (diagnostics / tests), up to `count` entries of:
 *   [0] ops of the sequential program  [1] Pauli rotations  [2] literal X/H/CNOT ops  [3] streaming sweeps per
 *   evaluation  [4] of those, LDS-tiled multi-op sweeps  [5] ops of the fused-kernel program
 *   [6] support-compacted program: -1 not analysed yet, 0 none, else the size of the reachable support
 *   [7] tile sweeps of the stored Hamiltonian's expectation (0 until first used / when not tiled)  [8] x-groups
 *   that keep their own sweep  [9] (group, pattern) entries  [10] merged terms  [11] pair x term evaluations per tile
 *   [12] 1 when streaming energies of this program keep the state as 2^n real amplitudes
 *   [13..15] support-compacted program: ops, active pairs per evaluation, entries of the restricted Hamiltonian
 *   [16..21] sector path (0 until its tables exist): support size, circuit sweeps, active pairs per evaluation, <H> sweeps
 *   (0: circuit only, <H> by the compact cover), matrix elements of the materialised Hamiltonian (padding included), table
 *   bytes  [22..24] with option "sector_profile" = 1: HIP-event time in microseconds of the circuit sweeps and of the <H>
 *   kernel of the most recent sector evaluation; bytes that kernel reads per evaluation
 *   [25] determinants of the block of H (restricted to the support) that the last ovqe_sector_ground_state diagonalised
 *   [26..27] support-compacted program: colliding lane pairs per evaluation (LDS bank conflicts of the circuit's pair
 *   rotations) with the support numbered in discovery order, and with the numbering in use (option "sparse_renumber")
 *   [28..29] sector path on a REGULAR support (the full coset of the program's Z2 symmetries, e.g. the spin-parity quarter of
 *   the register that the reference's QUCCSD templates populate; option "sector_regular", default 1): slot bits of a circuit
 *   tile — the sweeps then run from bit arithmetic, without pair words — and the number of free (dependent) index bits; 0 else */
int ovqe_program_info(ovqe_handle h, int64_t *info, int count);
/* The stored program as its sequence of Pauli rotations exp(-i (coeff[r] theta[pidx[r]] + phi0[r]) P_r), P_r = (x[r], z[r]) in
 * index-bit space, in execution order (pidx < 0: a constant angle) — for a gate program in Clifford-frame form
 * (ovqe_set_gate_program, option "clifford_frame") the rotations with their CONJUGATED strings, i.e. what the reference's
 * template list (ref:openvqe/common_files/circuit.py:13-106, executed by ref:openvqe/ucc_family/get_energy_qucc.py:47-52) becomes
 * once its Clifford gates are moved to the end; for ovqe_set_program the rotations as given.  *count receives the number of
 * rotations; up to `capacity` of them are written (arrays may be NULL when capacity is 0).  OVQE_ERR_STATE when no program is set
 * or the program still holds literal X / H / CNOT ops (a literal list, an open or forced frame).  Diagnostics / tests: lets a
 * checker evaluate the very sequence the kernels run with an independent simulator. */
int ovqe_get_rotation_program(ovqe_handle h, int64_t capacity, uint64_t *x, uint64_t *z, double *coeff, double *phi0,
                              int32_t *pidx, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* OVQE_SV_H */
