/* generated from include/ovqe_sv.h by tools/make_cdef.py: do not edit.  For cffi: ffi.cdef(this file). */
enum {
    OVQE_OK = 0,
    OVQE_ERR_INVALID = -1,
    OVQE_ERR_NO_DEVICE = -2,
    OVQE_ERR_HIP = -3,
    OVQE_ERR_ALLOC = -4,
    OVQE_ERR_STATE = -5,
    OVQE_GATE_X = 0,
    OVQE_GATE_H = 1,
    OVQE_GATE_RX = 2,
    OVQE_GATE_RY = 3,
    OVQE_GATE_RZ = 4,
    OVQE_GATE_CNOT = 5,
    OVQE_GRAD_FERMIONIC = 0,
    OVQE_GRAD_QUBIT = 1,
};
typedef struct ovqe_sv *ovqe_handle;
int ovqe_version(void);
const char *ovqe_last_error(ovqe_handle h);
int ovqe_device_count(int *count);
int ovqe_create(int n_qubits, int device, ovqe_handle *out);
int ovqe_create_shard(int n_local, int n_global, uint64_t shard_index, int device, ovqe_handle *out);
int ovqe_create_view(int n_qubits, int device, void *dev_ptr, ovqe_handle *out);
int ovqe_destroy(ovqe_handle h);
int ovqe_set_stream(ovqe_handle h, void *hip_stream);
int ovqe_set_option(ovqe_handle h, const char *name, int64_t value);
int ovqe_state_ptr(ovqe_handle h, void **dev_ptr);
int ovqe_adopt_state(ovqe_handle h, void *dev_ptr);
int ovqe_init_basis(ovqe_handle h, uint64_t index);
int ovqe_set_state(ovqe_handle h, const double *amps_re_im);
int ovqe_get_support(ovqe_handle h, int64_t capacity, uint64_t *indices, double *amps, int64_t *count);
int ovqe_get_state(ovqe_handle h, double *amps_re_im);
int ovqe_get_amplitudes(ovqe_handle h, int64_t count, const uint64_t *local_indices, double *amps_re_im);
int ovqe_randomize(ovqe_handle h, uint64_t seed, double norm2_total, double *scale_out);
int ovqe_norm2(ovqe_handle h, double *out);
int ovqe_apply_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi);
int ovqe_apply_pauli_rotations(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *phi);
int ovqe_apply_gate(ovqe_handle h, int opcode, int b0, int b1, double angle);
int ovqe_expectation(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                     double constant, double *out);
int ovqe_bilinear(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t T, const uint64_t *x,
                  const uint64_t *z, const double *coeff_re, const double *coeff_im, double *out_re_im);
int ovqe_apply_pauli_sum(ovqe_handle h, const void *ket_dev, void *out_dev, int64_t T, const uint64_t *x, const uint64_t *z,
                         const double *coeff_re, const double *coeff_im, int accumulate);
int ovqe_bilinear_batch(ovqe_handle h, const void *bra_dev, const void *ket_dev, int64_t n_ops, const int64_t *offsets,
                        const uint64_t *x, const uint64_t *z, const double *coeff_re, const double *coeff_im,
                        double *out_re_im);
int ovqe_xsum_create(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff_re, const double *coeff_im,
                     int chunk_bits, int32_t *id);
int ovqe_xsum_destroy(ovqe_handle h, int32_t id);
int ovqe_xsum_partners(ovqe_handle h, int32_t id, int64_t capacity, uint64_t *d, int64_t *passes, int64_t *count);
int ovqe_xsum_info(ovqe_handle h, int32_t id, int64_t *info, int count);
int ovqe_xsum_expect_local(ovqe_handle h, int32_t id, double *out);
int ovqe_xsum_expect_remote(ovqe_handle h, int32_t id, uint64_t d, uint64_t chunk, const void *ket_chunk);
int ovqe_xsum_expect_finish(ovqe_handle h, int32_t id, double *out_re_im);
int ovqe_xsum_apply_local(ovqe_handle h, int32_t id, void *out_dev, double ident);
int ovqe_xsum_apply_remote(ovqe_handle h, int32_t id, uint64_t d, uint64_t chunk, const void *ket_chunk, void *out_dev);
int ovqe_set_hamiltonian(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff,
                         double constant);
int ovqe_set_program(ovqe_handle h, int64_t R, const uint64_t *x, const uint64_t *z, const double *coeff,
                     const double *phi0, const int32_t *pidx, int32_t K, uint64_t hf_index);
int ovqe_set_gate_program(ovqe_handle h, int64_t G, const int32_t *opcode, const int32_t *b0, const int32_t *b1,
                          const double *ascale, const double *aconst, const int32_t *pidx, int32_t K,
                          uint64_t hf_index);
int ovqe_energy(ovqe_handle h, const double *theta, int32_t K, double *energy);
int ovqe_energy_batch(ovqe_handle h, int64_t B, const double *theta, int32_t K, double *energies);
int ovqe_energy_batch_device(ovqe_handle h, int64_t B, const void *theta_dev, int32_t K, void *energies_dev);
int ovqe_prepare_state(ovqe_handle h, const double *theta, int32_t K);
int ovqe_pool_gradients(ovqe_handle h, int64_t n_ops, const int64_t *offsets, const uint64_t *x, const uint64_t *z,
                        const double *coeff_re, const double *coeff_im, int mode, double *grads);
int ovqe_apply_exp_pauli_sum(ovqe_handle h, int64_t T, const uint64_t *x, const uint64_t *z, const double *coeff_re,
                             const double *coeff_im, double theta);
int ovqe_energy_gradient(ovqe_handle h, const double *theta, int32_t K, double *energy, double *grad);
int ovqe_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy, double *residual,
                      int *iterations);
int ovqe_sector_ground_state(ovqe_handle h, double tol, int max_iter, uint64_t seed, double *energy, double *residual,
                             int *iterations);
int ovqe_time_pauli_rotation(ovqe_handle h, uint64_t x, uint64_t z, double phi, int warmup, int reps,
                             double *avg_ms);
int ovqe_last_batch_ms(ovqe_handle h, double *ms);
int ovqe_last_support(ovqe_handle h, int32_t which, int64_t *support);
int ovqe_program_info(ovqe_handle h, int64_t *info, int count);
int ovqe_get_rotation_program(ovqe_handle h, int64_t capacity, uint64_t *x, uint64_t *z, double *coeff, double *phi0,
                              int32_t *pidx, int64_t *count);
