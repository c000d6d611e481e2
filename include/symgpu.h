/* symgpu.h — C ABI of libsymgpu.so: the MI355X (gfx950) implementation of Symmer's symplectic hot path.
 *
 * The reference (UCL-CCS/symmer) is pure Python and has NO FFI for this path; the seam it offers is the
 * set of NumPy-level functions below (file:line relative to the reference checkout).  Each entry point
 * here replaces one of them; `INTEGRATION.md` shows the ctypes stub a Symmer maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns SYMGPU_OK (0) or a negative error code and
 *    never throws or exits; `symgpu_last_error()` gives the text of the last failure on this thread.
 *  - host buffers are caller-owned, C-contiguous, and are not retained past the call.
 *  - packed symplectic row: 2*Wq uint64 words, X words first then Z words, Wq = max(1, ceil(n/64));
 *    bit j (LSB = 0) of word w <-> qubit 64*w + j; padding bits MUST be zero.
 *    == np.packbits(block, axis=1, bitorder='little') viewed as '<u8' after zero-padding to 64*Wq columns.
 *  - GF(2) matrices: R rows of Wc uint64 words, same bit rule; "leftmost column" = lowest set bit of the
 *    first non-zero word.
 *  - coefficients: complex128 as interleaved double[2] (re, im).
 *  - one context per DEVICE; calls on a device are serialised on its HIP stream; a handle is single-owner and belongs to
 *    the device it was created on.  Multi-GPU = one process per device (symgpu_init + symgpu_comm_*, RCCL over xGMI) or one
 *    process driving several devices (symgpu_init_all + symgpu_set_device + symgpu_comm_init_all).
 */
#ifndef SYMGPU_H
#define SYMGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYMGPU_OK 0
#define SYMGPU_E_INVALID (-1)   /* bad argument (null pointer, negative size, Wq mismatch, T >= 2^32 for cleanup ...) */
#define SYMGPU_E_HIP (-2)       /* a HIP runtime call failed; see symgpu_last_error() */
#define SYMGPU_E_NOMEM (-3)     /* device allocation failed */
#define SYMGPU_E_CAPACITY (-4)  /* caller-supplied output capacity too small; *n_out holds the required row count */
#define SYMGPU_E_NODEVICE (-5)  /* no HIP device / symgpu_init not called */
#define SYMGPU_E_COLLISION (-6) /* row-hash collision survived every reseed (never observed; exactness guard) */
#define SYMGPU_E_RCCL (-7)      /* RCCL could not be loaded or a collective failed */

typedef struct symgpu_op_s *symgpu_op_t; /* device-resident operator: packed rows (+ optional coefficients) */

/* ---- context -------------------------------------------------------------------------------- */
#define SYMGPU_MAX_DEVICES 16
int symgpu_init(int device);              /* create the device's context (stream, allocator) if needed and make it the calling thread's current device */
/* Single-process multi-device mode ("one host process ... driving <= 8 devices — no MPI launcher needed"): contexts for devices
 * 0 .. n-1 (n <= 0: all visible), peer access between them.  A host thread has a CURRENT device (symgpu_set_device; initially the first
 * device initialised): uploads, allocations and every call without a handle go there.  A call that takes handles runs on the handles'
 * device whatever the current one is, and refuses handles of different devices (SYMGPU_E_INVALID) — symgpu_op_copy_rows is the one call
 * that crosses devices (peer copy over xGMI).  Each device has its own stream; calls on different devices overlap, so one thread can
 * launch a block of an all-pairs kernel on every device and then collect the results. */
int symgpu_init_all(int n_devices);
int symgpu_set_device(int device);
int symgpu_current_device(int *device);
int symgpu_n_initialised(int *n);          /* number of devices with a context in this process */
int symgpu_shutdown(void);                 /* all devices */
const char *symgpu_last_error(void);
int symgpu_device_count(int *n);          /* does not initialise a device */
int symgpu_sync(void);                    /* wait for the library stream */
int symgpu_device_sync(void);             /* hipDeviceSynchronize on the library's device (all streams) */
int symgpu_device_name(char *buf, int len);
int symgpu_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* HIP-event timer on the library stream (used by bench.py for per-kernel durations) */
int symgpu_timer_start(void);
int symgpu_timer_stop(float *ms);
/* per-launch HIP-event timing of the dominant kernel of a class (0 = product row stream k_mul_rows, 1 = commutation k_commutes,
 * 2 = GF(2) sweep k_sweep_m4r (main launches), 3 = cleanup output stage k_emit_fused (k_emit_stream with SYMGPU_EMIT_FUSED=0), 4 = one-launch rotation k_rot_resident,
 * 5 = register-resident run of Clifford rotations k_cchain_reg): enable, run, then read {launch count, total ms}. */
#define SYMGPU_PROF_CLASSES 6
int symgpu_prof_enable(int kernel_class, int on);
int symgpu_prof_read(int kernel_class, int64_t *n_launches, double *total_ms);
/* statistics for tests: which = 0: number of row-hash collisions that forced the cleanup to reseed its hash and retry (the exactness
 * guard behind symplectic_cleanup, operators/utils.py:230-279; expected 0 outside the tests, which weaken the hash on purpose);
 * 1: rotations completed by the one-launch LDS-resident kernel; 2: calls of that kernel that reported a failed row verification or a
 * barrier time-out (the multi-launch path then recomputes); 3: device allocations that were not served from the allocator's arena;
 * 4-6: host nanoseconds spent by the one-launch rotation in preparation, in the launch call and waiting for the kernel's status word;
 * 7 / 8: payload bytes copied host -> device / device -> host by this library (operators, coefficients, index arrays, result matrices; not the
 * few-byte counts and flags a call reads back); 9 / 10: operator uploads / downloads (calls that moved a whole operator or its coefficients).
 * The drop-in classes keep their operands on the device between calls; the tests assert through 7-10 that a multi-step workflow
 * (symmer/projection/base.py:44-124, rotate -> project -> cleanup) moves its operator once in and once out. */
int symgpu_debug_counter(int which, int64_t *value);
/* Fast paths that gave up in this process and were replaced by a slower, equally exact form — the one-launch rotation, the one-launch
 * radix sort, the fused selector launch of the GF(2) elimination: their in-kernel waits assume co-resident workgroups and are bounded, so
 * a shared or partitioned GPU turns them off.  Each is announced once on stderr; this returns the list ("" = nothing degraded). */
int symgpu_degraded(char *buf, int len);
/* tuning aid (library built with `make TUNING=1`): with SYMGPU_RES_TRACE=1 every workgroup of the one-launch rotation kernel stamps the 100 MHz wall clock at its phase
 * boundaries; this copies the stamps of the last traced launch, 16 words per workgroup */
int symgpu_debug_rotation_trace(uint64_t *out, int max_workgroups, int *n_workgroups);

/* measured on-box HBM ceilings for the roofline: one-shot 16-byte-per-thread fill (GB/s written) and copy (GB/s read+written)
 * over scratch buffers of `bytes` each (best of 3). */
int symgpu_membw_probe(int64_t bytes, double *fill_GBps, double *copy_GBps);

/* ---- device-resident operators -------------------------------------------------------------- */
int symgpu_op_upload(const uint64_t *rows, const double *coeff /* may be NULL */, int64_t T, int Wq, symgpu_op_t *out);
int symgpu_op_alloc(int64_t capacity_rows, int Wq, int with_coeff, symgpu_op_t *out);
int symgpu_op_download(symgpu_op_t op, uint64_t *rows, double *coeff /* may be NULL */, int64_t capacity_rows);
int symgpu_op_info(symgpu_op_t op, int64_t *T, int *Wq, int64_t *capacity_rows);
int symgpu_op_free(symgpu_op_t op);
int symgpu_op_set_rows(symgpu_op_t op, int64_t T);   /* trim (T <= capacity), e.g. after an all-gather with padding */
/* host rows (+ coefficients if both sides have them) -> rows [row_offset, row_offset + count) of an existing operator
 * (within its capacity; T grows to cover them).  Used by the host-staged all-gather. */
int symgpu_op_write(symgpu_op_t op, int64_t row_offset, const uint64_t *rows, const double *coeff /* may be NULL */, int64_t count);
/* device-to-device: rows (and coefficients, if both have them) src[src_offset .. +count) -> dst[dst_offset ..); stream ordered */
int symgpu_op_copy_rows(symgpu_op_t dst, int64_t dst_offset, symgpu_op_t src, int64_t src_offset, int64_t count);
int symgpu_op_random(int64_t T, int n_qubits, double density, uint64_t seed, symgpu_op_t *out); /* synthetic input, generated on device */
/* Handle-level primitives behind the device-resident drop-in classes (no reference counterpart: the reference keeps NumPy arrays):
 *  clone       rows (+ coefficients) copied device to device into a new handle
 *  set_coeff   the T coefficients replaced from a host array (allocated if the operator had none); rows and their cached layouts stay
 *  scale       in place c <- (conjugate_first ? conj(c) : c) * (re + i im)   (multiply_by_constant base.py:750-762, dagger :1366-1376)
 *  ycount      PauliwordOp.Y_count (base.py:604-615) of a resident operator -> int64[T] on the host
 *  upload_bool / download_bool   the reference layout itself (np.bool_ [T][2n], X columns then Z columns, base.py:42-74) <-> packed rows,
 *              packed / unpacked ON THE DEVICE: np.packbits on the host runs at ~1.5 GB/s of bools, PCIe + a ballot kernel at >10 GB/s */
int symgpu_op_clone(symgpu_op_t in, symgpu_op_t *out);
int symgpu_op_set_coeff(symgpu_op_t op, const double *coeff_host);
int symgpu_op_scale(symgpu_op_t op, double re, double im, int conjugate_first);
int symgpu_op_ycount(symgpu_op_t op, int64_t *out_host);
int symgpu_op_upload_bool(const uint8_t *symp /* [T][2n] */, const double *coeff /* may be NULL */, int64_t T, int n_qubits, symgpu_op_t *out);
int symgpu_op_download_bool(symgpu_op_t op, int n_qubits, uint8_t *symp_out /* [capacity_rows][2n] */, int64_t capacity_rows);
/* XOR-fold of all packed rows (2*Wq words) and plain sum of coefficients: size-independent checksums */
int symgpu_op_checksum(symgpu_op_t op, uint64_t *xor_words /* [2*Wq] */, double *coeff_sum /* [2] */);
/* number of set bits in all packed rows: sum_{i,o} |a_i ^ b_o| of a product slab follows from the operands' bit-column counts in O(N + M) */
int symgpu_op_popcount(symgpu_op_t op, uint64_t *sum);

/* ---- a2: PauliwordOp.Y_count  (symmer/operators/base.py:604-615) ----------------------------- */
int symgpu_ycount(const uint64_t *rows, int64_t T, int Wq, int64_t *out);

/* ---- a6: commutes_termwise / adjacency_matrix (base.py:938-971, 1054-1062; utils.py:9-78) ------
 * out[i*M + j] = 1 iff A[i] commutes with B[j] (np.bool_ layout of the reference's return value). */
int symgpu_commutes(const uint64_t *A, int64_t N, const uint64_t *B, int64_t M, int Wq, uint8_t *out);
/* device-resident; out_dev is a DEVICE pointer obtained from symgpu_dev_alloc (N*M bytes) */
int symgpu_commutes_dev(symgpu_op_t A, int64_t a_begin, int64_t a_end, symgpu_op_t B, uint8_t *out_dev);
/* bit-packed variant: out_bits[i*ceil(M/64) + j/64] bit (j%64); 1/8 byte per pair */
int symgpu_commutes_bits_dev(symgpu_op_t A, int64_t a_begin, int64_t a_end, symgpu_op_t B, uint64_t *out_bits_dev);
int symgpu_dev_alloc(int64_t bytes, void **ptr);
int symgpu_dev_free(void *ptr);
int symgpu_dev_download(const void *dev, void *host, int64_t bytes);
int symgpu_dev_upload(void *dev, const void *host, int64_t bytes);
int symgpu_dev_checksum_u8(const uint8_t *dev, int64_t n, uint64_t *sum); /* sum of bytes (number of commuting pairs) */
int symgpu_dev_popcount_u64(const uint64_t *dev, int64_t n_words, uint64_t *sum);

/* ---- a3/a4: all-pairs product  (base.py:764-794 `_multiply_by_operator`, :821-859 `__mul__`) ----
 * Output row o*Ni + i = inner[i] xor outer[o]; coefficient = c_inner[i]*c_outer[o]*i^e with
 * e = (3(Y_i+Y_o) + Y_out + 2|x_left & z_right|) mod 4, left = inner if inner_is_left else outer
 * (the reference's dagger-swap for N < M, base.py:847-849, folded into one exponent).  No cleanup. */
int symgpu_mul_allpairs(const uint64_t *inner, const double *ci, int64_t Ni,
                        const uint64_t *outer, const double *co, int64_t No, int Wq, int inner_is_left,
                        uint64_t *out_rows, double *out_coeff);
/* device-resident slab: outer rows [o_begin, o_end) -> out (capacity >= (o_end-o_begin)*Ni rows) */
int symgpu_mul_allpairs_dev(symgpu_op_t inner, symgpu_op_t outer, int64_t o_begin, int64_t o_end,
                            int inner_is_left, symgpu_op_t out);

/* ---- a5: symplectic_cleanup / PauliwordOp.cleanup (utils.py:230-279, base.py:617-638) ------------
 * Merge duplicate rows (sequential sum in input order), keep |c| > thr (strict) if use_thr, output in
 * first-occurrence order.  W = words per row (2*Wq).  n_out always receives the row count. */
int symgpu_cleanup(const uint64_t *rows, const double *coeff, int64_t T, int W, double thr, int use_thr,
                   uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out);
int symgpu_cleanup_dev(symgpu_op_t in, double thr, int use_thr, symgpu_op_t *out);
/* product + cleanup fused: product rows are never materialised (linear row hash of pairs) */
int symgpu_mul_cleanup(const uint64_t *inner, const double *ci, int64_t Ni,
                       const uint64_t *outer, const double *co, int64_t No, int Wq, int inner_is_left,
                       double thr, int use_thr,
                       uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out);
int symgpu_mul_cleanup_dev(symgpu_op_t inner, symgpu_op_t outer, int inner_is_left, double thr, int use_thr,
                           symgpu_op_t *out);
/* The same two device calls with the FIRST-OCCURRENCE INDEX of every output term kept on the result (read it with symgpu_op_first_index):
 * the position of the first input row of a plain cleanup, (o << 32) | i of a product's first pair (o: outer index, i: inner index; the
 * reference's pair index is o * Ni + i, base.py:783-792).  A caller that cleans ONE product in parts — symmer_amd/parallel.py, the
 * hash-partitioned multi-GPU cleanup — merges the parts in the reference's first-occurrence order with it (utils.py:271). */
int symgpu_cleanup_indexed_dev(symgpu_op_t in, double thr, int use_thr, symgpu_op_t *out);
int symgpu_mul_cleanup_indexed_dev(symgpu_op_t inner, symgpu_op_t outer, int inner_is_left, double thr, int use_thr, symgpu_op_t *out);
int symgpu_op_first_index(symgpu_op_t op, uint64_t *first_host, int64_t capacity_rows);
/* Device side of the hash-partitioned multi-GPU cleanup (symmer_amd/parallel.py): out = op[idx] (rows + coefficients); the indices of an
 * indexed product of two gathered sub-operands rewritten to pair indices of the complete operands (o_global * Ni_global + i_global); an index
 * array attached to an operator (the shares received from other ranks); and the merge of indexed operators: concatenated, ordered by index
 * (keys below 2^key_bits; 0 = 64), then — do_cleanup — rows that several parts share merged at their smallest index with the cleanup's
 * semantics and threshold.  The result carries its indices again (ascending). */
int symgpu_op_gather(symgpu_op_t op, const int64_t *idx_host, int64_t n, symgpu_op_t *out);
int symgpu_part_global_index(symgpu_op_t part, const int64_t *inner_idx_host, int64_t n_inner, const int64_t *outer_idx_host, int64_t n_outer,
                             int64_t Ni_global);
int symgpu_op_set_first_index(symgpu_op_t op, const uint64_t *first_host);
int symgpu_merge_indexed_dev(const symgpu_op_t *parts, int n_parts, int key_bits, int do_cleanup, double thr, int use_thr, symgpu_op_t *out);

/* ---- a7: _rotate_by_single_Pword (base.py:1090-1161), one fused pass ------------------------------
 * clifford_k < 0: non-Clifford: cleanup([commuting, cos*anticommuting, -i*sin*(anticommuting*Q)], thr)
 * clifford_k >= 0 (= round(2*angle/pi)): [rotated anticommuting rows, commuting rows], no merge;
 *   odd k: c*i^e*(-i); k in {2,3}: negated (not reduced mod 4, base.py:1148).
 * *all_commute = 1 (and out untouched / *out = NULL) when every row commutes with Q. */
int symgpu_rotate_single(const uint64_t *rows, const double *coeff, int64_t N, int Wq, const uint64_t *q_row,
                         double cos_t, double sin_t, int clifford_k, double thr,
                         uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out, int *all_commute);
int symgpu_rotate_single_dev(symgpu_op_t in, const uint64_t *q_row_host, double cos_t, double sin_t, int clifford_k,
                             double thr, symgpu_op_t *out, int *all_commute);
/* The same call, also returning the result's term count (0 when *out == NULL): the drop-in class needs it for every rotation
 * (base.py:1159-1161: a rotation that leaves no term returns 0 * I) and saves a symgpu_op_info round trip per call. */
int symgpu_rotate_single_dev_n(symgpu_op_t in, const uint64_t *q_row_host, double cos_t, double sin_t, int clifford_k,
                               double thr, symgpu_op_t *out, int *all_commute, int64_t *n_out);

/* A run of K Clifford rotations (perform_rotations, base.py:1163-1186, on pi/2-multiples; CircuitSymmerlator) of a CLEAN
 * operator: `in` must come from a cleanup (no duplicate rows, every |c| > 1e-15) — then every step is a stable partition
 * [anticommuting | commuting] + row ^= Q + exact phase, exactly what the reference's rotation followed by cleanup() returns, the
 * term count never changes and nothing has to come back to the host between the steps.  Up to 1,536 rows the whole run is ONE
 * single-workgroup launch; larger operators (<= 2^22 rows) run the per-rotation kernels back to back without a read-back.
 * q_rows: K packed rows; ks: clifford_k per rotation, 0..3 as for symgpu_rotate_single (k = round(2*angle/pi) mapped as
 * symmer_amd.kernels.rotation_args). */
int symgpu_rotate_clifford_chain_dev(symgpu_op_t in, const uint64_t *q_rows_host, const int *ks_host, int64_t K, symgpu_op_t *out);
/* perform_rotations (base.py:1163-1186) on a device-resident operator in ONE call: rotations r = 0 .. K-1 (q_rows[r][2*Wq], cos_t[r], sin_t[r],
 * ks[r] = clifford_k as for symgpu_rotate_single_dev) applied in order, each followed by the reference's cleanup() — which is the identity once the
 * operator is clean (`clean` != 0 on entry says it already is), so it runs once; runs of Clifford rotations of a clean operator go through the chain
 * entry point.  acted[r] (may be NULL; zeroed by the caller) is set where a single rotation changed the operator.  The call always processes all K
 * rotations (*n_done == K on success): an operator that loses all its terms is taken through the reference's alternation between "no terms" and
 * 0 * I (cleanup() of an empty operator, base.py:631-632, utils.py:275-278) INSIDE the call — the caller must not apply it again — and *clean_out
 * reflects the state after the last rotation.  *out = NULL: nothing changed, keep using `in` (which is never freed here). */
int symgpu_perform_rotations_dev(symgpu_op_t in, const uint64_t *q_rows_host, const double *cos_t, const double *sin_t, const int *ks_host, int64_t K,
                                 double thr, int clean, symgpu_op_t *out, uint8_t *acted, int64_t *n_done, int *clean_out);

/* ---- a8: _rref_binary (utils.py:292-315): in place, no row swaps, leftmost pivot, eliminate above and
 * below.  xor_count (may be NULL) = sum_i |update_set_i| as the reference loop performs them.
 * pivots (may be NULL) = pivot column per row or -1. */
int symgpu_rref(uint64_t *rows, int64_t R, int64_t Wc, int64_t *xor_count, int64_t *pivots);
int symgpu_rref_dev(uint64_t *rows_dev, int64_t R, int64_t Wc, int64_t *xor_count, int64_t *pivots_host);

/* ---- a9: IndependentOp.symmetry_generators (independent_op.py:124-126) ----------------------------
 * H: M packed rows over n qubits.  out: generators as packed rows in the reference's order; *k = count.
 * capacity in rows (2n always suffices). */
int symgpu_symmetry_kernel(const uint64_t *H, int64_t M, int n_qubits, int Wq,
                           uint64_t *out, int64_t capacity, int64_t *k, int64_t *xor_count);
int symgpu_symmetry_kernel_dev(symgpu_op_t H, int n_qubits, uint64_t *out, int64_t capacity, int64_t *k,
                               int64_t *xor_count);

/* ---- f2: generator routines on packed rows, device side (csrc/genrec.hip) ------------------------------------------------
 * symgpu_op_gf2_rank   rank over GF(2) of the operator's symplectic rows = number of non-zero rows of _rref_binary(symp_matrix)
 *   (utils.py:292-315); check_independent (utils.py:504-519) is `rank == T`.
 * symgpu_generators_dev   PauliwordOp.generators (base.py:1436-1456): the non-zero rows of _rref_binary(symp_matrix), in row order, with
 *   coefficient 1 — a new resident operator.
 * symgpu_generator_reconstruction_dev   PauliwordOp.generator_reconstruction (base.py:523-560): reduced = cref_binary(vstack([G, M]))
 *   (utils.py:349-359); recon_host int64 [T][g] = reduced[g:, :g] (the reference returns .astype(int)), mask_host uint8 [T] =
 *   all(~reduced[g:, g:], axis=1).  g = G's terms (1 <= g <= 2n), T = M's terms (>= 1).  The transposed stack is built bit-packed on the
 *   device, reduced by the blocked elimination and read out in pivot order: no one-byte-per-bit matrix on either side. */
int symgpu_op_gf2_rank(symgpu_op_t op, int64_t *rank);
int symgpu_generators_dev(symgpu_op_t op, symgpu_op_t *out);
int symgpu_generator_reconstruction_dev(symgpu_op_t G, symgpu_op_t M, int n_qubits, int64_t *recon_host, uint8_t *mask_host);

/* ---- f3 / f4: the callers next to the hot path, device side -----------------------------------------------------------
 * symgpu_project_dev   S3Projection._perform_projection (symmer/projection/base.py:44-84) on an operator that has already been taken
 *   through the stabiliser rotations: terms that anticommute with any of the k fixed (single-qubit) stabiliser rows vanish; the others
 *   are multiplied by (-1)^|row & neg_mask| (neg_mask: the symplectic positions of the stabilisers with eigenvalue -1, i.e. the product
 *   of eigenvalues over the occupied stabilised positions, :68-71); the qubits not listed in keep_qubits (ascending) are deleted and equal
 *   terms merged with the cleanup's semantics (:82).  stab_rows [k][2*Wq], neg_mask [2*Wq], keep_qubits [n_keep >= 1]: host arrays.
 * symgpu_noncontextual_dev   PauliwordOp.is_noncontextual (base.py:1074-1088, check_adjmat_noncontextual utils.py:567-589): 1 iff the
 *   terms that do not commute with every term split into disjoint cliques of the commutation graph.
 * symgpu_state_inner_dev   QuantumState bra * ket (base.py:1808-1815): sum over the basis rows present in both CLEANED states of
 *   c_a * c_b, added in the order of a's rows (pass the state with fewer terms as a, as the reference does).  out: double[2]. */
int symgpu_project_dev(symgpu_op_t op, const uint64_t *stab_rows, int k, const uint64_t *neg_mask, const int *keep_qubits, int n_keep,
                       int n_qubits, double thr, int use_thr, symgpu_op_t *out, int64_t *n_survived /* may be NULL: terms that commute with every stabiliser */);
int symgpu_noncontextual_dev(symgpu_op_t op, int *is_noncontextual);
int symgpu_state_inner_dev(symgpu_op_t a, symgpu_op_t b, double *out);

/* ---- e: multi-GPU (one process per GPU; RCCL over xGMI) ------------------------------------------- */
#define SYMGPU_UNIQUE_ID_BYTES 128
int symgpu_comm_available(void);                                                /* librccl loadable? (no device, no collective) */
int symgpu_comm_unique_id(uint8_t id[SYMGPU_UNIQUE_ID_BYTES]);                 /* rank 0 */
int symgpu_comm_init(const uint8_t id[SYMGPU_UNIQUE_ID_BYTES], int rank, int nranks);
int symgpu_comm_destroy(void);
/* A caller whose watchdog gave up on a symgpu_comm_init that has not returned calls this: should ncclCommInitRank come back later,
 * its communicator is destroyed instead of installed (the ranks have agreed on the host-staged data plane by then). */
int symgpu_comm_abandon(void);
/* all-gather equal-sized shards of packed rows (+coefficients if both have them) into `full`
 * (capacity >= nranks * shard rows); rank r's rows land at [r*T_shard, (r+1)*T_shard). */
int symgpu_comm_allgather_op(symgpu_op_t shard, symgpu_op_t full);
int symgpu_comm_barrier(void);
/* The same all-gather for ONE process that drives n devices (symgpu_init_all): one communicator per device (ncclCommInitAll); shards[d] and
 * fulls[d] live on device d, every shard has the same capacity Ts, device d's rows land at [d * Ts, (d + 1) * Ts) of every full operator.
 * The n all-gathers are enqueued from the calling thread as one RCCL group (ncclGroupStart / ncclGroupEnd), each on its device's stream. */
int symgpu_comm_init_all(int n_devices);
int symgpu_comm_allgather_ops(const symgpu_op_t *shards, const symgpu_op_t *fulls, int n);

#ifdef __cplusplus
}
#endif
#endif /* SYMGPU_H */
