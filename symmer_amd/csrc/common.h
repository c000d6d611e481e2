// common.h — internal declarations shared by the libsymgpu translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/symgpu.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;

namespace symgpu {

// ---- error handling --------------------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define HIP_TRY(expr)                                                         \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return symgpu::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define SG_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != SYMGPU_OK) return _rc; \
    } while (0)

#define SG_REQUIRE(cond, msg)                       \
    do {                                            \
        if (!(cond)) {                              \
            symgpu::set_error("invalid argument: %s (%s)", msg, #cond); \
            return SYMGPU_E_INVALID;                \
        }                                           \
    } while (0)

#define KERNEL_CHECK() HIP_TRY(hipGetLastError())

// Environment switches.  The RUNTIME switches (one table: DESIGN.md, "Environment switches") are read with getenv where they act, on every
// call: each selects a path the library also takes by itself (a size limit, a refused attribute, a time-out), so that the tests can force it
// on inputs of any size.  Everything else — tuning knobs and variants that were measured slower — is compiled out of the default build:
// `make TUNING=1` (-DSYMGPU_TUNING) makes SG_TUNE read the environment again.
#ifdef SYMGPU_TUNING
#define SG_TUNE(name) getenv(name)
#else
#define SG_TUNE(name) (static_cast<const char *>(nullptr))
#endif

// ---- context ---------------------------------------------------------------------------------
struct EmitProbe { const void *key = nullptr; int phase = 0, choice = 0; u64 stamp = 0; float best[2] = {1e30f, 1e30f}; hipEvent_t ev[2] = {nullptr, nullptr}; };
struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // side stream: VALU-bound coefficient kernel overlaps the HBM-bound row stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    u32 *mail_host = nullptr, *mail_dev = nullptr;   // 64 bytes of mapped, coherent host memory: word 0 = sequence number, words 1.. = the few
    u32 mail_seq = 0;                                // status words / counts a call reads back in the middle of its work (read_back_words)
    bool mail_failed = false;
    int num_cu = 256;
    // linear-hash tables (cleanup): 8 byte positions x 256 values x {h1,h2}; reseeded on collision
    u64 *hash_tab = nullptr;       // device, [8][256][2]
    u64 hash_seed = 0;
    std::vector<u64> host_hash_tab;  // host copy of hash_tab (to hash single rows, e.g. a rotation's Q, with the SAME device's tables)
    u64 *xs_pow = nullptr;         // device, [32][64]: columns of M^(2^j), M = the hash's xorshift step (k_hash_rows_long, cleanup.hip)
    // rotation hash join (rotate.hip): persistent open-addressing table of [tag 32 | generation 10 | row index + 1 : 22] entries.
    // An entry of another generation is empty, so the table is cleared once per 1023 rotations instead of once per rotation.
    u64 *rot_table = nullptr;
    size_t rot_table_cap = 0;      // entries (power of two)
    u32 rot_gen = 0;
    u32 *rot_flags = nullptr;      // device u32[4]: [0] = generation in which a duplicate input row was seen
    void *rot_host_cnt = nullptr, *rot_host_cnt_dev = nullptr;   // pinned host copy of a rotation's counts and its device address (rotate.hip)
    // one-launch rotation (rotate_resident.hip): partner notes [generation 10 | row + 1 : 22] of the join (same generation as
    // rot_table), the granules of the in-launch all-gathers + failure words, and the epoch that tags them (1 .. 16382)
    u32 *rot_partner = nullptr;
    size_t rot_partner_cap = 0;
    u64 *res_table = nullptr;      // its join table [canonical-key tag 32 | row + 1 : 32], all-zero between launches
    size_t res_table_cap = 0;
    bool res_dirty = false;        // a launch did not complete: table and notes must be cleared before the next one
    u64 *res_state = nullptr;
    u32 res_epoch = 0;
    u32 *sort_state = nullptr;     // one-launch radix sort (sort.hip): barrier counter, time-out flag, tile histograms
    u32 sort_bar_base = 0;
    bool sort_coop_disabled = false;
    EmitProbe emit_probe[8];          // fused output stage of the cleanup: the output blocks whose block order is being / has been measured (cleanup.hip emit_order_begin)
    u64 emit_stamp = 0;
    u32 *m7_flags = nullptr;       // stream-K commutation kernel (commute_m4r7.hip): per-workgroup words compared with the launch's epoch
    u32 m7_epoch = 0;
    u32 *sort_scan_ticket = nullptr;   // radix sort: "last workgroup finishes the scan" ticket, zero between launches
    bool res_disabled = false;     // a barrier timed out once (workgroups not co-resident): the process keeps to the multi-launch paths
    u32 res_finished_base = 0, res_host_tag = 0;   // one-launch rotation: arrival counter base of the next launch, tag of its report
};
Context &ctx();                            // the calling thread's current device's context
Context *ctx_of_device(int device);
void select_device(int device);            // this thread's current device (no HIP call; require_ctx binds the runtime)
int selected_device();
void forget_bound_device();                // after a library (RCCL) may have changed the thread's HIP device behind our back
// hipFuncSetAttribute acts on the CURRENT device's copy of a kernel, so "set once" means once per device: evaluates `expr` (-> bool) the
// first time the enclosing code runs on a device and remembers the answer for that device (the statics belong to the call site)
#define SG_DEVICE_ONCE(expr)                                                                          \
    ([&]() -> bool {                                                                                  \
        static u32 _done = 0, _ok = 0;                                                                \
        const u32 _bit = 1u << symgpu::ctx().device;                                                  \
        if (!(_done & _bit)) { _done |= _bit; if (expr) _ok |= _bit; }                                \
        return (_ok & _bit) != 0;                                                                     \
    }())
extern i64 g_counters[16];                 // debug counters (symgpu_debug_counter): [1] one-launch rotations, [2] their failures, [3] hipMalloc calls of dev_alloc,
                                           // [7] / [8] payload bytes host -> device / device -> host, [9] / [10] operator uploads / downloads (calls)
inline void count_h2d(size_t bytes) { g_counters[7] += (i64)bytes; }
inline void count_d2h(size_t bytes) { g_counters[8] += (i64)bytes; }
int require_ctx();
// A call that is handed operator handles runs on THEIR device for its duration (and refuses handles of different devices); the
// thread's own selection comes back when the scope ends.
struct DeviceScope {
    int saved = -1;
    bool active = false;
    int enter(const struct ::symgpu_op_s *a, const struct ::symgpu_op_s *b = nullptr, const struct ::symgpu_op_s *c = nullptr);
    ~DeviceScope();
};
#define SG_ENTER(...)                   \
    symgpu::DeviceScope _dev_scope;     \
    SG_TRY(_dev_scope.enter(__VA_ARGS__))
// A fast path gave up in this process (an in-kernel wait timed out because the workgroups were not co-resident — a shared or partitioned
// GPU —, or the runtime refused an LDS attribute) and a slower form has taken over: said ONCE on stderr and kept for symgpu_degraded().
void note_degraded(const char *what);

// per-launch event timing of one kernel class (bench.py roofline leg)
struct ProfScope {
    int cls; bool on;
    hipEvent_t a = nullptr, b = nullptr;
    explicit ProfScope(int kernel_class);
    ~ProfScope();
};

// cached device allocator (hipMalloc is slow; rotations chain many small temporaries)
int dev_alloc(size_t bytes, void **ptr);
void prefault_host(void *dst, size_t bytes);                      // touch the pages of a large D2H destination first
int dev_free(void *ptr);
void dev_cache_release();

// RAII scratch buffer on the cached allocator
struct Scratch {
    void *p = nullptr;
    Scratch() {}
    ~Scratch() { if (p) dev_free(p); }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    int alloc(size_t bytes) { if (p) { dev_free(p); p = nullptr; } return dev_alloc(bytes ? bytes : 16, &p); }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace symgpu

// ---- device-resident operator ------------------------------------------------------------------
struct symgpu_op_s {
    int device = 0;          // the device whose memory holds the operator (set by symgpu_op_alloc)
    u64 *rows = nullptr;     // [capacity][2*Wq] row-major packed
    double *coeff = nullptr; // [capacity][2] or null
    i64 T = 0, capacity = 0;
    int Wq = 0;
    // cached word-major copy of rows[0..T) (layout.hip), padded to wm_pad terms; dropped whenever rows/T change
    u64 *wm = nullptr;
    i64 wm_pad = 0, wm_T = -1;
    // 1 = known to hold no two equal rows (result of a cleanup, or of a rotation of such an operator); 0 = unknown.
    // Reset by op_invalidate, i.e. whenever rows change.  The odd-k Clifford rotation needs to know (rotate.hip).
    int dup_free = 0;
    // cached linear row hashes h1 of rows[0..T) under hash seed `hash_seed` (0 = none): a rotation hands them on to its result
    // for free (h(P ^ Q) = h(P) ^ h(Q)), so a chain of rotations hashes the operator once.  Dropped by op_invalidate.
    u64 *hash = nullptr;
    u64 hash_seed = 0;
    // cached bit-major copy of rows[0..T) for the Four-Russians commutation kernel (commute_m4r.hip): bt[c][jw], bt_pad words per
    // bit-row; valid while bt_T == T.  An adjacency matrix computed slab by slab transposes its right operand once.
    u64 *bt = nullptr;
    i64 bt_pad = 0, bt_T = -1;
    // cached Y counts |x & z| of rows[0..T) (product.hip: the coefficient expansion adds 3 (Y_i + Y_o) to the phase bytes);
    // valid while yc_T == T.  Dropped by op_invalidate.
    int *yc = nullptr;
    i64 yc_T = -1;
    // first[t] (optional; results of the *_indexed cleanups): the input index under which output term t was filed — its first occurrence:
    // the position of a plain cleanup's input row, (o << 32) | i of a product's pair.  Dropped by op_invalidate.
    u64 *first = nullptr;
};

namespace symgpu {

// layout.hip — word-major ("bit-sliced column") copy: out[w][t], t padded to Tpad (zero filled)
int to_wordmajor(const u64 *rows, i64 T, int W, u64 *out, i64 Tpad, hipStream_t st = nullptr);
// cached word-major copy of a whole operator (padded to a multiple of `mult` terms); built on the main stream
int op_wordmajor(symgpu_op_s *op, i64 mult, const u64 **out, i64 *pad);
// cached Y counts of a whole operator (int per term); built on the main stream
int op_ycount(symgpu_op_s *op, const int **out);
void op_invalidate(symgpu_op_s *op);

// the n_a + n_b <= 8 device words a[0..n_a) ++ b[0..n_b) to the host, behind everything queued on the stream so far (context.hip)
int read_back_words(const u32 *a, int n_a, const u32 *b, int n_b, u32 *host_out, const u32 *c1 = nullptr);
struct ReadBack { int n; u32 seq; u32 plain[8]; };
int read_back_post(const u32 *a, int n_a, const u32 *b, int n_b, ReadBack *rb, const u32 *c1 = nullptr /* one more word */);   // ... more work may be queued before the wait
int read_back_wait(ReadBack *rb, u32 *host_out);

// sort.hip
int exclusive_scan_u32(const u32 *in, u32 *out, i64 n, u32 *total_dev /* may be null: device u32 */);
// bitmaps of up to POPC_SCAN_SMALL_MAX 64-bit words: out[w] = set bits in the words before w, *total_dev = all of them, one launch (n32: see sort.hip)
constexpr i64 POPC_SCAN_SMALL_MAX = (i64)1 << 13;
int popc_scan_small(const u64 *bits, i64 n, i64 n32, u32 *out, u32 *total_dev);
// first_hist (optional, device, 256 * ceil(n / SORT_TILE) u32): the TILE-major ([tile][256]) tile histograms of the FIRST pass
// (hist[tile * 256 + digit] = keys of tile `tile` whose bits [begin_bit, begin_bit + 8) equal `digit`), formed by a kernel of the
// caller's that reads the keys anyway; the buffer then serves the later passes
constexpr int SORT_TILE = 4096;
int radix_sort_pairs_u64_u32(u64 *keys, u32 *vals, u64 *keys_tmp, u32 *vals_tmp, i64 n, int begin_bit, int end_bit,
                             bool *result_in_tmp, u32 *first_hist = nullptr);
int radix_sort_keys_u64(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, u32 *first_hist = nullptr);
// the same as one persistent launch (up to 2^19 keys); *done = false: not applicable, use the multi-launch form
int radix_sort_keys_u64_coop(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done);
int radix_sort_keys_u64_small(u64 *keys, u64 *keys_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done);   // the one launch where it pays (<= ~2e5 keys)
int radix_sort_pairs_u64_u32_coop(u64 *keys, u32 *vals, u64 *keys_tmp, u32 *vals_tmp, i64 n, int begin_bit, int end_bit, bool *result_in_tmp, bool *done);
const u32 *radix_sort_coop_flag();
void radix_sort_coop_note(u32 flag, bool *timed_out);
int radix_sort_coop_check(bool *timed_out);     // after a stream synchronisation: did a one-launch sort give up (output invalid)?

// commute.hip
// b_owner (may be null): the operator B's rows belong to (all of them), so that per-operand layouts can be cached on it
int commutes_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits, symgpu_op_s *b_owner = nullptr);
int ycount_dev(const u64 *rows, i64 T, int Wq, int *out);
// commute_m4r.hip — the same contract on the Four-Russians kernel (LDS tables)
int commutes_m4r_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits, symgpu_op_s *b_owner = nullptr);
bool commutes_m4r_worthwhile(i64 N, i64 M);
int bits_to_bytes_dev(const u64 *bits, i64 stride_words, i64 N, i64 M, uint8_t *out);   // bit-packed rows -> np.bool_ [N][M], any M, any alignment
// commute_m4r7.hip — the same product with two 7-bit tables per step (called by commutes_m4r_dev, which owns the operand preparation of B)
int commutes_m4r7_launch(const u64 *A, i64 N, i64 M, int Wq, const u64 *bt_p, i64 Mw_pad, int R, bool bytes, void *dst, i64 stride);

// product.hip
// ---- shared device helpers ----------------------------------------------------------------------
// exact phase application: multiply (re, im) by i^e
__device__ __forceinline__ void apply_phase(double re, double im, int e, double &ore, double &oim) {
    const bool swap = e & 1;
    double a = swap ? im : re, b = swap ? re : im;
    // e=0: ( re,  im)  e=1: (-im,  re)  e=2: (-re, -im)  e=3: ( im, -re)
    const bool neg_a = (e == 1) || (e == 2), neg_b = (e == 2) || (e == 3);
    ore = neg_a ? -a : a;
    oim = neg_b ? -b : b;
}
// plain IEEE complex product (no FMA contraction: matches numpy for exactly representable inputs) times i^e
__device__ __forceinline__ void pair_coefficient(double ar, double ai, double br, double bi, int e, double &ore, double &oim) {
    const double re = __dsub_rn(__dmul_rn(ar, br), __dmul_rn(ai, bi));
    const double im = __dadd_rn(__dmul_rn(ar, bi), __dmul_rn(ai, br));
    apply_phase(re, im, e, ore, oim);
}

// packed pair key of the fused product + cleanup: [hash: 64-F bits][e: 2][o: bo][i: bi], F = bi + bo + 2 (cleanup.hip)
struct PairKeyArgs {
    const u64 *hI, *hO;     // per-operand row hashes
    u64 *keys;              // [No*Ni] out, index o*Ni + i  (squared mode: compacted, see below)
    int bi, bo;
    i64 o_base;             // absolute index of the launch's first outer row
    // squared mode (P * P, cleanup.hip): only the pairs with i >= o get a key (the twin (o, i) of an off-diagonal pair is the
    // same row with the same or the opposite coefficient), compacted in pair-index order: slot(o, i) = o*Ni - o(o-1)/2 + (i - o)
    int squared = 0;
    // round 6, ebytes != null: instead of the 8-byte key ONE byte per pair at the key's index, e | (i == o) << 2 — all that the marking of
    // the single terms reads when the pairs that share a key are found from the operand hash tables (pair_dups.hip); `keys` is not written
    unsigned char *ebytes = nullptr;
};

int mul_coeff_dev(const u64 *inner, const double *ci, i64 Ni, const u64 *outer, const double *co, i64 o_begin, i64 o_end,
                  int Wq, int inner_is_left, double *out_coeff);
int mul_rows_dev(const u64 *inner, i64 Ni, const u64 *outer, i64 o_begin, i64 o_end, int Wq, u64 *out_rows);
// packed (hash | phase exponent | o | i) keys of all pairs, for the fused product + cleanup
int mul_keys_dev(const u64 *inner, i64 Ni, const u64 *outer, i64 No, int Wq, int inner_is_left, PairKeyArgs ka);

// wide.hip — few pairs of very long rows: the word axis is the parallel one
bool wide_pairs_worthwhile(i64 Ni, i64 No, int Wq);
int wide_commutes_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits);
int wide_mul_coeff_dev(const u64 *inner, const double *ci, i64 Ni, const u64 *outer, const double *co, i64 o_begin, i64 o_end, int Wq,
                       int inner_is_left, double *out_coeff, const PairKeyArgs *keys);

// cleanup.hip
extern i64 g_hash_reseeds;                                      // row-hash collisions that forced a reseed (never seen outside the tests)
int ensure_hash_tables(u64 seed);
int hash_rows(const u64 *rows, i64 T, int W, u64 *out1);       // h1 of every row (current tables)
u64 host_row_hash(const u64 *row, int W);                       // the same hash on the host
int cleanup_core(const u64 *rows, const double *coeff, i64 T, int W,          // plain mode (pair mode if inner != null)
                 const u64 *inner, i64 Ni, const u64 *outer, i64 No,
                 double thr, int use_thr, symgpu_op_t *out, int Wq_out,
                 const double *ci = nullptr, const double *co = nullptr, int inner_is_left = 1,    // pair mode: operand coefficients
                 bool want_first = false);                                                          // the result carries symgpu_op_s::first

// pair_dups.hip — the pairs of a product whose 64-bit key another pair shares, found without sorting the keys
bool pair_dups_fits(i64 Ni, i64 No, bool squared, i64 Tk, int *B_out, int *sb_out = nullptr);      // host-side: would pair_dups_dev take this product?
int pair_dups_dev(const u64 *hI, i64 Ni, const u64 *hO, i64 No, bool squared, i64 Tk, u64 *flags, u32 *giveup, bool *applies);

// gf2.hip
int rref_dev(u64 *rows, i64 R, i64 Wc, i64 *xor_count, i64 *pivots_host);

}  // namespace symgpu
