// rotate_resident.hip — a single-Pauli rotation (reference: PauliwordOp._rotate_by_single_Pword, symmer/operators/base.py:1090-1161)
// as ONE persistent launch.
//
// The multi-launch paths of rotate.hip (analyze | match | scan | write) are bound by their three kernel boundaries, by the second
// pass over the rows and by the host round trips between them, not by bytes: 40 us of kernels for 8.5 us of traffic at 10^5 terms of
// 1,000 qubits.  Here the operator is spread over the chip instead: one workgroup per CU, each owning a contiguous block of
// ceil(T / G) rows that it reads from HBM ONCE into its LDS (256 CUs x <= 150 KiB = 38 MB of operator; BASELINE cfg2 is 25.6 MB),
// and everything that the kernel boundaries used to order is ordered inside the launch by two all-gathers of 8-byte granules
// {tag, counts} (cdna_hip_programming.md Guideline 16, form R2: the data is the flag, agent-scope relaxed stores and loads, no fence
// because no plain-stored payload crosses workgroups):
//
//   A   rows -> LDS, with the flags and phase exponents of every row formed on the way in registers (one 16-byte chunk per lane, DPP
//       lane exchange as in product.hip's row stream; rows that are not a power-of-two number of chunks: from LDS afterwards);
//       non-Clifford: every anticommuting row enters the join table with ONE compare-and-swap under its CANONICAL key
//       min(h, h ^ h(Q)) — a row P_k and the row P_k ^ Q it would merge with share that key, so whoever of the two comes second
//       finds the other in the slot, notes it in LDS and tells the first through partner[] (agent-scope store).  Table and notes
//       are all-zero between launches: every claimed slot and every note read is zeroed again by its owner after all-gather #1
//   g1  all-gather #1: kept commuting rows per workgroup (and: every partner note is in place)
//   B   final coefficients (cos c_t + (-i sin) i^e' c_partner, or the new row's (-i sin) i^e c_t), classes, ranks inside the block
//   g2  all-gather #2: kept anticommuting / new rows per workgroup; the commuting rows are written while it is in flight
//   C   rows (LDS -> HBM, 16 bytes per lane), coefficients and handed-on hashes to their final slots; counts to pinned host memory
//
// Output order, sums and thresholds are those of rotate.hip's hash-join path (commuting | cos * anticommuting (+ partner) | new rows,
// strict |c| > thr; Clifford: rotated anticommuting | commuting), bit for bit — tests/test_gpu_parity.py runs both.
// Exactness does not rest on the hash: the second row of every pair is compared with its partner chunk by chunk (row ^ Q against
// the partner's row in HBM); a mismatch, a third row under one canonical key, or an all-gather that does not complete (workgroups
// not co-resident) makes the call report failure and the caller takes the multi-launch path.
// Preconditions (else *done = 0): the operator is known to be duplicate free (symgpu_op_s::dup_free), carries its row hashes
// (non-Clifford), has rows of <= 128 words (4,096 qubits), and the per-row state of its block fits the LDS (the rows themselves may stay
// in memory: res_layout's hbm form).
#include "common.h"
#include "rotate_common.h"
#include <time.h>
#include <stdlib.h>

namespace symgpu {

static inline i64 host_ns() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (i64)ts.tv_sec * 1000000000LL + ts.tv_nsec; }

typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int RES_THREADS = 1024;
constexpr int RES_MAX_WG = 256;                   // granules are swept by ONE wavefront, four per lane
constexpr u32 RES_SPIN_LIMIT = 1u << 20;          // ~1 s of polling: the workgroups are not co-resident (another process on the GPU)
constexpr size_t RES_LDS_MAX = 160 * 1024;
constexpr int RES_MIN_ROWS = 64;                  // rows per workgroup below which more workgroups only lengthen the all-gathers
constexpr int RES_LD_UNROLL = 8;                  // 16-byte loads in flight per lane (cfg2: 6,256 chunks per block = one round of 8,192)
constexpr u32 RES_NO_SLOT = 0xFFFFFFFFu;
// class byte of a row: bits 0-2 = kept commuting row / kept anticommuting row / kept new row (what is written out); bit 3 = the row has
// a partner (its coefficient in LDS is final); join state in s_ps: bit 4 = it claimed slot s_ps, bit 5 = it found its partner s_ps,
// bit 6 = its first probe met the occupant {s_posn : s_ps} (resolved after the loads)
enum { CL_C = 1, CL_A = 2, CL_N = 4, CL_MATCHED = 8, CL_CLAIMED = 16, CL_SECOND = 32, CL_PENDING = 64 };

// LDS of one workgroup: the rows' 16-byte chunks (minus the first `nreg` x 1,024, which stay in registers), then 26 bytes per row:
// coefficient (16), one word that is first the row's join state (claimed slot / partner / occupant) and later its rank (4), the rank
// of its new row (4), info and class bytes.  Hashes are read from HBM where they are needed (twice, coalesced).
struct ResLayout { int rows, coef, ps, posn, info, cls, q, wtot, misc, total, lds_chunks; };
constexpr int RES_MAX_W = 128;                   // words per row (4,096 qubits; round 5: 64)
struct QArgW { u64 w[RES_MAX_W]; };
// hbm: the rows are NOT kept on the chip (round 6: operators beyond the 38 MB of LDS + registers): only the 26 bytes per row stay in LDS, the
// rows are read a second time — from the Infinity Cache, mostly — when they are written out
__host__ __device__ inline ResLayout res_layout(int R, int Wq, int nreg, int hbm = 0) {
    ResLayout L;
    int o = 0;
    L.lds_chunks = hbm ? 0 : R * Wq - nreg * 1024;
    if (L.lds_chunks < 0) L.lds_chunks = 0;
    L.rows = o; o += L.lds_chunks * 16;
    L.coef = o; o += R * 16;
    L.ps = o; o += R * 4;
    L.posn = o; o += R * 4;
    L.info = o; o += R;
    L.cls = o; o += R;
    o = (o + 15) & ~15;
    L.q = o; o += RES_MAX_W * 8;
    L.wtot = o; o += 256 * 8;
    L.misc = o; o += 32 * 4;
    L.total = o;
    return L;
}

struct ResArgs {
    const u32x4 *rows; const double *coeff; const u64 *hin;
    i64 T; int Wq, R, GA, nreg; u32 yq;          // nreg: 1,024-chunk rounds of the row load that stay in registers (0 or 2)
    int hbm;                                      // 1: rows are not kept on the chip (res_layout)
    u32x4 *out_rows; double *out_coeff; u64 *out_hash;
    double cos_t, sin_t, thr; int k;
    u64 hq;
    u64 *slots; u32 mask; u32 *partner;           // join table [canonical-key tag 32 | row + 1 : 32] and partner notes (row + 1), zero between launches
    u64 *gran1, *gran2; u32 *fail;                // fail[0] / fail[1] = epoch of a failed verification / of a time-out
    u32 *finished; u32 finish_target;             // fail[2]: workgroups that have left, counted over all launches; the one that reaches the target reports
    u32 epoch;
    u64 *trace;                                   // [G][16] wall-clock stamps of the phases (SYMGPU_RES_TRACE=1), else null
    int inject;                                   // tests: the last workgroup leaves at once without a word (SYMGPU_RES_INJECT=1)
    u64 *host_words; u32 *host_late; u32 host_tag; u32 *published;   // pinned host memory: the report (res_report) and the late-failure word
    QArgW q;
};

enum { M_OK = 1, M_FAIL = 2, M_NC = 3, M_NA = 4, M_NN = 5, M_NANTI = 6, M_PREF_C = 8, M_TOT_C = 9, M_PREF_A = 10, M_TOT_A = 11, M_PREF_N = 12,
       M_TOT_N = 13, M_TOT_ANTI = 14 };

__device__ __forceinline__ u64 ag_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ag_load32(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store32(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wavefront re-reads all G granules until every tag is `tag`; on success the three count fields (16, 16 and 15 bits) are summed
// over the workgroups before `w` (pref) and over all of them (tot); `flagged`: some workgroup set bit 47 (RES_GRAN_FAIL: it failed a
// verification or gave up).  Returns false after RES_SPIN_LIMIT sweeps.
constexpr u64 RES_GRAN_FAIL = 1ULL << 47;
__device__ __forceinline__ bool ag_sweep(const u64 *gran, int G, u32 tag, int w, int lane, u32 (&pref)[3], u32 (&tot)[3], bool &flagged) {
    u64 v[4];
    for (u32 spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = lane + 64 * j;
            v[j] = idx < G ? ag_load(gran + idx) : ((u64)tag << 48);
            ok &= (u32)(v[j] >> 48) == tag;
        }
        if (__ballot(ok) == ~0ULL) break;
        if (spins >= RES_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) { pref[f] = 0; tot[f] = 0; }
    bool fl = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = lane + 64 * j;
        if (idx < G) {
            fl |= (v[j] & RES_GRAN_FAIL) != 0;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const u32 x = (u32)(v[j] >> (16 * f)) & (f == 2 ? 0x7FFFu : 0xFFFFu);
                tot[f] += x;
                if (idx < w) pref[f] += x;
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 3; ++f)
        for (int off = 32; off > 0; off >>= 1) { pref[f] += (u32)__shfl_xor((int)pref[f], off); tot[f] += (u32)__shfl_xor((int)tot[f], off); }
    flagged = __ballot(fl) != 0ULL;
    return true;
}

// The counts of a launch in pinned host memory: two 8-byte words that carry the call's tag — the data is the flag, the host waits
// until both show it.  [tag 16 | nC 22 | nA 22], [tag 16 | code 4 | nN 22 | nAnti 22]
__device__ __forceinline__ void res_report(const ResArgs &a, u32 code, u32 nC, u32 nA, u32 nN, u32 nAnti) {
    __hip_atomic_store(&a.host_words[0], ((u64)a.host_tag << 48) | ((u64)nC << 22) | nA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&a.host_words[1], ((u64)a.host_tag << 48) | ((u64)code << 44) | ((u64)nN << 22) | nAnti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A workgroup leaves (done or timed out).  Success is reported EARLY, by one workgroup as soon as the second all-gather has told it
// the counts (k_rot_resident): the host prepares and enqueues whatever comes next while the rows are still on their way out.  Failures
// are reported by the LAST workgroup to leave — it sees every failure word written before the others' arrivals; should one appear
// after success has been reported (a time-out behind a completed all-gather: not reachable by construction) it goes to the `late`
// word, which fails the next call loudly.
__device__ __forceinline__ void res_leave(const ResArgs &a) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u32 prev = atomicAdd(a.finished, 1u);
    if (prev + 1u == a.finish_target) {
        u32 code = 0;
        if (ag_load32(&a.fail[0]) == a.epoch) code = 2;
        if (ag_load32(&a.fail[1]) == a.epoch) code = 3;
        if (code) {
            if (ag_load32(a.published) == a.epoch) __hip_atomic_store(a.host_late, ((u32)a.epoch << 4) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else res_report(a, code, 0, 0, 0, 1);
        }
    }
}

// rows leave with non-temporal stores; write-through (sc0 sc1) stores measured the same kernel duration (27.5 us)
__device__ __forceinline__ void row_store(u32x4 v, u32x4 *p) { __builtin_nontemporal_store(v, p); }
#define RES_STAMP(i) do { if (a.trace && tid == 0) a.trace[(size_t)w * 16 + (i)] = wall_clock64(); } while (0)

// |c| > thr as NumPy's abs(complex) decides it (hypot), without the hypot for every coefficient that has a component above thr
__device__ __forceinline__ bool res_keep(double re, double im, double thr) {
    return (fabs(re) > thr || fabs(im) > thr) ? true : hypot(re, im) > thr;
}

// flag and the two phase exponents of a row from its counts (rotate.hip: k_rot_analyze): bit 0 = anticommutes with Q, bits 1-2 = the
// exponent e of P * Q, bits 3-4 = the exponent e' of (P ^ Q) * Q, i.e. the e of the row's partner
__device__ __forceinline__ uint8_t res_info(u32 anti, u32 fp, u32 yp, u32 yout, u32 yq) {
    const u32 e = (3u * (yp + yq) + yout + 2u * fp) & 3u;
    const u32 ep = (3u * (yout + yq) + yp + 2u * (fp ^ (yq & 1u))) & 3u;
    return (uint8_t)((anti & 1u) | (e << 1) | (ep << 3));
}

// MODE 0: non-Clifford (hash join); MODE 1: Clifford (k = clifford_k of rotation_args, 0..3)
// WQ: 16-byte chunks per row if that is a power of two <= 64 (analysis in registers while the rows stream in), else 0 (from LDS)
template <int MODE, int WQ>
__global__ __launch_bounds__(RES_THREADS) void k_rot_resident(const ResArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nreg = WQ > 0 ? a.nreg : 0;                                      // (rows of a generic length are analysed from LDS: all of them live there)
    const bool hbm = a.hbm != 0;                                               // (block-uniform) rows read again from memory where they are needed
    const ResLayout L = res_layout(a.R, a.Wq, nreg, hbm ? 1 : 0);
    u32x4 *s_rows = reinterpret_cast<u32x4 *>(smem + L.rows);
    f64x2 *s_coef = reinterpret_cast<f64x2 *>(smem + L.coef);
    u32 *s_ps = reinterpret_cast<u32 *>(smem + L.ps);                          // join state, later: rank of the row in its class
    u32 *s_posn = reinterpret_cast<u32 *>(smem + L.posn);                      // (high word of a pending row's occupant), later: rank of the new row
    uint8_t *s_info = smem + L.info;
    uint8_t *s_cls = smem + L.cls;
    u64 *s_q = reinterpret_cast<u64 *>(smem + L.q);
    u64 *s_wtot = reinterpret_cast<u64 *>(smem + L.wtot);
    u32 *s_misc = reinterpret_cast<u32 *>(smem + L.misc);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Wq = WQ > 0 ? WQ : a.Wq, W = 2 * Wq;
    const int G = gridDim.x, w = blockIdx.x;
    if (a.inject && w == G - 1) return;
    RES_STAMP(0);
    const i64 row0 = (i64)w * a.R;
    const int Rw = (int)(a.T - row0 < (i64)a.R ? a.T - row0 : (i64)a.R);
    const int nchunk = Rw * Wq;
    const int reg_chunks = nreg * RES_THREADS;                                 // chunks [0, reg_chunks) of the block live in keep0 / keep1
    const u32 tag1 = 2 * a.epoch, tag2 = 2 * a.epoch + 1;
    const f64x2 *coeff2 = reinterpret_cast<const f64x2 *>(a.coeff);
    const u32 yq = a.yq;

    if (tid < W) s_q[tid] = a.q.w[tid];
    if (tid < 32) s_misc[tid] = 0;
    __syncthreads();
    const u32x4 *sq4 = reinterpret_cast<const u32x4 *>(s_q);
    u32x4 keep0 = (u32x4)(0u), keep1 = (u32x4)(0u);
    // chunk `it * 1024 + tid` of the block, from the registers or from LDS
    const u32x4 *const rows_blk = a.rows + row0 * Wq;
#define RES_CHUNK(it, i) (hbm ? rows_blk[(i)] : (((it) < nreg) ? ((it) == 0 ? keep0 : keep1) : s_rows[(i) - reg_chunks]))
    // ---- A1: the block's rows and coefficients: HBM -> registers / LDS, read once; flags and phase exponents on the way ---------
    {
        const u32x4 *src = a.rows + row0 * Wq;
        if (WQ == 0 && hbm)                                                    // (nothing to do with the rows here: A2 analyses them from memory)
            for (int r = tid; r < Rw; r += RES_THREADS) s_coef[r] = coeff2[row0 + r];
        for (int i0 = 0; i0 < ((WQ == 0 && hbm) ? 0 : nchunk); i0 += RES_LD_UNROLL * RES_THREADS) {
            u32x4 v[RES_LD_UNROLL];
#pragma unroll
            for (int j = 0; j < RES_LD_UNROLL; ++j) {
                const int i = i0 + j * RES_THREADS + tid;
                v[j] = i < nchunk ? (hbm ? src[i] : __builtin_nontemporal_load(src + i)) : (u32x4)(0u);   // (hbm: the rows are read again: no streaming hint)
            }
            if (i0 == 0)
                for (int r = tid; r < Rw; r += RES_THREADS) s_coef[r] = coeff2[row0 + r];
            u64 hrow[RES_LD_UNROLL], casold[RES_LD_UNROLL];
            u32 caspos[RES_LD_UNROLL];
            if constexpr (WQ > 0) {
                // a row is an aligned group of WQ lanes (RES_THREADS is a multiple of WQ): X words in its lower, Z words in its upper half
                const int c = tid & (WQ - 1);
                // non-Clifford: the chunk-0 lane of an anticommuting row issues the row's compare-and-swap on the join table as soon as
                // the flag is known; the answers (~2 us each) come back while the remaining chunks are analysed and stored
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < RES_LD_UNROLL; ++j) {
                        const int i = i0 + j * RES_THREADS + tid;
                        hrow[j] = (c == 0 && i < nchunk) ? a.hin[row0 + i / WQ] : 0ULL;
                        casold[j] = 0; caspos[j] = RES_NO_SLOT;
                    }
                }
#pragma unroll
                for (int j = 0; j < RES_LD_UNROLL; ++j) {
                    const int i = i0 + j * RES_THREADS + tid;
                    const u32x4 x = v[j];
                    u32 par, ye;                                               // par: |x & zq| + |z & xq| (+ |x & zq| << 16);  ye: Y_P | Y_out << 16
                    if constexpr (WQ == 1) {
                        const u32x4 qv = sq4[0];
                        const u32 f = __popc(x.x & qv.z) + __popc(x.y & qv.w);
                        par = f + __popc(x.z & qv.x) + __popc(x.w & qv.y) + (f << 16);
                        ye = (__popc(x.x & x.z) + __popc(x.y & x.w)) | ((__popc((x.x ^ qv.x) & (x.z ^ qv.z)) + __popc((x.y ^ qv.y) & (x.w ^ qv.w))) << 16);
                    } else {
                        const u32x4 qs = sq4[c], qo = sq4[c ^ (WQ / 2)];
                        const bool xhalf = c < WQ / 2;
                        const u32 p = __popc(x.x & qo.x) + __popc(x.y & qo.y) + __popc(x.z & qo.z) + __popc(x.w & qo.w);
                        const u32x4 o = {rot_other_half<WQ>(x.x), rot_other_half<WQ>(x.y), rot_other_half<WQ>(x.z), rot_other_half<WQ>(x.w)};
                        const u32 yp = __popc(x.x & o.x) + __popc(x.y & o.y) + __popc(x.z & o.z) + __popc(x.w & o.w);
                        const u32 yo = __popc((x.x ^ qs.x) & (o.x ^ qo.x)) + __popc((x.y ^ qs.y) & (o.y ^ qo.y)) + __popc((x.z ^ qs.z) & (o.z ^ qo.z)) +
                                       __popc((x.w ^ qs.w) & (o.w ^ qo.w));
                        par = rot_row_sum<WQ>(p + (xhalf ? (p << 16) : 0u));
                        ye = rot_row_sum<WQ>(xhalf ? (yp | (yo << 16)) : 0u);
                    }
                    if (c == 0 && i < nchunk) {
                        s_info[i / WQ] = res_info(par & 1u, (par >> 16) & 1u, ye & 0xFFFFu, ye >> 16, yq);
                        if (MODE == 0 && (par & 1u)) {
                            const u64 h = hrow[j], hp = h ^ a.hq, ck = h < hp ? h : hp;
                            caspos[j] = (u32)mix64(ck) & a.mask;
                            casold[j] = atomicCAS(reinterpret_cast<unsigned long long *>(&a.slots[caspos[j]]), 0ULL,
                                                  (unsigned long long)((ck & 0xFFFFFFFF00000000ULL) | (u64)(row0 + i / WQ + 1)));
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < RES_LD_UNROLL; ++j) {
                const int i = i0 + j * RES_THREADS + tid;
                if (i0 == 0 && j < 2 && j < nreg) { if (j == 0) keep0 = v[j]; else keep1 = v[j]; }
                else if (i < nchunk && !hbm) s_rows[i - reg_chunks] = v[j];
            }
            if constexpr (WQ > 0 && MODE == 0) {
                // What the first probes found (only now: the empty statement keeps the compiler from testing each answer right behind
                // its compare-and-swap, which would serialise the eight round trips): 0 = the slot is this row's; an occupant is
                // left in {s_posn : s_ps} for A3, which continues the walk from there.
                asm volatile("" : "+v"(casold[0]), "+v"(casold[1]), "+v"(casold[2]), "+v"(casold[3]), "+v"(casold[4]), "+v"(casold[5]), "+v"(casold[6]), "+v"(casold[7]));
                static_assert(RES_LD_UNROLL == 8, "the statement above names eight answers");
                const int c = tid & (WQ - 1);
#pragma unroll
                for (int j = 0; j < RES_LD_UNROLL; ++j) {
                    const int i = i0 + j * RES_THREADS + tid;
                    if (c == 0 && i < nchunk) {
                        const bool anti = caspos[j] != RES_NO_SLOT, pending = anti && casold[j] != 0;
                        s_ps[i / WQ] = pending ? (u32)casold[j] : caspos[j];
                        s_posn[i / WQ] = (u32)(casold[j] >> 32);
                        s_cls[i / WQ] = (uint8_t)(pending ? CL_PENDING : (anti ? CL_CLAIMED : 0));
                    }
                }
            }
        }
    }
    __syncthreads();
    if constexpr (WQ == 0) {
        // ---- A2: flags and phase exponents from LDS, GA lanes per row (row lengths that are not a power-of-two number of chunks) ---
        const int GA = a.GA, g = tid & (GA - 1), rsub = tid / GA, rpp = RES_THREADS / GA;
        const u64 *rows64 = hbm ? reinterpret_cast<const u64 *>(rows_blk) : reinterpret_cast<const u64 *>(s_rows);
        for (int r0 = 0; r0 < Rw; r0 += rpp) {
            const int r = r0 + rsub;
            u32 pf = 0, yy = 0;                     // pf: parity of |x & zq| + |z & xq| (bit 0) and of |x & zq| (bit 1); yy: Y_P | Y_out << 16
            if (r < Rw) {
                const u64 *row = rows64 + (size_t)r * W;
                u64 par = 0, flip = 0;
                for (int ww = g; ww < Wq; ww += GA) {
                    const u64 x = row[ww], z = row[Wq + ww], xq = s_q[ww], zq = s_q[Wq + ww];
                    par ^= (x & zq) ^ (z & xq);
                    flip ^= x & zq;
                    yy += (u32)__popcll(x & z) + ((u32)__popcll((x ^ xq) & (z ^ zq)) << 16);
                }
                pf = ((u32)__popcll(par) & 1u) | (((u32)__popcll(flip) & 1u) << 1);
            }
            for (int off = GA >> 1; off > 0; off >>= 1) { pf ^= (u32)__shfl_xor((int)pf, off); yy += (u32)__shfl_xor((int)yy, off); }
            if (g == 0 && r < Rw) s_info[r] = res_info(pf & 1u, (pf >> 1) & 1u, yy & 0xFFFFu, yy >> 16, yq);
        }
        __syncthreads();
    }
    RES_STAMP(1);

    u32 prefC = 0, totC = 0;
    if (MODE == 0) {
        // ---- A3: commuting rows are classified; every anticommuting row meets its partner, if it has one, in the join table ----
        u32 nC = 0;
        bool bad = false;
        for (int r = tid; r < Rw; r += RES_THREADS) {
            const uint8_t info = s_info[r];
            uint8_t cls = 0;
            u32 ps = 0;
            if (!(info & 1)) {
                const f64x2 c = s_coef[r];
                if (res_keep(c.x, c.y, a.thr)) { cls = CL_C; ++nC; }
            } else if (WQ > 0 && (s_cls[r] & CL_CLAIMED)) {
                cls = CL_CLAIMED; ps = s_ps[r];                                // claimed by the probe issued from the load loop
            } else {
                // generic row lengths: the whole walk; otherwise: the first probe met the occupant left in {s_posn : s_ps} — the walk goes on there
                const i64 t = row0 + r;
                const u64 h = a.hin[t], hp = h ^ a.hq, ck = h < hp ? h : hp;
                const u64 entry = (ck & 0xFFFFFFFF00000000ULL) | (u64)(t + 1);
                u32 pos = (u32)mix64(ck) & a.mask;
                bool have_old = WQ > 0;
                for (;;) {
                    const u64 old = have_old ? (((u64)s_posn[r] << 32) | s_ps[r])
                                             : atomicCAS(reinterpret_cast<unsigned long long *>(&a.slots[pos]), 0ULL, (unsigned long long)entry);
                    have_old = false;
                    if (old == 0) { cls = CL_CLAIMED; ps = pos; break; }       // first of its key: a partner, if any, will leave a note
                    if ((old >> 32) == (ck >> 32)) {
                        const i64 o = (i64)(old & 0xFFFFFFFFULL) - 1;
                        const u64 ho = a.hin[o];
                        if (ho == hp) {                                        // the row this one merges with (verified below)
                            cls = CL_SECOND; ps = (u32)o;
                            ag_store32(&a.partner[o], (u32)(t + 1));
                            break;
                        }
                        if (ho == h) { bad = true; break; }                    // two rows with one hash: not for this path
                    }
                    pos = (pos + 1) & a.mask;
                }
            }
            s_ps[r] = ps;
            s_cls[r] = cls;
        }
        for (int off = 32; off > 0; off >>= 1) nC += (u32)__shfl_xor((int)nC, off);
        if (lane == 0 && nC) atomicAdd(&s_misc[M_NC], nC);
        if (bad) s_misc[M_FAIL] = 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every partner note has left this wavefront
        __syncthreads();
        if (tid == 0) ag_store(&a.gran1[w], ((u64)tag1 << 48) | s_misc[M_NC]);
        RES_STAMP(2);
        // verification of the pairs found from THIS block (while the all-gather is in flight): row ^ Q against the partner's row
        {
            int r = tid / Wq, c = tid - r * Wq;
            const int dr = RES_THREADS / Wq, dc = RES_THREADS - dr * Wq;
            bool mism = false;
            for (int it = 0, i = tid; i < nchunk; ++it, i += RES_THREADS) {
                if (s_cls[r] & CL_SECOND) {
                    const u32x4 mine = RES_CHUNK(it, i) ^ sq4[c], theirs = a.rows[(i64)s_ps[r] * Wq + c];
                    mism |= (mine.x != theirs.x) | (mine.y != theirs.y) | (mine.z != theirs.z) | (mine.w != theirs.w);
                }
                r += dr; c += dc;
                if (c >= Wq) { c -= Wq; ++r; }
            }
            if (mism) s_misc[M_FAIL] = 1;
        }
        // ---- g1 ----------------------------------------------------------------------------------------------------------------
        if (wave == 0) {
            u32 pref[3], tot[3];
            bool flagged;
            const bool ok = ag_sweep(a.gran1, G, tag1, w, lane, pref, tot, flagged);
            if (lane == 0) { s_misc[M_OK] = ok ? 1u : 0u; s_misc[M_PREF_C] = pref[0]; s_misc[M_TOT_C] = tot[0]; }
        }
        __syncthreads();
        RES_STAMP(3);
        if (s_misc[M_FAIL] && tid == 0) ag_store32(&a.fail[0], a.epoch);
        if (!s_misc[M_OK]) {                                                   // time-out: release the others and leave
            if (tid == 0) {
                ag_store32(&a.fail[1], a.epoch);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ag_store(&a.gran2[w], ((u64)tag2 << 48) | RES_GRAN_FAIL);
                res_leave(a);
            }
            return;
        }
        prefC = s_misc[M_PREF_C]; totC = s_misc[M_TOT_C];
        // ---- B: classes of the anticommuting rows and the final coefficient of those that merge; slots and notes go back to zero ---
        for (int r = tid; r < Rw; r += RES_THREADS) {
            const uint8_t info = s_info[r];
            if (!(info & 1)) continue;
            const uint8_t st = s_cls[r];
            int part = (st & CL_SECOND) ? (int)s_ps[r] : -1;
            if (st & CL_CLAIMED) {
                const u32 pv = ag_load32(&a.partner[row0 + r]);
                if (pv) { part = (int)pv - 1; ag_store32(&a.partner[row0 + r], 0u); }
                ag_store(&a.slots[s_ps[r]], 0ULL);
            }
            const f64x2 c = s_coef[r];
            double sr = __dmul_rn(c.x, a.cos_t), si = __dmul_rn(c.y, a.cos_t);
            uint8_t cls = 0;
            double pr, pi;
            if (part >= 0) {                                                   // (0 + cos c_t) + (-i sin) i^{e'} c_p, in that order
                const f64x2 cp = coeff2[part];
                phase_mul(cp.x, cp.y, (info >> 3) & 3, pr, pi);
                sr = __dadd_rn(sr, __dmul_rn(pi, a.sin_t));
                si = __dadd_rn(si, -__dmul_rn(pr, a.sin_t));
                s_coef[r] = f64x2{sr, si};                                     // final; an unmatched row keeps c: cos c and the new row's
                cls |= CL_MATCHED;                                             // coefficient are formed from it when they are written
            } else {                                                           // its product row is new
                phase_mul(c.x, c.y, (info >> 1) & 3, pr, pi);
                if (res_keep(__dmul_rn(pi, a.sin_t), -__dmul_rn(pr, a.sin_t), a.thr)) cls |= CL_N;
            }
            if (res_keep(sr, si, a.thr)) cls |= CL_A;
            s_cls[r] = cls;
        }
    } else {
        // ---- Clifford: class and rotated coefficient per row (k_rotc_classify of rotate.hip) -----------------------------------
        const int k = a.k;
        for (int r = tid; r < Rw; r += RES_THREADS) {
            const uint8_t info = s_info[r];
            uint8_t cls = 0;
            if (!(info & 1)) {
                cls = CL_C;
            } else {
                const f64x2 c = s_coef[r];
                if (k & 1) {
                    if (res_keep(c.x, c.y, a.thr)) {
                        double x, y;
                        phase_mul(c.x, c.y, (info >> 1) & 3, x, y);
                        double pr = y, pi = -x;                                // c * i^e * (-i)
                        if (k == 3) { pr = -pr; pi = -pi; }
                        s_coef[r] = f64x2{pr, pi};
                        cls = CL_N | CL_MATCHED;                               // (MATCHED: the coefficient in LDS is the one to write)
                    }
                } else {
                    if (k == 2) s_coef[r] = f64x2{-c.x, -c.y};
                    cls = CL_A | CL_MATCHED;
                }
            }
            s_cls[r] = cls;
        }
    }
    __syncthreads();
    RES_STAMP(4);
    u32 *s_pos = s_ps;                                                         // the join state is dead: the word now holds the row's rank
    // ---- ranks of the rows inside the block: ballots per pass of 1,024 rows give the rank inside the wavefront and the counts per
    //      (pass, wavefront); wavefront 0 adds them up and publishes the block's granule at once (the all-gather is in flight while
    //      everybody turns the ranks into block-wide ones)
    {
        const int K = (Rw + RES_THREADS - 1) / RES_THREADS;                    // <= 16
        const u64 lt = (1ULL << lane) - 1ULL;
        for (int j = 0; j < K; ++j) {
            const int r = j * RES_THREADS + tid;
            const uint8_t cl = r < Rw ? s_cls[r] : 0;
            const bool an = r < Rw && (s_info[r] & 1);
            const u64 b0 = __ballot(cl & CL_C), b1 = __ballot(cl & CL_A), b2 = __ballot(cl & CL_N), b3 = __ballot(an);
            if (r < Rw) {
                s_pos[r] = (u32)__popcll(((cl & CL_C) ? b0 : b1) & lt);
                s_posn[r] = (u32)__popcll(b2 & lt);
            }
            if (lane == 0) s_wtot[j * 16 + wave] = (u64)__popcll(b0) | ((u64)__popcll(b1) << 16) | ((u64)__popcll(b2) << 32) | ((u64)__popcll(b3) << 48);
        }
        __syncthreads();
        if (wave == 0) {
            const int ne = K * 16;                                             // <= 256 entries, four per lane
            u64 sum = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int e = lane + 64 * i; if (e < ne) sum += s_wtot[e]; }
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            if (lane == 0) {
                const u64 nC = sum & 0xFFFFu, nA = (sum >> 16) & 0xFFFFu, nN = (sum >> 32) & 0xFFFFu, nAnti = (sum >> 48) & 0xFFFFu;
                // Clifford: ONE all-gather carries {rotated rows (class A or N: only one of them occurs per k), commuting rows, all anticommuting}
                if (MODE == 1) ag_store(&a.gran2[w], ((u64)tag2 << 48) | (nA + nN) | (nC << 16) | (nAnti << 32));
                else ag_store(&a.gran2[w], ((u64)tag2 << 48) | nA | (nN << 16) | (nAnti << 32) | (s_misc[M_FAIL] ? RES_GRAN_FAIL : 0ULL));
            }
        }
        for (int j = 0; j < K; ++j) {
            const int r = j * RES_THREADS + tid;
            u64 base = 0;
            for (int e = 0; e < j * 16 + wave; ++e) base += s_wtot[e];
            if (r < Rw) {
                s_pos[r] += (s_cls[r] & CL_C) ? (u32)base & 0xFFFFu : (u32)(base >> 16) & 0xFFFFu;
                s_posn[r] += (u32)(base >> 32) & 0xFFFFu;
            }
        }
    }
    __syncthreads();
    // ---- the report: the last wavefront of the last workgroup (the one with the shortest block) follows all-gather #2 before it
    //      turns to its share of the rows, and tells the host the counts the moment they are final and nobody has failed
    if (w == G - 1 && wave == RES_THREADS / 64 - 1) {
        u32 pref[3], tot[3];
        bool flagged;
        const bool ok = ag_sweep(a.gran2, G, tag2, w, lane, pref, tot, flagged);
        if (lane == 0 && ok && !flagged) {
            ag_store32(a.published, a.epoch);
            if (MODE == 1) res_report(a, 0, tot[1], (a.k & 1) ? 0 : tot[0], (a.k & 1) ? tot[0] : 0, tot[2]);
            else res_report(a, 0, totC, tot[0], tot[1], tot[2]);
        }
    }
    f64x2 *out_coeff2 = reinterpret_cast<f64x2 *>(a.out_coeff);
    if (MODE == 0) {
        // ---- C1: the commuting rows go out while the second all-gather is in flight -------------------------------------------
        int r = tid / Wq, c = tid - r * Wq;
        const int dr = RES_THREADS / Wq, dc = RES_THREADS - dr * Wq;
        for (int it = 0, i = tid; i < nchunk; ++it, i += RES_THREADS) {
            if (s_cls[r] & CL_C) row_store(RES_CHUNK(it, i), &a.out_rows[(i64)(prefC + s_pos[r]) * Wq + c]);
            r += dr; c += dc;
            if (c >= Wq) { c -= Wq; ++r; }
        }
        for (int r2 = tid; r2 < Rw; r2 += RES_THREADS)
            if (s_cls[r2] & CL_C) {
                const i64 d = (i64)prefC + s_pos[r2];
                out_coeff2[d] = s_coef[r2];
                if (a.out_hash) a.out_hash[d] = a.hin[row0 + r2];
            }
    }
    RES_STAMP(5);
    // ---- g2 ------------------------------------------------------------------------------------------------------------------------
    if (wave == 0) {
        u32 pref[3], tot[3];
        bool flagged;
        const bool ok = ag_sweep(a.gran2, G, tag2, w, lane, pref, tot, flagged);
        if (lane == 0) {
            s_misc[M_OK] = ok ? 1u : 0u;
            if (MODE == 1) {      // rotated rows first (filed under A or N, whichever this k produces), then the commuting ones
                s_misc[M_PREF_A] = pref[0]; s_misc[M_TOT_A] = (a.k & 1) ? 0 : tot[0]; s_misc[M_PREF_N] = pref[0]; s_misc[M_TOT_N] = (a.k & 1) ? tot[0] : 0;
                s_misc[M_PREF_C] = pref[1]; s_misc[M_TOT_C] = tot[1];
            } else {
                s_misc[M_PREF_A] = pref[0]; s_misc[M_TOT_A] = tot[0]; s_misc[M_PREF_N] = pref[1]; s_misc[M_TOT_N] = tot[1];
            }
            s_misc[M_TOT_ANTI] = tot[2];
        }
    }
    __syncthreads();
    RES_STAMP(6);
    if (!s_misc[M_OK]) {
        if (tid == 0) {
            ag_store32(&a.fail[1], a.epoch);
            res_leave(a);
        }
        return;
    }
    if (MODE == 1) { prefC = s_misc[M_PREF_C]; totC = s_misc[M_TOT_C]; }
    const u32 prefA = s_misc[M_PREF_A], totA = s_misc[M_TOT_A], prefN = s_misc[M_PREF_N], totN = s_misc[M_TOT_N];
    // output order: non-Clifford [commuting | cos * anticommuting | new rows]; Clifford [rotated anticommuting | commuting]
    const i64 baseC = (MODE == 1 ? (i64)totA + totN : 0) + prefC;
    const i64 baseA = (MODE == 1 ? 0 : (i64)totC) + prefA;
    const i64 baseN = (MODE == 1 ? 0 : (i64)totC + totA) + prefN;
    // ---- C2: the remaining rows, coefficients and hashes ----------------------------------------------------------------------
    {
        int r = tid / Wq, c = tid - r * Wq;
        const int dr = RES_THREADS / Wq, dc = RES_THREADS - dr * Wq;
        for (int it = 0, i = tid; i < nchunk; ++it, i += RES_THREADS) {
            const uint8_t cl = s_cls[r];
            if (cl & ((MODE == 1 ? CL_C : 0) | CL_A | CL_N)) {
                const u32x4 x = RES_CHUNK(it, i);
                if (MODE == 1 && (cl & CL_C)) row_store(x, &a.out_rows[(baseC + s_pos[r]) * Wq + c]);
                if (cl & CL_A) row_store(x, &a.out_rows[(baseA + s_pos[r]) * Wq + c]);
                if (cl & CL_N) row_store(x ^ sq4[c], &a.out_rows[(baseN + s_posn[r]) * Wq + c]);
            }
            r += dr; c += dc;
            if (c >= Wq) { c -= Wq; ++r; }
        }
        for (int r2 = tid; r2 < Rw; r2 += RES_THREADS) {
            const uint8_t cl = s_cls[r2];
            if (!(cl & ((MODE == 1 ? CL_C : 0) | CL_A | CL_N))) continue;
            const f64x2 c = s_coef[r2];
            const u64 h = a.out_hash ? a.hin[row0 + r2] : 0ULL;
            if (MODE == 1 && (cl & CL_C)) {
                const i64 d = baseC + s_pos[r2];
                out_coeff2[d] = c;
                if (a.out_hash) a.out_hash[d] = h;
            }
            if (cl & CL_A) {
                const i64 d = baseA + s_pos[r2];
                out_coeff2[d] = (cl & CL_MATCHED) ? c : f64x2{__dmul_rn(c.x, a.cos_t), __dmul_rn(c.y, a.cos_t)};
                if (a.out_hash) a.out_hash[d] = h;
            }
            if (cl & CL_N) {
                const i64 d = baseN + s_posn[r2];
                f64x2 cn = c;
                if (MODE == 0) {                                               // (-i sin) i^e c, from the row's own coefficient
                    double pr, pi;
                    phase_mul(c.x, c.y, (s_info[r2] >> 1) & 3, pr, pi);
                    cn = f64x2{__dmul_rn(pi, a.sin_t), -__dmul_rn(pr, a.sin_t)};
                }
                out_coeff2[d] = cn;
                if (a.out_hash) a.out_hash[d] = h ^ a.hq;
            }
        }
    }
#undef RES_CHUNK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               // every wavefront's rows have left ...
    __syncthreads();
    RES_STAMP(7);
    if (tid == 0) res_leave(a);              // ... before the block counts as gone
}

typedef void (*ResKernel)(const ResArgs);
static ResKernel res_kernel(bool clifford, int Wq) {
#define RES_PICK(M) \
    switch (Wq) { case 1: return k_rot_resident<M, 1>; case 2: return k_rot_resident<M, 2>; case 4: return k_rot_resident<M, 4>; case 8: return k_rot_resident<M, 8>; \
                  case 16: return k_rot_resident<M, 16>; case 32: return k_rot_resident<M, 32>; default: return k_rot_resident<M, 0>; }
    if (clifford) { RES_PICK(1) } else { RES_PICK(0) }
#undef RES_PICK
}

static u64 *g_res_trace = nullptr;
static int g_res_trace_wgs = 0;

int rotate_resident_trace(u64 *out, int max_wgs, int *n_wgs) {
    if (!g_res_trace) { *n_wgs = 0; return SYMGPU_OK; }
    const int n = g_res_trace_wgs < max_wgs ? g_res_trace_wgs : max_wgs;
    HIP_TRY(hipStreamSynchronize(ctx().stream));
    HIP_TRY(hipMemcpy(out, g_res_trace, (size_t)n * 16 * 8, hipMemcpyDeviceToHost));
    *n_wgs = n;
    return SYMGPU_OK;
}

int rotate_resident_try(symgpu_op_t in, const u64 *q_host, double cos_t, double sin_t, int clifford_k, double thr, symgpu_op_t *out,
                        int *all_commute, int *done) {
    *done = 0;
    Context &c = ctx();
    const i64 t_enter = host_ns();
    if (const char *e = getenv("SYMGPU_ROT_RESIDENT")) {                          // read on every call: 0 = off, 2 = on again after a failure, 3 = tests: inject a time-out
        if (e[0] == '0') return SYMGPU_OK;
        if (e[0] == '2') c.res_disabled = false;
    }
    if (c.res_disabled) return SYMGPU_OK;
    const i64 T = in->T;
    const int Wq = in->Wq, W = 2 * Wq;
    if (T < 1 || W > RES_MAX_W || T >= ((i64)1 << 22) - 1) return SYMGPU_OK;
    const bool clifford = clifford_k >= 0;
    if ((!clifford || (clifford_k & 1)) && !in->dup_free) return SYMGPU_OK;      // merges possible: the multi-launch paths check / handle them
    bool have_hash = in->hash && c.hash_tab && in->hash_seed == c.hash_seed;
    // geometry: at most one workgroup per CU, at least RES_MIN_ROWS rows each
    i64 G = (T + RES_MIN_ROWS - 1) / RES_MIN_ROWS;
    const i64 gmax = c.num_cu < RES_MAX_WG ? c.num_cu : RES_MAX_WG;
    if (G > gmax) G = gmax;
    const i64 R = (T + G - 1) / G;
    G = (T + R - 1) / R;
    if (R > 16384) return SYMGPU_OK;
    // the rows of the block in LDS; if they do not fit, the first two 1,024-chunk rounds of the load stay in registers (+32 KB per CU:
    // 1.46e5 -> 1.76e5 terms of 1,000 qubits — the operator a repeated rotation of 1e5 terms grows into, 1.5e5, is resident)
    const bool pow2 = Wq <= 32 && (Wq & (Wq - 1)) == 0;
    int nreg = 0, hbm = 0;
    ResLayout L = res_layout((int)R, Wq, 0);
    if ((size_t)L.total > RES_LDS_MAX && pow2) { nreg = 2; L = res_layout((int)R, Wq, nreg); }
    // beyond that (round 6): only the 26 bytes per row stay on the chip, the rows are read a second
    // time when they are written out — 1e5 terms of 2,000 qubits (51 MB): 73 us on the multi-launch path, see DESIGN 3.4
    const char *hbm_env = getenv("SYMGPU_ROT_HBM");                                  // 0: off; 2 (tests): also for operators that would fit the chip
    if (((size_t)L.total > RES_LDS_MAX && !(hbm_env && hbm_env[0] == '0')) || (hbm_env && hbm_env[0] == '2')) { nreg = 0; hbm = 1; L = res_layout((int)R, Wq, 0, 1); }
    if ((size_t)L.total > RES_LDS_MAX) return SYMGPU_OK;
    const bool attr_ok = SG_DEVICE_ONCE(([] {
        for (int m = 0; m < 2; ++m)
            for (int wq : {1, 2, 4, 8, 16, 32, 3})
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(res_kernel(m == 1, wq)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RES_LDS_MAX) != hipSuccess) return false;
        return true;
    }()));
    if (!attr_ok) {
        (void)hipGetLastError(); c.res_disabled = true;
        note_degraded("one-launch rotation (k_rot_resident) off: the runtime refused its LDS size; rotations take the multi-launch kernels");
        return SYMGPU_OK;
    }
    hipStream_t st = c.stream;
    if (!clifford && !have_hash) {
        // a duplicate-free operator without cached hashes (straight from a cleanup): hash its rows once, on the handle
        SG_TRY(ensure_hash_tables(c.hash_tab ? c.hash_seed : 1));
        if (in->hash) { dev_free(in->hash); in->hash = nullptr; }
        SG_TRY(dev_alloc((size_t)in->capacity * 8 + 16, (void **)&in->hash));
        in->hash_seed = c.hash_seed;
        SG_TRY(hash_rows(in->rows, T, W, in->hash));
        have_hash = true;
    }
    constexpr size_t state_words = 2 * RES_MAX_WG + 2;                            // granules, {fail[0], fail[1]}, {finished, published}
    u32 &finished_base = c.res_finished_base;                                    // (per device: the counters live in the device's res_state)
    if (!c.res_state) {
        HIP_TRY(hipMalloc((void **)&c.res_state, state_words * 8));
        c.res_epoch = 0;
    }
    if (c.res_epoch == 0 || c.res_epoch >= 16382) {                               // fresh state, or the 16-bit granule tag wraps
        HIP_TRY(hipMemsetAsync(c.res_state, 0, state_words * 8, st));
        c.res_epoch = 0;
        finished_base = 0;
    }
    ++c.res_epoch;
    ResArgs a;
    a.rows = reinterpret_cast<const u32x4 *>(in->rows); a.coeff = in->coeff; a.hin = have_hash ? in->hash : nullptr;
    a.T = T; a.Wq = Wq; a.R = (int)R;
    int GA = 1;
    while (GA < Wq && GA < 64) GA <<= 1;
    a.GA = GA;
    a.nreg = nreg;
    a.hbm = hbm;
    a.yq = 0;
    for (int ww = 0; ww < Wq; ++ww) a.yq += (u32)__builtin_popcountll(q_host[ww] & q_host[Wq + ww]);
    a.cos_t = cos_t; a.sin_t = sin_t; a.thr = thr; a.k = clifford_k;
    a.hq = have_hash ? host_row_hash(q_host, W) : 0;
    a.slots = nullptr; a.mask = 0; a.partner = nullptr;
    if (!clifford) {
        // join table (>= 4 slots per row) and partner notes: all-zero between launches — the kernel zeroes what it used; after a
        // launch that did not complete (res_dirty) they are cleared here
        size_t cap = 4096;
        static const int slots_per_row = [] { const char *e = SG_TUNE("SYMGPU_RES_SLOTS"); const int v = e ? atoi(e) : 4; return v >= 2 && v <= 64 ? v : 4; }();
        while ((i64)cap < (i64)slots_per_row * T) cap <<= 1;
        if (cap > c.res_table_cap) {
            if (c.res_table) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(c.res_table); c.res_table = nullptr; c.res_table_cap = 0; }
            HIP_TRY(hipMalloc((void **)&c.res_table, cap * 8));
            c.res_table_cap = cap;
            HIP_TRY(hipMemsetAsync(c.res_table, 0, cap * 8, st));
        } else if (c.res_dirty) {
            HIP_TRY(hipMemsetAsync(c.res_table, 0, c.res_table_cap * 8, st));
        }
        if ((size_t)T > c.rot_partner_cap) {
            if (c.rot_partner) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(c.rot_partner); c.rot_partner = nullptr; c.rot_partner_cap = 0; }
            size_t pcap = 4096;
            while (pcap < (size_t)T) pcap <<= 1;
            HIP_TRY(hipMalloc((void **)&c.rot_partner, pcap * 4));
            c.rot_partner_cap = pcap;
            HIP_TRY(hipMemsetAsync(c.rot_partner, 0, pcap * 4, st));
        } else if (c.res_dirty) {
            HIP_TRY(hipMemsetAsync(c.rot_partner, 0, c.rot_partner_cap * 4, st));
        }
        c.res_dirty = false;
        a.slots = c.res_table; a.mask = (u32)(cap - 1); a.partner = c.rot_partner;
    }
    a.gran1 = c.res_state; a.gran2 = c.res_state + RES_MAX_WG; a.fail = reinterpret_cast<u32 *>(c.res_state + 2 * RES_MAX_WG);
    a.finished = a.fail + 2; a.finish_target = finished_base + (u32)G;
    finished_base += (u32)G;
    a.epoch = c.res_epoch;
    { const char *e = getenv("SYMGPU_ROT_RESIDENT"); a.inject = (e && e[0] == '3') ? 1 : 0; }                      // 3 = tests: the kernel reports a failed verification
    a.trace = nullptr;
    if (const char *e = SG_TUNE("SYMGPU_RES_TRACE")) if (e[0] == '1') {
        if (!g_res_trace) { HIP_TRY(hipMalloc((void **)&g_res_trace, (size_t)RES_MAX_WG * 16 * 8)); }
        HIP_TRY(hipMemsetAsync(g_res_trace, 0, (size_t)RES_MAX_WG * 16 * 8, st));
        a.trace = g_res_trace;
        g_res_trace_wgs = (int)G;
    }
    // the report: bytes 32..55 of the context's pinned count block ([late 4 | - 4 | word0 8 | word1 8]); tags 1..65535 (the block starts zeroed)
    RotCounts *hcnt = nullptr, *hcnt_dev = nullptr;
    SG_TRY(host_counts(&hcnt, &hcnt_dev));
    volatile u32 *host_late = reinterpret_cast<u32 *>(hcnt) + 8;
    volatile u64 *host_words = reinterpret_cast<u64 *>(hcnt) + 5;
    if (*host_late != 0) {                                                        // a launch failed AFTER it had reported its counts
        c.res_disabled = true; c.res_epoch = 0; c.res_dirty = true;
        note_degraded("one-launch rotation (k_rot_resident) off: a launch failed after reporting; rotations take the multi-launch kernels");
        *host_late = 0;
        set_error("rotate resident: a previous launch failed after it had reported success; its result is invalid");
        return SYMGPU_E_HIP;
    }
    u32 &host_tag = c.res_host_tag;
    host_tag = host_tag >= 65535 ? 1 : host_tag + 1;
    a.host_late = reinterpret_cast<u32 *>(hcnt_dev) + 8; a.host_words = reinterpret_cast<u64 *>(hcnt_dev) + 5; a.host_tag = host_tag;
    a.published = a.fail + 3;
    for (int ww = 0; ww < RES_MAX_W; ++ww) a.q.w[ww] = ww < W ? q_host[ww] : 0ULL;
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(clifford ? T : 2 * T, Wq, 1, &res));                   // upper bound: no host round trip before the rows are written
    if (have_hash) {
        const int rc = dev_alloc((size_t)res->capacity * 8 + 16, (void **)&res->hash);
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        res->hash_seed = in->hash_seed;
    }
    a.out_rows = reinterpret_cast<u32x4 *>(res->rows); a.out_coeff = res->coeff; a.out_hash = res->hash;
    const i64 t_launch = host_ns();
    {
        ProfScope prof(4);
        hipLaunchKernelGGL(res_kernel(clifford, Wq), dim3((unsigned)G), dim3(RES_THREADS), (size_t)L.total, st, a);
    }
    hipError_t e = hipGetLastError();
    const i64 t_wait = host_ns();
    g_counters[4] += t_launch - t_enter; g_counters[5] += t_wait - t_launch;
    if (e != hipSuccess) { symgpu_op_free(res); c.res_dirty = true; return hip_fail(e, "rotate resident", __FILE__, __LINE__); }
    // One workgroup reports the counts as soon as they are final (after the second all-gather, while the rows are still being written):
    // poll the two tagged words in pinned memory instead of synchronising the stream — whatever the caller enqueues next is stream
    // ordered behind the kernel, and its preparation overlaps the kernel's tail.  The kernel gives up by itself after ~1 s and then
    // reports a failure code from its last workgroup; no report although the stream is idle: failure.
    u64 w0 = 0, w1 = 0;
    {
        bool seen = false;
        for (u64 spin = 0; spin < (1ULL << 34); ++spin) {
            w0 = __atomic_load_n(host_words, __ATOMIC_ACQUIRE); w1 = __atomic_load_n(host_words + 1, __ATOMIC_ACQUIRE);
            if ((u32)(w0 >> 48) == host_tag && (u32)(w1 >> 48) == host_tag) { seen = true; break; }
            if ((spin & 0xFFFFF) == 0xFFFFF) {
                const hipError_t qe = hipStreamQuery(st);
                if (qe != hipErrorNotReady) {                                    // finished (or failed): one last look
                    if (qe != hipSuccess) e = qe;
                    w0 = __atomic_load_n(host_words, __ATOMIC_ACQUIRE); w1 = __atomic_load_n(host_words + 1, __ATOMIC_ACQUIRE);
                    seen = (u32)(w0 >> 48) == host_tag && (u32)(w1 >> 48) == host_tag;
                    break;
                }
            }
        }
        if (!seen && e == hipSuccess) {
            e = hipStreamSynchronize(st);
            w0 = __atomic_load_n(host_words, __ATOMIC_ACQUIRE); w1 = __atomic_load_n(host_words + 1, __ATOMIC_ACQUIRE);
            seen = (u32)(w0 >> 48) == host_tag && (u32)(w1 >> 48) == host_tag;
        }
        if (e != hipSuccess) { symgpu_op_free(res); c.res_dirty = true; return hip_fail(e, "rotate resident", __FILE__, __LINE__); }
        if (!seen) { w0 = 0; w1 = 1ULL << 44; }                                   // no report at all: code 1
    }
    g_counters[6] += host_ns() - t_wait;
    RotCounts hc;
    hc.nC = (u32)(w0 >> 22) & 0x3FFFFFu; hc.nA = (u32)w0 & 0x3FFFFFu; hc.nN = (u32)(w1 >> 22) & 0x3FFFFFu; hc.nAnti = (u32)w1 & 0x3FFFFFu;
    hc.dup = (u32)(w1 >> 44) & 0xFu;
    if (hc.dup != 0) {                                                             // verification failed (2), timed out (3), or no report at all (1)
        symgpu_op_free(res);
        ++g_counters[2];
        if (hc.dup != 2) {                                                         // time-out: arrival counts are in an unknown state
            c.res_disabled = true; c.res_epoch = 0;
            note_degraded("one-launch rotation (k_rot_resident) off: an in-kernel wait timed out (workgroups not co-resident?); rotations take the multi-launch kernels");
        }
        // code 2 too: every owner zeroes its slot and note in phase B, but that relies on every workgroup getting there; one memset on a
        // path that is about to take the multi-launch kernels anyway makes the next launch independent of it
        c.res_dirty = true;
        return SYMGPU_OK;
    }
    *done = 1;
    ++g_counters[1];
    if (hc.nAnti == 0) { symgpu_op_free(res); *all_commute = 1; return SYMGPU_OK; }   // identity action (base.py:1131-1133)
    res->T = (i64)hc.nC + hc.nA + hc.nN;
    res->dup_free = clifford ? in->dup_free : 1;
    *out = res;
    *all_commute = 0;
    return SYMGPU_OK;
}

}  // namespace symgpu
