// rotate.hip — single-Pauli rotation (reference: PauliwordOp._rotate_by_single_Pword,
// symmer/operators/base.py:1090-1161) as ONE fused device pass over a device-resident operator.
//
//   R(t) P R(t)^+ = P                        if [P, Q] = 0
//                 = cos(t) P + sin(t)(-i P Q) if {P, Q} = 0
//
// analyze : per row, anticommutation parity with Q and the phase exponent e of P*Q
//           (e = (3(Y_P+Y_Q) + Y_out + 2|x_P & z_Q|) mod 4), G lanes per row, xor/popcount + shuffles.
// scan    : positions of anticommuting rows (exclusive scan of flags).
// build   : non-Clifford: stack [commuting | cos * anticommuting | (-i sin) i^e * (anticommuting ^ Q)] in the
//           reference's order (base.py:1158-1161), then first-occurrence cleanup (cleanup.hip) merges P^Q
//           partners; Clifford (angle = k*pi/2): [rotated anticommuting | commuting], no merge (base.py:1139-1154);
//           odd k: row ^ Q with c * i^e * (-i), k in {2,3}: negated (k is NOT reduced mod 4, base.py:1148).
// Equivalent to the reference's three intermediate cleanups when the input has no duplicate rows
// (every operator that left cleanup() qualifies).  Inputs WITH duplicate rows: the non-Clifford path
// detects them in its hash insert and falls back to stack + cleanup; the odd-k Clifford path forms anticom_self * Q
// through __mul__ in the reference (base.py:1143), which merges duplicate product rows and thresholds the SUMS — so an
// operator not known to be duplicate-free (symgpu_op_s::dup_free) is checked once with the same hash insert, and if
// duplicates exist the rotated anticommuting part goes through cleanup before the commuting rows are appended.
#include "common.h"
#include "rotate_common.h"
#include <stdlib.h>

namespace symgpu {


// flags[t] = 1 iff row t anticommutes with q;  ph[t] = phase exponent e of (row_t * q).
// HASH: also the linear row hash h1 of cleanup.hip (same tables, same per-lane Horner) for the hash-join fast path.
__device__ __forceinline__ u64 rot_rotl64(u64 x, int r) { r &= 63; return r ? ((x << r) | (x >> (64 - r))) : x; }

// insert row t (hash h) — duplicates (same tag AND same words) raise the flag; one lane per row
__device__ __forceinline__ void jt_insert(const JoinTable jt, const u64 *__restrict__ rows, int W, i64 t, u64 h) {
    const u64 entry = (h & 0xFFFFFFFF00000000ULL) | ((u64)jt.gen << 22) | (u64)(t + 1);
    u32 pos = (u32)mix64(h) & jt.mask;
    for (;;) {
        u64 v = __hip_atomic_load(&jt.slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (jt_gen(v) != jt.gen) {                                   // empty (or left over from an earlier rotation): claim it
            const u64 old = atomicCAS(reinterpret_cast<unsigned long long *>(&jt.slots[pos]), (unsigned long long)v, (unsigned long long)entry);
            if (old == v) return;
            v = old;
            if (jt_gen(v) != jt.gen) continue;                       // lost the race to a stale writer?  cannot happen, but retry the slot
        }
        if ((v >> 32) == (h >> 32)) {                                // same 32-bit tag: duplicate row, or a tag collision
            const i64 o = jt_row(v);
            bool same = true;
            for (int w = 0; w < W; ++w) same &= (rows[o * W + w] == rows[t * W + w]);
            if (same) { jt.flags[0] = jt.gen; return; }
        }
        pos = (pos + 1) & jt.mask;
    }
}

// HASH: compute the row hashes (-> hout); otherwise INSERT reads them from hin.  INSERT: put every row into the join table.
// QARG: the rotation's Pauli row Q arrives BY VALUE in the kernel arguments (rows of <= 64 words) instead of in `q_dev`, which
// block 0 then fills for the kernels that follow on the stream — no host-to-device copy in front of the first kernel of a rotation.
template <bool HASH, bool INSERT, bool QARG = false>
__global__ __launch_bounds__(256) void k_rot_analyze(const u64 *__restrict__ rows, i64 T, int Wq, int G, u64 *__restrict__ q_dev,
                                                      u32 *__restrict__ flags, uint8_t *__restrict__ ph, const u64 *__restrict__ tab_g,
                                                      u64 *__restrict__ hout, const u64 *__restrict__ hin, JoinTable jt, QArg qa = QArg()) {
    __shared__ u64 tab[HASH ? 8 * 256 : 1];
    __shared__ u64 sq[QARG ? 64 : 1];
    if (QARG) {
        if ((int)threadIdx.x < 2 * Wq) {
            const u64 v = qa.w[threadIdx.x];
            sq[threadIdx.x] = v;
            if (blockIdx.x == 0) q_dev[threadIdx.x] = v;
        }
    }
    if (HASH) {
        for (int k = threadIdx.x; k < 8 * 256; k += 256) tab[k] = tab_g[2 * k];      // h1 entries only
    }
    if (HASH || QARG) __syncthreads();
    const u64 *q = QARG ? sq : q_dev;
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    // Y count of q (every lane redundantly; Wq is small)
    int yq = 0;
    for (int w = 0; w < Wq; ++w) yq += __popcll(q[w] & q[Wq + w]);
    const int W = 2 * Wq;
    for (i64 t0 = (i64)blockIdx.x * rows_per_block; t0 < T; t0 += (i64)gridDim.x * rows_per_block) {
        const i64 t = t0 + rsub;
        u64 par = 0, flip = 0;
        int yp = 0, yout = 0;
        u64 h1 = 0;
        if (t < T) {
            const u64 *r = rows + t * 2 * Wq;
            for (int w = g; w < Wq; w += G) {
                const u64 x = r[w], z = r[Wq + w], xq = q[w], zq = q[Wq + w];
                par ^= (x & zq) ^ (z & xq);
                flip ^= x & zq;
                yp += __popcll(x & z);
                yout += __popcll((x ^ xq) & (z ^ zq));
            }
            if (HASH) {
                // lane g of a 64-lane virtual row owns words g, g+64, ...; here G <= 64 lanes cover the row, so every
                // lane loops over the virtual lanes vg = g, g+G, ... < 64 it stands for
                const int n_blk = (W + 63) / 64;
                for (int vg = g; vg < 64; vg += G) {
                    u64 hv = 0;
                    for (int b = 0; b < n_blk; ++b) {
                        const int w = b * 64 + vg;
                        u64 a1 = 0;
                        if (w < W) {
                            const u64 x = r[w];
#pragma unroll
                            for (int k = 0; k < 8; ++k) a1 ^= tab[k * 256 + (int)((x >> (8 * k)) & 255)];
                        }
                        hv ^= hv << 13; hv ^= hv >> 7; hv ^= hv << 17;
                        hv ^= rot_rotl64(a1, vg);
                    }
                    h1 ^= hv;
                }
            }
        }
        int pp = __popcll(par) & 1, fp = __popcll(flip) & 1;
        for (int off = G >> 1; off > 0; off >>= 1) {
            pp ^= __shfl_xor(pp, off);
            fp ^= __shfl_xor(fp, off);
            yp += __shfl_xor(yp, off);
            yout += __shfl_xor(yout, off);
            if (HASH) h1 ^= __shfl_xor(h1, off);
        }
        if (g == 0 && t < T) {
            flags[t] = (u32)pp;
            ph[t] = (uint8_t)((3 * (yp + yq) + yout + 2 * fp) & 3);
            if (HASH) hout[t] = h1;
            if (INSERT) jt_insert(jt, rows, W, t, HASH ? h1 : hin[t]);
        }
    }
}


// The same analysis with ONE 16-byte chunk of a row per lane (rows whose 16-byte chunk count WQ = words per X block is a power
// of two <= 64 and whose hashes are cached): a row is an aligned group of WQ lanes, X words in its lower and Z words in its upper
// half, loads are 1 KiB per wave instruction (k_rot_analyze reads 8 bytes per lane and walks six dependent chains per block at
// 1e5 rows: 22 us; this: 9 us).  Lane exchange as in product.hip's row stream: x & z and (x ^ xq) & (z ^ zq) need the other half
// of the row (DPP / ds_bpermute), x & zq and z & xq only the other half of Q, which every lane reads from LDS.
template <int WQ, bool INSERT, bool QARG>
__global__ __launch_bounds__(256) void k_rot_analyze_chunks(const u32x4 *__restrict__ rows, i64 T, u64 *__restrict__ q_dev, u32 *__restrict__ flags,
                                                             uint8_t *__restrict__ ph, const u64 *__restrict__ hin, JoinTable jt, QArg qa) {
    __shared__ __attribute__((aligned(16))) u64 sq[2 * WQ];
    __shared__ int s_yq;
    if (INSERT) {
        // The join-table insert needs nothing from the analysis (row index + cached hash), and done by the one lane per row group
        // that ends the analysis it was a chain load -> atomic load -> CAS with 4 lanes of 64 busy: 10 of the kernel's 18 us.  The
        // FIRST ceil(T / 256) blocks of the grid are insert blocks, ONE LANE PER ROW (dispatched first: their latency chain runs
        // while the analysis blocks stream the rows behind them).
        const i64 n_ins = (T + 255) / 256;
        if ((i64)blockIdx.x < n_ins) {
            const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
            if (t < T) jt_insert(jt, reinterpret_cast<const u64 *>(rows), 2 * WQ, t, hin[t]);
            return;
        }
    }
    const i64 bx = (i64)blockIdx.x - (INSERT ? (T + 255) / 256 : 0);     // analysis block index
    if ((int)threadIdx.x < 2 * WQ) {
        const u64 v = QARG ? qa.w[threadIdx.x] : q_dev[threadIdx.x];
        sq[threadIdx.x] = v;
        if (QARG && bx == 0) q_dev[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int y = 0;
        for (int w = 0; w < WQ; ++w) y += __popcll(sq[w] & sq[WQ + w]);
        s_yq = y;
    }
    __syncthreads();
    constexpr int R = 256 / WQ;                                       // rows per block
    const int c = threadIdx.x & (WQ - 1);                            // chunk of the row: c < WQ/2 holds X words (WQ == 1: x and z word)
    const i64 t = bx * R + threadIdx.x / WQ;
    const bool valid = t < T;
    const u32x4 v = valid ? rows[t * WQ + c] : (u32x4)(0u);
    u32 par, ye;                                                      // par: |x & zq| + |z & xq| (+ |x & zq| << 16);  ye: Y_P | Y_out << 16
    if constexpr (WQ == 1) {
        const u32x4 qv = *reinterpret_cast<const u32x4 *>(sq);
        const u32 f = __popc(v.x & qv.z) + __popc(v.y & qv.w);
        par = f + __popc(v.z & qv.x) + __popc(v.w & qv.y) + (f << 16);
        ye = (__popc(v.x & v.z) + __popc(v.y & v.w)) | ((__popc((v.x ^ qv.x) & (v.z ^ qv.z)) + __popc((v.y ^ qv.y) & (v.w ^ qv.w))) << 16);
    } else {
        const u32x4 qs = reinterpret_cast<const u32x4 *>(sq)[c], qo = reinterpret_cast<const u32x4 *>(sq)[c ^ (WQ / 2)];
        const bool xhalf = c < WQ / 2;
        const u32 p = __popc(v.x & qo.x) + __popc(v.y & qo.y) + __popc(v.z & qo.z) + __popc(v.w & qo.w);    // x & zq (X half), z & xq (Z half)
        const u32x4 o = {rot_other_half<WQ>(v.x), rot_other_half<WQ>(v.y), rot_other_half<WQ>(v.z), rot_other_half<WQ>(v.w)};
        const u32 yp = __popc(v.x & o.x) + __popc(v.y & o.y) + __popc(v.z & o.z) + __popc(v.w & o.w);
        const u32 yo = __popc((v.x ^ qs.x) & (o.x ^ qo.x)) + __popc((v.y ^ qs.y) & (o.y ^ qo.y)) + __popc((v.z ^ qs.z) & (o.z ^ qo.z)) +
                       __popc((v.w ^ qs.w) & (o.w ^ qo.w));
        par = rot_row_sum<WQ>(p + (xhalf ? (p << 16) : 0u));
        ye = rot_row_sum<WQ>(xhalf ? (yp | (yo << 16)) : 0u);          // both halves form the same x & z words: count them once
    }
    if (c == 0 && valid) {
        flags[t] = par & 1u;
        ph[t] = (uint8_t)((3u * ((ye & 0xFFFFu) + (u32)s_yq) + (ye >> 16) + 2u * ((par >> 16) & 1u)) & 3u);
    }
}

// Clifford odd-k: product rows whose coefficient is <= thr are dropped by the reference's `*` (cleanup inside
// _multiply_by_operator, base.py:789-793): fold that into the flag that is scanned.
__global__ void k_rot_keepflags(const u32 *__restrict__ anti, const double *__restrict__ coeff, i64 T, double thr, int drop_small,
                                u32 *__restrict__ keep) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        u32 k = anti[t];
        if (k && drop_small && !(hypot(coeff[2 * t], coeff[2 * t + 1]) > thr)) k = 0;
        keep[t] = k;
    }
}

// coefficients of the stacked operator.  MODE 0: non-Clifford; MODE 1: Clifford.
// apos = exclusive scan of `sel` (anticommuting-and-kept flags); n_sel = its total.
template <int MODE>
__global__ void k_rot_coeff(const double *__restrict__ coeff, const u32 *__restrict__ anti, const u32 *__restrict__ sel,
                            const u32 *__restrict__ apos, const u32 *__restrict__ cpos, const uint8_t *__restrict__ ph, i64 T, i64 n_sel,
                            i64 n_comm, double cos_t, double sin_t, int k, double *__restrict__ out_coeff, u32 *__restrict__ dst_main,
                            u32 *__restrict__ dst_prod) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        const double re = coeff[2 * t], im = coeff[2 * t + 1];
        u32 d_main = 0xffffffffu, d_prod = 0xffffffffu;
        if (MODE == 0) {
            if (!anti[t]) {
                d_main = cpos[t];                                   // commuting rows first, input order
                out_coeff[2 * (i64)d_main] = re; out_coeff[2 * (i64)d_main + 1] = im;
            } else {
                d_main = (u32)(n_comm + apos[t]);                   // cos * P
                out_coeff[2 * (i64)d_main] = __dmul_rn(re, cos_t); out_coeff[2 * (i64)d_main + 1] = __dmul_rn(im, cos_t);
                d_prod = (u32)(n_comm + n_sel + apos[t]);           // (-i sin) * i^e * c   on row P^Q
                double pr, pi;
                phase_mul(re, im, ph[t], pr, pi);
                out_coeff[2 * (i64)d_prod] = __dmul_rn(pi, sin_t); out_coeff[2 * (i64)d_prod + 1] = -__dmul_rn(pr, sin_t);
            }
        } else {
            if (!anti[t]) {
                d_main = (u32)(n_sel + cpos[t]);                    // commuting rows after the rotated ones
                out_coeff[2 * (i64)d_main] = re; out_coeff[2 * (i64)d_main + 1] = im;
            } else if (sel[t]) {
                double pr = re, pi = im;
                if (k & 1) {                                        // c * i^e * (-i)
                    double a, b;
                    phase_mul(re, im, ph[t], a, b);
                    pr = b; pi = -a;
                    d_prod = apos[t];
                } else {
                    d_main = apos[t];
                }
                if (k == 2 || k == 3) { pr = -pr; pi = -pi; }
                const u32 d = (k & 1) ? d_prod : d_main;
                out_coeff[2 * (i64)d] = pr; out_coeff[2 * (i64)d + 1] = pi;
            }
        }
        dst_main[t] = d_main;
        dst_prod[t] = d_prod;
    }
}

// rows of the stacked operator as 16-byte chunks
__global__ void k_rot_rows(const u32x4 *__restrict__ rows, const u32x4 *__restrict__ q, i64 T, int Wq, const u32 *__restrict__ dst_main,
                           const u32 *__restrict__ dst_prod, u32x4 *__restrict__ out) {
    const i64 total = T * Wq;
    for (i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (i64)gridDim.x * blockDim.x) {
        const i64 t = idx / Wq;
        const int c = (int)(idx - t * Wq);
        const u32x4 v = rows[idx];
        const u32 dm = dst_main[t], dp = dst_prod[t];
        if (dm != 0xffffffffu) out[(i64)dm * Wq + c] = v;
        if (dp != 0xffffffffu) out[(i64)dp * Wq + c] = v ^ q[c];
    }
}

__global__ void k_not_flags(const u32 *__restrict__ a, i64 T, u32 *__restrict__ out) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) out[t] = a[t] ? 0u : 1u;
}

static int grid_for(i64 n, int block = 256, int cap = 8192) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- fast non-Clifford path: hash-table join instead of the sort-based cleanup ---------------------------------------
// For an operator WITHOUT duplicate rows (anything that left cleanup()) the only possible merge is between a product row
// P_k ^ Q and the existing row R_j = P_k ^ Q, which is anticommuting too and whose own product row is P_k: partners come
// in pairs.  So: hash every row (linear hash: h(P^Q) = h(P) ^ h(Q)), insert the rows in an open-addressing table, look
// up h(P_k) ^ h(Q) for every anticommuting row and verify the candidate word by word.  A duplicate row in the input
// (same hash AND same words) raises `dup`; the caller then takes the general sort-based path, which handles it.
// Output order and sums are those of the reference (base.py:1158-1161 + cleanup): kept commuting rows, kept
// anticommuting rows with  cos*c_t + (-i sin) i^{e_p} c_p  (first-occurrence entry first), then the kept unmatched product
// rows; strict |c| > thr everywhere.

// pinned, device-mapped host copy of the counts (one per context): written by k_rotf_scan3, read after the final synchronisation
int host_counts(RotCounts **host, RotCounts **dev) {
    Context &c = ctx();
    if (!c.rot_host_cnt) {
        HIP_TRY(hipHostMalloc(&c.rot_host_cnt, 64, hipHostMallocMapped));
        for (int i = 0; i < 8; ++i) reinterpret_cast<volatile u64 *>(c.rot_host_cnt)[i] = 0;
        HIP_TRY(hipHostGetDevicePointer(&c.rot_host_cnt_dev, c.rot_host_cnt, 0));
    }
    *host = reinterpret_cast<RotCounts *>(c.rot_host_cnt);
    *dev = reinterpret_cast<RotCounts *>(c.rot_host_cnt_dev);
    return SYMGPU_OK;
}

// The same join, ONE lane per row, fused with the per-1024-row block counts (k_rotf_count): probe the generation-tagged table for
// h(P) ^ h(Q); only a tag hit reads rows (a lane then compares the two rows word by word).  1024 rows per block.
__global__ __launch_bounds__(1024) void k_rotf_match2(const u64 *__restrict__ rows, const double *__restrict__ coeff, const u64 *__restrict__ h, i64 T,
                                                       int W, const u64 *__restrict__ q, u64 hq, const u32 *__restrict__ anti,
                                                       const uint8_t *__restrict__ ph, JoinTable jt, double cos_t, double sin_t, double thr,
                                                       double *__restrict__ selfc, double *__restrict__ prodc, uint8_t *__restrict__ cls,
                                                       u32 *__restrict__ blk) {
    __shared__ u32 s_c[4];
    if (threadIdx.x < 4) s_c[threadIdx.x] = 0;
    __syncthreads();
    const i64 t = (i64)blockIdx.x * 1024 + threadIdx.x;
    const bool valid = t < T;
    const bool is_anti = valid && anti[t];
    uint8_t c = 0;
    if (valid) {
        i64 partner = -1;
        if (is_anti) {
            const u64 key = h[t] ^ hq;
            u32 pos = (u32)mix64(key) & jt.mask;
            for (;;) {
                const u64 v = jt.slots[pos];
                if (jt_gen(v) != jt.gen) break;
                if ((v >> 32) == (key >> 32)) {
                    const i64 o = jt_row(v);
                    bool same = true;
                    for (int w = 0; w < W; ++w) same &= (rows[o * W + w] == (rows[t * W + w] ^ q[w]));
                    if (same) { partner = o; break; }
                }
                pos = (pos + 1) & jt.mask;
            }
        }
        const double re = coeff[2 * t], im = coeff[2 * t + 1];
        if (!is_anti) {
            selfc[2 * t] = re; selfc[2 * t + 1] = im;
            if (hypot(re, im) > thr) c = 1;
        } else {
            double sr = __dmul_rn(re, cos_t), si = __dmul_rn(im, cos_t);
            if (partner >= 0) {                                  // merge: (0 + cos*c_t) + (-i sin) i^{e_p} c_p, in that order
                double pr, pi;
                phase_mul(coeff[2 * partner], coeff[2 * partner + 1], ph[partner], pr, pi);
                sr = __dadd_rn(sr, __dmul_rn(pi, sin_t));
                si = __dadd_rn(si, -__dmul_rn(pr, sin_t));
            } else {                                             // its product row is new
                double pr, pi;
                phase_mul(re, im, ph[t], pr, pi);
                const double nr = __dmul_rn(pi, sin_t), ni = -__dmul_rn(pr, sin_t);
                prodc[2 * t] = nr; prodc[2 * t + 1] = ni;
                if (hypot(nr, ni) > thr) c |= 4;
            }
            selfc[2 * t] = sr; selfc[2 * t + 1] = si;
            if (hypot(sr, si) > thr) c |= 2;
        }
        cls[t] = c;
    }
    const int lane = threadIdx.x & 63;
    const u64 b0 = __ballot(c & 1), b1 = __ballot(c & 2), b2 = __ballot(c & 4), b3 = __ballot(is_anti);
    if (lane == 0) {
        atomicAdd(&s_c[0], (u32)__popcll(b0)); atomicAdd(&s_c[1], (u32)__popcll(b1));
        atomicAdd(&s_c[2], (u32)__popcll(b2)); atomicAdd(&s_c[3], (u32)__popcll(b3));
    }
    __syncthreads();
    if (threadIdx.x < 4) blk[blockIdx.x * 4 + threadIdx.x] = s_c[threadIdx.x];
}

// three exclusive scans (kept-commuting, kept-anticommuting, kept-new) + the anticommuting count: the per-1024-row block counts
// come from k_rotf_match2 / k_rotc_classify, every block here adds the counts of the blocks before it (<= 4096) to its local ranks.
__global__ __launch_bounds__(1024) void k_rotf_scan3(const uint8_t *__restrict__ cls, i64 T, const u32 *__restrict__ blk, int n_blk,
                                                      u32 *__restrict__ pos_self, u32 *__restrict__ pos_new, RotCounts *__restrict__ cnt,
                                                      const u32 *__restrict__ jt_flags, u32 jt_gen_now, RotCounts *__restrict__ host_cnt = nullptr) {
    __shared__ u32 s_w[3][16];
    __shared__ u32 s_base[4], s_all[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 4) { s_base[threadIdx.x] = 0; s_all[threadIdx.x] = 0; }
    __syncthreads();
    {   // counts of the blocks before this one (and, for the last block, of all blocks)
        u32 before[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};
        for (int b = threadIdx.x; b < n_blk; b += 1024)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const u32 x = blk[b * 4 + k]; all[k] += x; if (b < (int)blockIdx.x) before[k] += x; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (int off = 32; off > 0; off >>= 1) { before[k] += __shfl_down(before[k], off); all[k] += __shfl_down(all[k], off); }
            if (lane == 0) { if (before[k]) atomicAdd(&s_base[k], before[k]); if (all[k]) atomicAdd(&s_all[k], all[k]); }
        }
    }
    const i64 t = (i64)blockIdx.x * 1024 + threadIdx.x;
    const uint8_t c = (t < T) ? cls[t] : 0;
    u32 ex[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const u64 b = __ballot((c >> k) & 1);
        ex[k] = __popcll(b & ((1ULL << lane) - 1ULL));
        if (lane == 0) s_w[k][wave] = __popcll(b);
    }
    __syncthreads();
    u32 off[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        for (int w2 = 0; w2 < wave; ++w2) off[k] += s_w[k][w2];
    if (t < T) {
        pos_self[t] = (c & 1) ? s_base[0] + off[0] + ex[0] : s_base[1] + off[1] + ex[1];
        pos_new[t] = s_base[2] + off[2] + ex[2];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        cnt->nC = s_all[0]; cnt->nA = s_all[1]; cnt->nN = s_all[2]; cnt->nAnti = s_all[3];
        cnt->dup = (jt_flags && jt_flags[0] == jt_gen_now) ? 1u : 0u;        // one read-back for the counts and the duplicate flag
        // the same five words straight into pinned host memory: the host reads them after the stream synchronisation that ends the
        // rotation, without a device-to-host copy in between
        if (host_cnt) { host_cnt->nC = s_all[0]; host_cnt->nA = s_all[1]; host_cnt->nN = s_all[2]; host_cnt->nAnti = s_all[3]; host_cnt->dup = cnt->dup; }
    }
}

__global__ void k_rotf_write(const u32x4 *__restrict__ rows, const u32x4 *__restrict__ q, i64 T, int Wq, const uint8_t *__restrict__ cls,
                             const u32 *__restrict__ pos_self, const u32 *__restrict__ pos_new, const RotCounts *__restrict__ cnt,
                             const double *__restrict__ selfc, const double *__restrict__ prodc, u32x4 *__restrict__ out_rows,
                             double *__restrict__ out_coeff, int clifford, const u64 *__restrict__ hin, u64 hq, u64 *__restrict__ hout) {
    // hin / hout (may be null): row hashes of the input and of the result — h is linear, so h(P ^ Q) = h(P) ^ h(Q)
    const i64 total = T * Wq;
    // output order: non-Clifford [commuting | cos * anticommuting | new rows]; Clifford [rotated anticommuting | commuting]
    const i64 baseC = clifford ? (i64)cnt->nA + cnt->nN : 0;
    const i64 baseA = clifford ? 0 : (i64)cnt->nC;
    const i64 baseN = clifford ? 0 : (i64)cnt->nC + cnt->nA;
    // The first ceil(T / 256) blocks move the 16-byte coefficients and the 8-byte hashes, ONE LANE PER ROW (consecutive rows of a class
    // go to consecutive slots: near-coalesced); the remaining blocks move the rows, one 16-byte chunk per lane.  Done by the chunk-0
    // lane of every row inside the row stream, the small stores — one lane of 16, scattered between the 256-byte row stores —
    // are what the product's row stream showed to be expensive (product.hip, tools/ubench_fused.hip).
    const i64 n_cf = (T + blockDim.x - 1) / blockDim.x;
    if ((i64)blockIdx.x < n_cf) {
        const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= T) return;
        const uint8_t k = cls[t];
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        if (k & 3) {
            const i64 d = (k & 1) ? baseC + pos_self[t] : baseA + pos_self[t];
            reinterpret_cast<f64x2 *>(out_coeff)[d] = reinterpret_cast<const f64x2 *>(selfc)[t];
            if (hout) hout[d] = hin[t];
        }
        if (k & 4) {
            const i64 d = baseN + pos_new[t];
            reinterpret_cast<f64x2 *>(out_coeff)[d] = reinterpret_cast<const f64x2 *>(prodc)[t];
            if (hout) hout[d] = hin[t] ^ hq;
        }
        return;
    }
    const int wsh = (Wq & (Wq - 1)) == 0 ? __builtin_ctz((unsigned)Wq) : -1;      // chunks per row a power of two: shift instead of a 64-bit divide
    const i64 n_row_blocks = (i64)gridDim.x - n_cf;
    // four chunks per lane and step, every load of a step issued before the first store: class byte, both slot words and the
    // chunk itself do not depend on each other (the one-chunk loop chained class -> branch -> loads: 3.8 TB/s at 10^6 terms)
    constexpr int WU = 4;
    const i64 stride = n_row_blocks * blockDim.x;
    for (i64 idx0 = ((i64)blockIdx.x - n_cf) * blockDim.x + threadIdx.x; idx0 < total; idx0 += stride * WU) {
        i64 t[WU];
        int c[WU];
        uint8_t k[WU];
        u32 ps[WU], pn[WU];
        u32x4 v[WU];
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const i64 idx = idx0 + u * stride < total ? idx0 + u * stride : idx0;
            t[u] = wsh >= 0 ? idx >> wsh : idx / Wq;
            c[u] = (int)(idx - t[u] * Wq);
            k[u] = cls[t[u]];
            ps[u] = pos_self[t[u]];
            pn[u] = pos_new[t[u]];
            v[u] = rows[idx];
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            if (idx0 + u * stride >= total) break;
            if (k[u] & 3) {
                const i64 d = (k[u] & 1) ? baseC + ps[u] : baseA + ps[u];
                __builtin_nontemporal_store(v[u], &out_rows[d * Wq + c[u]]);
            }
            if (k[u] & 4) {
                const i64 d = baseN + pn[u];
                __builtin_nontemporal_store(v[u] ^ q[c[u]], &out_rows[d * Wq + c[u]]);
            }
        }
    }
}

// Clifford rotation (angle = k * pi/2), classes + per-1024-row block counts in one launch:
//   commuting row            -> class 1, coefficient unchanged
//   anticommuting, even k    -> class 2 (the row itself), coefficient c (k = 0) or -c (k = 2)
//   anticommuting, odd k     -> class 4 (row ^ Q), coefficient c * i^e * (-i), negated for k = 3; rows with |c| <= thr are
//                               dropped as the reference's `*` does (cleanup inside _multiply_by_operator, base.py:789-793)
// `k` arrives already mapped by rotation_args (negative multiples are not reduced mod 4, base.py:1148).
__global__ __launch_bounds__(1024) void k_rotc_classify(const u32 *__restrict__ anti, const uint8_t *__restrict__ ph, const double *__restrict__ coeff,
                                                         i64 T, int k, double thr, uint8_t *__restrict__ cls, double *__restrict__ selfc,
                                                         double *__restrict__ prodc, u32 *__restrict__ blk) {
    __shared__ u32 s_c[4];
    if (threadIdx.x < 4) s_c[threadIdx.x] = 0;
    __syncthreads();
    const i64 t = (i64)blockIdx.x * 1024 + threadIdx.x;
    uint8_t c = 0;
    bool a = false;
    if (t < T) {
        const double re = coeff[2 * t], im = coeff[2 * t + 1];
        a = anti[t] != 0;
        if (!a) {
            c = 1;
            selfc[2 * t] = re; selfc[2 * t + 1] = im;
        } else if (k & 1) {
            if (hypot(re, im) > thr) {
                double x, y;
                phase_mul(re, im, ph[t], x, y);
                double pr = y, pi = -x;
                if (k == 3) { pr = -pr; pi = -pi; }
                prodc[2 * t] = pr; prodc[2 * t + 1] = pi;
                c = 4;
            }
        } else {
            const bool neg = (k == 2);
            selfc[2 * t] = neg ? -re : re; selfc[2 * t + 1] = neg ? -im : im;
            c = 2;
        }
        cls[t] = c;
    }
    const int lane = threadIdx.x & 63;
    const u64 b0 = __ballot(c & 1), b1 = __ballot(c & 2), b2 = __ballot(c & 4), b3 = __ballot(a);
    if (lane == 0) {
        atomicAdd(&s_c[0], (u32)__popcll(b0)); atomicAdd(&s_c[1], (u32)__popcll(b1));
        atomicAdd(&s_c[2], (u32)__popcll(b2)); atomicAdd(&s_c[3], (u32)__popcll(b3));
    }
    __syncthreads();
    if (threadIdx.x < 4) blk[blockIdx.x * 4 + threadIdx.x] = s_c[threadIdx.x];
}

// Clifford fast path: analyze (done by the caller) -> classify+count -> scan -> write, one host round trip at the very end.
static int rotate_fast_clifford(symgpu_op_t in, const u64 *q_dev, const u64 *q_host, const u32 *anti, const uint8_t *ph, int k, double thr,
                                symgpu_op_t *out, int *all_commute, int *done) {
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq;
    *done = 0;
    if (T > ((i64)1 << 22)) return SYMGPU_OK;                  // block-count array of the 2-launch scan: <= 4096 blocks
    Scratch selfc, prodc, cls, pself, pnew, cnt, blk;
    const int n_blk = (int)((T + 1023) / 1024);
    SG_TRY(blk.alloc((size_t)n_blk * 16));
    SG_TRY(selfc.alloc((size_t)T * 16));
    SG_TRY(prodc.alloc((size_t)T * 16));
    SG_TRY(cls.alloc((size_t)T));
    SG_TRY(pself.alloc((size_t)T * 4));
    SG_TRY(pnew.alloc((size_t)T * 4));
    SG_TRY(cnt.alloc(sizeof(RotCounts)));
    hipLaunchKernelGGL(k_rotc_classify, dim3(n_blk), dim3(1024), 0, st, anti, ph, in->coeff, T, k, thr, cls.as<uint8_t>(), selfc.as<double>(),
                       prodc.as<double>(), blk.as<u32>());
    RotCounts *hcnt = nullptr, *hcnt_dev = nullptr;
    SG_TRY(host_counts(&hcnt, &hcnt_dev));
    hipLaunchKernelGGL(k_rotf_scan3, dim3(n_blk), dim3(1024), 0, st, cls.as<uint8_t>(), T, blk.as<u32>(), n_blk, pself.as<u32>(), pnew.as<u32>(),
                       cnt.as<RotCounts>(), (const u32 *)nullptr, 0u, hcnt_dev);
    KERNEL_CHECK();
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(T, Wq, 1, &res));                   // a Clifford rotation never adds rows
    // row hashes, if the operand carries them, are handed on (rotated rows: h ^ h(Q)) so that a later non-Clifford rotation or
    // duplicate check of the chain does not hash again
    const u64 *in_hash = (in->hash && ctx().hash_tab && in->hash_seed == ctx().hash_seed) ? in->hash : nullptr;
    u64 hq = 0;
    if (in_hash) {
        hq = host_row_hash(q_host, 2 * Wq);
        const int rc = dev_alloc((size_t)res->capacity * 8 + 16, (void **)&res->hash);
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        res->hash_seed = in->hash_seed;
    }
    hipLaunchKernelGGL(k_rotf_write, dim3(grid_for((T * Wq + 3) / 4) + (unsigned)((T + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(in->rows),
                       reinterpret_cast<const u32x4 *>(q_dev), T, Wq, cls.as<uint8_t>(), pself.as<u32>(), pnew.as<u32>(), cnt.as<RotCounts>(),
                       selfc.as<double>(), prodc.as<double>(), reinterpret_cast<u32x4 *>(res->rows), res->coeff, 1, in_hash, hq, res->hash);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "rotate Clifford fast path", __FILE__, __LINE__); }
    const RotCounts hc = *hcnt;
    *done = 1;
    if (hc.nAnti == 0) { symgpu_op_free(res); *all_commute = 1; return SYMGPU_OK; }   // identity action (base.py:1131-1133)
    res->T = (i64)hc.nC + hc.nA + hc.nN;
    res->dup_free = in->dup_free;      // distinct rows stay distinct: P^Q anticommutes with Q, so it never meets a commuting row
    *out = res;
    *all_commute = 0;
    return SYMGPU_OK;
}

// The persistent join table: at least 4 slots per row, a fresh generation per call (cleared when the 10-bit generation wraps).
int join_table_for(i64 T, JoinTable *jt) {
    Context &c = ctx();
    size_t cap = 1024;
    while ((i64)cap < 4 * T) cap <<= 1;
    if (!c.rot_flags) {
        HIP_TRY(hipMalloc((void **)&c.rot_flags, 16));
        HIP_TRY(hipMemsetAsync(c.rot_flags, 0, 16, c.stream));
    }
    if (cap > c.rot_table_cap) {
        // from the library's allocator (arena): an operator that grows from rotation to rotation outgrows the table several times, and
        // a hipFree + hipMalloc pair costs 0.3 ms each time; everything that touches the table is on the one library stream
        if (c.rot_table) { dev_free(c.rot_table); c.rot_table = nullptr; c.rot_table_cap = 0; }
        SG_TRY(dev_alloc(cap * 8, (void **)&c.rot_table));
        c.rot_table_cap = cap;
        c.rot_gen = 0;
    }
    if (c.rot_gen == 0 || c.rot_gen >= 1023) {                 // new table, or the generation field wraps: every slot empty again
        HIP_TRY(hipMemsetAsync(c.rot_table, 0, c.rot_table_cap * 8, c.stream));
        HIP_TRY(hipMemsetAsync(c.rot_flags, 0, 16, c.stream));
        c.rot_gen = 0;
    }
    ++c.rot_gen;
    jt->slots = c.rot_table;
    jt->mask = (u32)(cap - 1);                                 // a call uses the first `cap` slots of a possibly larger table
    jt->gen = c.rot_gen;
    jt->flags = c.rot_flags;
    return SYMGPU_OK;
}

// flags + phase exponents of every row; with `jt` also the join-table insert (duplicate detection included), for which the row
// hashes are taken from the handle or computed now and cached on it
static int analyze_rows(symgpu_op_t in, u64 *q_dev, u32 *anti, uint8_t *ph, const JoinTable *jt, const u64 *q_host_arg = nullptr) {
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq;
    int G = 1;
    while (G < Wq && G < 64) G <<= 1;
    const int rpb = 256 / G;
    // word kernel: grid-stride over <= 1024 blocks (uncapped measured 20.6 / 14.2 us against 22.4 / 10.8 us with / without the insert)
    const i64 cap = [] { const char *e = SG_TUNE("SYMGPU_ROT_ANALYZE_CAP"); return e ? atoll(e) : (i64)1024; }();
    i64 g = (T + rpb - 1) / rpb;
    if (g > cap) g = cap;
    const JoinTable none = {nullptr, 0, 0, nullptr};
    QArg qa;
    const bool by_arg = q_host_arg != nullptr && 2 * Wq <= 64;           // Q travels in the kernel arguments (see k_rot_analyze)
    if (by_arg) for (int w = 0; w < 2 * Wq; ++w) qa.w[w] = q_host_arg[w];
    else if (q_host_arg) HIP_TRY(hipMemcpyAsync(q_dev, q_host_arg, (size_t)2 * Wq * 8, hipMemcpyHostToDevice, st));
#define LAUNCH_AN(H, I, HT, HO, HI, J) do { if (by_arg) hipLaunchKernelGGL((k_rot_analyze<H, I, true>), dim3((unsigned)g), dim3(256), 0, st, in->rows, T, Wq, G, q_dev, anti, ph, HT, HO, HI, J, qa); \
                                           else hipLaunchKernelGGL((k_rot_analyze<H, I, false>), dim3((unsigned)g), dim3(256), 0, st, in->rows, T, Wq, G, q_dev, anti, ph, HT, HO, HI, J, qa); } while (0)
    const bool have_hash = jt && in->hash && in->hash_seed == ctx().hash_seed;
    const bool chunks_on = [] { const char *e = SG_TUNE("SYMGPU_ROT_CHUNKS"); return !(e && e[0] == '0'); }();
    if (chunks_on && (!jt || have_hash) && Wq <= 64 && (Wq & (Wq - 1)) == 0) {
        // one 16-byte chunk per lane (k_rot_analyze_chunks); rows of other lengths and the launch that also hashes keep the word kernel
        const u32x4 *pr = reinterpret_cast<const u32x4 *>(in->rows);
        const i64 gb = (T + 256 / Wq - 1) / (256 / Wq), gi = (T + 255) / 256;     // analysis blocks, then (with a join table) insert blocks
#define LAUNCH_CH(WQV) do { \
            if (jt) { if (by_arg) hipLaunchKernelGGL((k_rot_analyze_chunks<WQV, true, true>), dim3((unsigned)(gb + gi)), dim3(256), 0, st, pr, T, q_dev, anti, ph, in->hash, *jt, qa); \
                      else hipLaunchKernelGGL((k_rot_analyze_chunks<WQV, true, false>), dim3((unsigned)(gb + gi)), dim3(256), 0, st, pr, T, q_dev, anti, ph, in->hash, *jt, qa); } \
            else { if (by_arg) hipLaunchKernelGGL((k_rot_analyze_chunks<WQV, false, true>), dim3((unsigned)gb), dim3(256), 0, st, pr, T, q_dev, anti, ph, (const u64 *)nullptr, none, qa); \
                   else hipLaunchKernelGGL((k_rot_analyze_chunks<WQV, false, false>), dim3((unsigned)gb), dim3(256), 0, st, pr, T, q_dev, anti, ph, (const u64 *)nullptr, none, qa); } } while (0)
        switch (Wq) {
            case 1: LAUNCH_CH(1); break;
            case 2: LAUNCH_CH(2); break;
            case 4: LAUNCH_CH(4); break;
            case 8: LAUNCH_CH(8); break;
            case 16: LAUNCH_CH(16); break;
            case 32: LAUNCH_CH(32); break;
            default: LAUNCH_CH(64); break;
        }
#undef LAUNCH_CH
    } else if (!jt) {
        LAUNCH_AN(false, false, (const u64 *)nullptr, (u64 *)nullptr, (const u64 *)nullptr, none);
    } else if (have_hash) {
        LAUNCH_AN(false, true, (const u64 *)nullptr, (u64 *)nullptr, in->hash, *jt);
    } else {
        if (in->hash) { dev_free(in->hash); in->hash = nullptr; }
        SG_TRY(dev_alloc((size_t)in->capacity * 8 + 16, (void **)&in->hash));      // cached on the operand: its next rotation skips the hashing
        in->hash_seed = ctx().hash_seed;
        LAUNCH_AN(true, true, ctx().hash_tab, in->hash, (const u64 *)nullptr, *jt);
    }
#undef LAUNCH_AN
    KERNEL_CHECK();
    return SYMGPU_OK;
}

// Non-Clifford rotation as a hash join, four launches and no clearing pass:
//   k_rot_analyze<.., INSERT>  flags + phase exponents (+ row hashes unless the handle carries them) + insert into the join table
//   k_rotf_match2              probe h(P) ^ h(Q), verify, classify, coefficients, per-block counts
//   k_rotf_scan3, k_rotf_write output slots, rows + coefficients + hashes of the result
// returns SYMGPU_OK with *done = 1 (result in *out / *all_commute) or *done = 0 (duplicate rows: use the general path)
static int rotate_fast_nonclifford(symgpu_op_t in, u64 *q_dev, const u64 *q_host, bool q_pending, u32 *anti, uint8_t *ph, double cos_t,
                                   double sin_t, double thr, symgpu_op_t *out, int *all_commute, int *done) {
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq, W = 2 * Wq;
    *done = 0;
    if (T >= ((i64)1 << 22) - 1) return SYMGPU_OK;             // 22-bit row index of a table entry; block-count array of the scan
    SG_TRY(ensure_hash_tables(ctx().hash_tab ? ctx().hash_seed : 1));
    const u64 seed = ctx().hash_seed;
    const u64 hq = host_row_hash(q_host, W);
    JoinTable jt;
    SG_TRY(join_table_for(T, &jt));
    Scratch selfc, prodc, cls, pself, pnew, cnt, blk;
    const int n_blk = (int)((T + 1023) / 1024);
    SG_TRY(blk.alloc((size_t)n_blk * 16));
    SG_TRY(selfc.alloc((size_t)T * 16));
    SG_TRY(prodc.alloc((size_t)T * 16));
    SG_TRY(cls.alloc((size_t)T));
    SG_TRY(pself.alloc((size_t)T * 4));
    SG_TRY(pnew.alloc((size_t)T * 4));
    SG_TRY(cnt.alloc(sizeof(RotCounts)));
    SG_TRY(analyze_rows(in, q_dev, anti, ph, &jt, q_pending ? q_host : nullptr));    // Q reaches the device with this launch
    hipLaunchKernelGGL(k_rotf_match2, dim3(n_blk), dim3(1024), 0, st, in->rows, in->coeff, in->hash, T, W, q_dev, hq, anti, ph, jt, cos_t, sin_t, thr,
                       selfc.as<double>(), prodc.as<double>(), cls.as<uint8_t>(), blk.as<u32>());
    RotCounts *hcnt = nullptr, *hcnt_dev = nullptr;
    SG_TRY(host_counts(&hcnt, &hcnt_dev));
    hipLaunchKernelGGL(k_rotf_scan3, dim3(n_blk), dim3(1024), 0, st, cls.as<uint8_t>(), T, blk.as<u32>(), n_blk, pself.as<u32>(), pnew.as<u32>(),
                       cnt.as<RotCounts>(), jt.flags, jt.gen, hcnt_dev);
    KERNEL_CHECK();
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(2 * T, Wq, 1, &res));               // upper bound: no host round trip before the write kernel
    int rc = dev_alloc((size_t)res->capacity * 8 + 16, (void **)&res->hash);
    if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
    res->hash_seed = seed;
    hipLaunchKernelGGL(k_rotf_write, dim3(grid_for((T * Wq + 3) / 4) + (unsigned)((T + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(in->rows),
                       reinterpret_cast<const u32x4 *>(q_dev), T, Wq, cls.as<uint8_t>(), pself.as<u32>(), pnew.as<u32>(), cnt.as<RotCounts>(),
                       selfc.as<double>(), prodc.as<double>(), reinterpret_cast<u32x4 *>(res->rows), res->coeff, 0, in->hash, hq, res->hash);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "rotate fast path", __FILE__, __LINE__); }
    const RotCounts hc = *hcnt;
    if (hc.nAnti == 0) { symgpu_op_free(res); if (!hc.dup) in->dup_free = 1; *all_commute = 1; *done = 1; return SYMGPU_OK; }   // identity action (base.py:1131-1133)
    if (hc.dup) { symgpu_op_free(res); return SYMGPU_OK; }     // duplicates in the input: general path
    res->T = (i64)hc.nC + hc.nA + hc.nN;
    res->dup_free = 1;                 // the input had no duplicates (checked above) and every P^Q that met a row was merged into it
    in->dup_free = 1;
    *out = res;
    *all_commute = 0;
    *done = 1;
    return SYMGPU_OK;
}

// ---- a whole chain of Clifford rotations in ONE launch (small operators) ------------------------------------------------------
// perform_rotations / CircuitSymmerlator apply thousands of pi/2-multiples to an operator of a few (hundred) terms: per rotation the
// launches and the count read-back of the general path cost 60 us, the data movement nothing.  For an operator that is CLEAN —
// no duplicate rows and every |c| > 1e-15, i.e. it has been through cleanup(), which is what perform_rotations guarantees after
// its first step — a Clifford rotation drops nothing and merges nothing (base.py:1139-1154 followed by cleanup(), :1185): it is
// a stable partition [anticommuting | commuting] with row ^= Q and c *= i^e (-i) (odd k), c = -c (k in {2,3}) on the anticommuting
// part, or nothing at all if every term commutes.  One workgroup keeps flags, phase exponents and slots in LDS and ping-pongs
// the rows between two global buffers (L2 resident), one block barrier per phase.
constexpr int CHAIN_TMAX = 8192;               // rows the single-workgroup kernel can hold: 8 per thread of the slot scan
constexpr int CHAIN_LOCAL_T = 128;             // ... and up to where it beats the multi-workgroup kernels: 3.5 us per rotation at 1 row, 5.0 at
                                               // 64, 6.6 at 128, 9.9 at 256, 29 at 1,000, against 6.7-7 us of the two-launch form (below)
constexpr int CHAIN_TWO_T = 262144;            // two launches per rotation (k_cchain_flags / k_cchain_move) up to here (256 group counts): 6.7 us at
                                               // 64-384 rows, 7.1 at 1,000, 10.4 at 8,000, 13.4 / 20.8 / 26.7 at 16,384 / 65,536 / 131,072 against 15.6
                                               // us (launch-rate bound) up to 8,000 rows and 17.3 / 24.0 / 32.4 us of the four-launch form

__global__ __launch_bounds__(1024) void k_clifford_chain(u64 *__restrict__ rowsA, double *__restrict__ coeffA, u64 *__restrict__ rowsB,
                                                          double *__restrict__ coeffB, int T, int Wq, int G, const u64 *__restrict__ qs,
                                                          const int *__restrict__ ks, int K, int *__restrict__ result_in_b) {
    __shared__ uint8_t s_anti[CHAIN_TMAX], s_ph[CHAIN_TMAX];
    __shared__ u32 s_pos[CHAIN_TMAX];
    __shared__ u32 s_wsum[16], s_total;
    const int W = 2 * Wq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G, rows_per_pass = 1024 / G;
    u64 *cur_r = rowsA, *nxt_r = rowsB;
    double *cur_c = coeffA, *nxt_c = coeffB;
    int in_b = 0;
    for (int r = 0; r < K; ++r) {
        const u64 *q = qs + (i64)r * W;
        const int k = ks[r];
        // ---- phase 1: anticommutation flag and phase exponent of every row (G lanes per row, as k_rot_analyze) ----
        int yq = 0;
        for (int w = 0; w < Wq; ++w) yq += __popcll(q[w] & q[Wq + w]);
        for (int t0 = 0; t0 < T; t0 += rows_per_pass) {
            const int t = t0 + rsub;
            u64 par = 0, flip = 0;
            int yp = 0, yout = 0;
            if (t < T) {
                const u64 *row = cur_r + (i64)t * W;
                for (int w = g; w < Wq; w += G) {
                    const u64 x = row[w], z = row[Wq + w], xq = q[w], zq = q[Wq + w];
                    par ^= (x & zq) ^ (z & xq);
                    flip ^= x & zq;
                    yp += __popcll(x & z);
                    yout += __popcll((x ^ xq) & (z ^ zq));
                }
            }
            int pp = __popcll(par) & 1, fp = __popcll(flip) & 1;
            for (int off = G >> 1; off > 0; off >>= 1) {
                pp ^= __shfl_xor(pp, off);
                fp ^= __shfl_xor(fp, off);
                yp += __shfl_xor(yp, off);
                yout += __shfl_xor(yout, off);
            }
            if (g == 0 && t < T) {
                s_anti[t] = (uint8_t)pp;
                s_ph[t] = (uint8_t)((3 * (yp + yq) + yout + 2 * fp) & 3);
            }
        }
        __syncthreads();
        // ---- phase 2: slots of the stable partition [anticommuting | commuting]: thread i owns rows 8i .. 8i+7 ----
        u32 mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int t = 8 * (int)threadIdx.x + j; if (t < T) mine += s_anti[t]; }
        u32 incl = mine;
        for (int off = 1; off < 64; off <<= 1) { const u32 v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) { u32 acc = 0; for (int w2 = 0; w2 < 16; ++w2) { const u32 v = s_wsum[w2]; s_wsum[w2] = acc; acc += v; } s_total = acc; }
        __syncthreads();
        const u32 n_anti = s_total;
        if (n_anti == 0) { __syncthreads(); continue; }              // every term commutes with Q: identity (base.py:1131-1133)
        {
            u32 a_before = s_wsum[wave] + incl - mine;                // anticommuting rows before row 8i
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 8 * (int)threadIdx.x + j;
                if (t < T) {
                    const bool a = s_anti[t];
                    s_pos[t] = a ? a_before : n_anti + ((u32)t - a_before);
                    a_before += a;
                }
            }
        }
        __syncthreads();
        // ---- phase 3: rows and coefficients to their slots in the other buffer ----
        for (int t0 = 0; t0 < T; t0 += rows_per_pass) {
            const int t = t0 + rsub;
            if (t < T) {
                const bool a = s_anti[t];
                const bool flipq = a && (k & 1);
                const u64 *row = cur_r + (i64)t * W;
                u64 *dst = nxt_r + (i64)s_pos[t] * W;
                for (int w = g; w < W; w += G) dst[w] = row[w] ^ (flipq ? q[w] : 0ULL);
                if (g == 0) {
                    double re = cur_c[2 * t], im = cur_c[2 * t + 1];
                    if (a) {
                        if (k & 1) { double x, y; phase_mul(re, im, s_ph[t], x, y); re = y; im = -x; }     // c * i^e * (-i)
                        if (k == 2 || k == 3) { re = -re; im = -im; }
                    }
                    nxt_c[2 * s_pos[t]] = re; nxt_c[2 * s_pos[t] + 1] = im;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        { u64 *tr = cur_r; cur_r = nxt_r; nxt_r = tr; double *tc = cur_c; cur_c = nxt_c; nxt_c = tc; in_b ^= 1; }
    }
    if (threadIdx.x == 0) *result_in_b = in_b;
}

// The same run with the operator resident in LDS (<= 128 rows and <= 64 KiB of rows: the circuit simulator's observables).  The
// global version above goes through L2 twice per rotation (rows written by the previous rotation are read back by other waves of
// the workgroup: 4.9 us per rotation at 64 rows of 256 bytes); here rows and coefficients ping-pong between two LDS buffers, Q of
// the NEXT rotation is fetched into registers while the current one runs, and a rotation is three barriers.
constexpr int CHAIN_LDS_T = 128;
__global__ __launch_bounds__(1024) void k_clifford_chain_lds(u64 *__restrict__ rows, double *__restrict__ coeff, int T, int Wq, int G,
                                                              const u64 *__restrict__ qs, const int *__restrict__ ks, int K) {
    extern __shared__ __attribute__((aligned(16))) u64 lds_dyn[];    // [2][T * W] rows
    __shared__ double s_c[2][2 * CHAIN_LDS_T];
    __shared__ u64 s_q2[2][128];                                     // Q of this rotation / of the next one (written before the barrier
                                                                      // that ends the previous rotation's row move, which still reads its Q)
    __shared__ uint8_t s_anti[CHAIN_LDS_T], s_ph[CHAIN_LDS_T];
    __shared__ u32 s_pos[CHAIN_LDS_T];
    __shared__ u32 s_cnt[2];
    const int W = 2 * Wq, n_words = T * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G, rows_per_pass = 1024 / G;
    u64 *buf0 = lds_dyn, *buf1 = lds_dyn + n_words;
    for (int i = threadIdx.x; i < n_words; i += 1024) buf0[i] = rows[i];
    for (int i = threadIdx.x; i < 2 * T; i += 1024) s_c[0][i] = coeff[i];
    int cur = 0;
    u64 q_next = (K > 0 && (int)threadIdx.x < W) ? qs[threadIdx.x] : 0ULL;
    for (int r = 0; r < K; ++r) {
        const int k = ks[r];
        u64 *s_q = s_q2[r & 1];
        if ((int)threadIdx.x < W) s_q[threadIdx.x] = q_next;
        __syncthreads();                                              // Q in place; the previous rotation's buffers complete
        if (r + 1 < K && (int)threadIdx.x < W) q_next = qs[(i64)(r + 1) * W + threadIdx.x];   // in flight during this rotation
        u64 *cr = cur ? buf1 : buf0, *nr = cur ? buf0 : buf1;
        double *cc = s_c[cur], *ncf = s_c[cur ^ 1];
        // ---- flags + phase exponents (G lanes per row; the Y count of Q is formed along the way by the same lanes) ----
        for (int t0 = 0; t0 < T; t0 += rows_per_pass) {
            const int t = t0 + rsub;
            u64 par = 0, flip = 0;
            int yp = 0, yout = 0, yq = 0;
            if (t < T) {
                const u64 *row = cr + t * W;
                for (int w = g; w < Wq; w += G) {
                    const u64 x = row[w], z = row[Wq + w], xq = s_q[w], zq = s_q[Wq + w];
                    par ^= (x & zq) ^ (z & xq);
                    flip ^= x & zq;
                    yp += __popcll(x & z);
                    yout += __popcll((x ^ xq) & (z ^ zq));
                    yq += __popcll(xq & zq);
                }
            }
            int pp = __popcll(par) & 1, fp = __popcll(flip) & 1;
            for (int off = G >> 1; off > 0; off >>= 1) {
                pp ^= __shfl_xor(pp, off);
                fp ^= __shfl_xor(fp, off);
                yp += __shfl_xor(yp, off);
                yout += __shfl_xor(yout, off);
                yq += __shfl_xor(yq, off);
            }
            if (g == 0 && t < T) {
                s_anti[t] = (uint8_t)pp;
                s_ph[t] = (uint8_t)((3 * (yp + yq) + yout + 2 * fp) & 3);
            }
        }
        __syncthreads();
        // ---- slots of the stable partition [anticommuting | commuting]: waves 0 and 1, one row per lane ----
        if (wave < 2) {
            const int t = wave * 64 + lane;
            const bool a = t < T && s_anti[t];
            const u64 bal = __ballot(a);
            if (lane == 0) s_cnt[wave] = (u32)__popcll(bal);
            s_pos[t < CHAIN_LDS_T ? t : 0] = (u32)__popcll(bal & ((1ULL << lane) - 1ULL));      // rank inside the wave, completed below
        }
        __syncthreads();
        const u32 n0 = s_cnt[0], n_anti = n0 + s_cnt[1];
        if (n_anti == 0) continue;                                    // every term commutes with Q: identity (base.py:1131-1133)
        // ---- rows and coefficients to their slots in the other buffer ----
        for (int t0 = 0; t0 < T; t0 += rows_per_pass) {
            const int t = t0 + rsub;
            if (t < T) {
                const bool a = s_anti[t];
                const u32 a_before = s_pos[t] + (t >= 64 ? n0 : 0u);
                const u32 pos = a ? a_before : n_anti + ((u32)t - a_before);
                const bool flipq = a && (k & 1);
                const u64 *row = cr + t * W;
                u64 *dst = nr + pos * W;
                for (int w = g; w < W; w += G) dst[w] = row[w] ^ (flipq ? s_q[w] : 0ULL);
                if (g == 0) {
                    double re = cc[2 * t], im = cc[2 * t + 1];
                    if (a) {
                        if (k & 1) { double x, y; phase_mul(re, im, s_ph[t], x, y); re = y; im = -x; }     // c * i^e * (-i)
                        if (k == 2 || k == 3) { re = -re; im = -im; }
                    }
                    ncf[2 * pos] = re; ncf[2 * pos + 1] = im;
                }
            }
        }
        cur ^= 1;                                                     // the barrier at the top of the next rotation completes the move
    }
    __syncthreads();
    const u64 *fr = cur ? buf1 : buf0;
    for (int i = threadIdx.x; i < n_words; i += 1024) rows[i] = fr[i];
    for (int i = threadIdx.x; i < 2 * T; i += 1024) coeff[i] = s_c[cur][i];
}


// ---- a run of Clifford rotations of a CLEAN operator, two launches per rotation -----------------------------------------------------
// Back-to-back launches are launch-rate bound (3.9 us per launch, four per rotation: 15.6 us at any size up to 8,000 rows).  For a
// clean operator a Clifford rotation is just a stable partition [anticommuting | commuting] with new coefficients, so two launches
// do: (A) flags, phase exponents, the new coefficient of every row and the anticommuting count of every 1024-row group, one 16-byte
// chunk per lane as in k_rot_analyze_chunks; (B) every block derives the output slots of its own rows — groups before it from the
// <= 256 group counts, rows before it inside its group from their byte flags — and moves rows and coefficients.  The group counts
// ping-pong between two arrays (B of rotation r clears the array A of rotation r+1 adds to).  Rows of a power-of-two number of chunks,
// <= CHAIN_TWO_T rows; everything else keeps the four-launch form.
template <int WQ, int CCH_ROWS>
__global__ __launch_bounds__(256) void k_cchain_flags(const u32x4 *__restrict__ rows, const double *__restrict__ coeff, i64 T, const u64 *__restrict__ q_dev,
                                                       int k, uint8_t *__restrict__ af, double *__restrict__ nc, u32 *__restrict__ cnt) {
    __shared__ __attribute__((aligned(16))) u64 sq[2 * WQ];
    __shared__ int s_yq;
    __shared__ u32 s_n;
    if ((int)threadIdx.x < 2 * WQ) sq[threadIdx.x] = q_dev[threadIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (threadIdx.x < 64) {                                          // Y count of Q: one wavefront
        int y = 0;
        for (int w = threadIdx.x; w < WQ; w += 64) y += __popcll(sq[w] & sq[WQ + w]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) y += __shfl_xor(y, off);
        if (threadIdx.x == 0) s_yq = y;
    }
    __syncthreads();
    constexpr int R = 256 / WQ, ITER = CCH_ROWS / R;                 // CCH_ROWS rows per block: the block's fixed costs are paid once
    const int c = threadIdx.x & (WQ - 1);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const i64 t = (i64)blockIdx.x * CCH_ROWS + it * R + threadIdx.x / WQ;
        const bool valid = t < T;
        const u32x4 v = valid ? rows[t * WQ + c] : (u32x4)(0u);
        u32 par, ye;
        if constexpr (WQ == 1) {
            const u32x4 qv = *reinterpret_cast<const u32x4 *>(sq);
            const u32 f = __popc(v.x & qv.z) + __popc(v.y & qv.w);
            par = f + __popc(v.z & qv.x) + __popc(v.w & qv.y) + (f << 16);
            ye = (__popc(v.x & v.z) + __popc(v.y & v.w)) | ((__popc((v.x ^ qv.x) & (v.z ^ qv.z)) + __popc((v.y ^ qv.y) & (v.w ^ qv.w))) << 16);
        } else {
            const u32x4 qs = reinterpret_cast<const u32x4 *>(sq)[c], qo = reinterpret_cast<const u32x4 *>(sq)[c ^ (WQ / 2)];
            const bool xhalf = c < WQ / 2;
            const u32 p = __popc(v.x & qo.x) + __popc(v.y & qo.y) + __popc(v.z & qo.z) + __popc(v.w & qo.w);
            const u32x4 o = {rot_other_half<WQ>(v.x), rot_other_half<WQ>(v.y), rot_other_half<WQ>(v.z), rot_other_half<WQ>(v.w)};
            const u32 yp = __popc(v.x & o.x) + __popc(v.y & o.y) + __popc(v.z & o.z) + __popc(v.w & o.w);
            const u32 yo = __popc((v.x ^ qs.x) & (o.x ^ qo.x)) + __popc((v.y ^ qs.y) & (o.y ^ qo.y)) + __popc((v.z ^ qs.z) & (o.z ^ qo.z)) +
                           __popc((v.w ^ qs.w) & (o.w ^ qo.w));
            par = rot_row_sum<WQ>(p + (xhalf ? (p << 16) : 0u));
            ye = rot_row_sum<WQ>(xhalf ? (yp | (yo << 16)) : 0u);
        }
        if (c == 0 && valid) {
            const bool anti = par & 1u;
            const int e = (int)((3u * ((ye & 0xFFFFu) + (u32)s_yq) + (ye >> 16) + 2u * ((par >> 16) & 1u)) & 3u);
            double re = coeff[2 * t], im = coeff[2 * t + 1];
            if (anti) {
                if (k & 1) {                                          // c * i^e * (-i), negated for k = 3 (k_rotc_classify)
                    double x, y;
                    phase_mul(re, im, e, x, y);
                    re = y; im = -x;
                    if (k == 3) { re = -re; im = -im; }
                } else if (k == 2) { re = -re; im = -im; }
                atomicAdd(&s_n, 1u);
            }
            af[t] = anti ? 1 : 0;
            nc[2 * t] = re; nc[2 * t + 1] = im;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&cnt[((i64)blockIdx.x * CCH_ROWS) >> 10], s_n);   // CCH_ROWS divides 1024: a block lies inside one group
}

template <int WQ, int CCH_ROWS>
__global__ __launch_bounds__(256) void k_cchain_move(const u32x4 *__restrict__ rows, i64 T, const u64 *__restrict__ q_dev, int k,
                                                      const uint8_t *__restrict__ af, const double *__restrict__ nc, const u32 *__restrict__ cnt,
                                                      int n_cnt, u32 *__restrict__ cnt_next, u32x4 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    constexpr int R = 256 / WQ, ITER = CCH_ROWS / R;
    __shared__ u32 s_sum[3];                                          // anticommuting rows: in the groups before, in all groups, before the block inside its group
    __shared__ uint8_t s_flag[CCH_ROWS];
    __shared__ u32 s_pos[CCH_ROWS];
    if (threadIdx.x < 3) s_sum[threadIdx.x] = 0;
    __syncthreads();
    const i64 t0 = (i64)blockIdx.x * CCH_ROWS;
    const int g = (int)(t0 >> 10);
    const i64 g0 = (i64)g << 10;
    u32 before = 0, all = 0, inside = 0;
    if ((int)threadIdx.x < n_cnt) {
        const u32 x = cnt[threadIdx.x];
        all = x;
        if ((int)threadIdx.x < g) before = x;
    }
    {   // byte flags of the rows [g0, t0): at most 1020 bytes, one u32 per thread (t0 - g0 is a multiple of CCH_ROWS >= 4)
        const i64 w = g0 + 4 * (i64)threadIdx.x;
        if (w < t0) inside = (*reinterpret_cast<const u32 *>(af + w) * 0x01010101u) >> 24;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        before += (u32)__shfl_xor((int)before, off); all += (u32)__shfl_xor((int)all, off); inside += (u32)__shfl_xor((int)inside, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_sum[0], before); atomicAdd(&s_sum[1], all); atomicAdd(&s_sum[2], inside); }
    if (blockIdx.x == 0 && (int)threadIdx.x < n_cnt) cnt_next[threadIdx.x] = 0;      // the array the next rotation's flags kernel adds to
    if ((int)threadIdx.x < CCH_ROWS) s_flag[threadIdx.x] = (t0 + threadIdx.x < T) ? af[t0 + threadIdx.x] : 0;
    __syncthreads();
    if ((int)threadIdx.x < CCH_ROWS && t0 + threadIdx.x < T) {
        // rank among the block's rows: the flags of the same wavefront's lower lanes (CCH_ROWS <= 256: up to 4 wavefronts)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const u64 bal = __ballot(s_flag[threadIdx.x] != 0);
        u32 rank = (u32)__popcll(bal & ((1ULL << lane) - 1ULL));
        for (int w2 = 0; w2 < wv; ++w2)
            for (int r = 0; r < 64; ++r) rank += s_flag[w2 * 64 + r];
        const u32 a_before = s_sum[0] + s_sum[2] + rank;              // anticommuting rows before this one
        const i64 t = t0 + threadIdx.x;
        const u32 pos = s_flag[threadIdx.x] ? a_before : s_sum[1] + (u32)(t - a_before);
        s_pos[threadIdx.x] = pos;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        reinterpret_cast<f64x2 *>(out_coeff)[pos] = reinterpret_cast<const f64x2 *>(nc)[t];
    }
    __syncthreads();
    const int c = threadIdx.x & (WQ - 1);
    const u32x4 qc = reinterpret_cast<const u32x4 *>(q_dev)[c];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int rl = it * R + threadIdx.x / WQ;
        const i64 t = t0 + rl;
        if (t < T) {
            u32x4 v = rows[t * WQ + c];
            if (s_flag[rl] && (k & 1)) v ^= qc;
            out_rows[(i64)s_pos[rl] * WQ + c] = v;
        }
    }
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_rotate_single_dev_n(symgpu_op_t in, const uint64_t *q_row_host, double cos_t, double sin_t, int clifford_k, double thr,
                               symgpu_op_t *out, int *all_commute, int64_t *n_out) {
    SG_REQUIRE(n_out, "rotate_single_dev_n: null argument");
    *n_out = 0;
    SG_TRY(symgpu_rotate_single_dev(in, q_row_host, cos_t, sin_t, clifford_k, thr, out, all_commute));
    if (*out) *n_out = (*out)->T;
    return SYMGPU_OK;
}

int symgpu_rotate_single_dev(symgpu_op_t in, const uint64_t *q_row_host, double cos_t, double sin_t, int clifford_k, double thr,
                             symgpu_op_t *out, int *all_commute) {
    SG_ENTER(in);
    SG_REQUIRE(in && q_row_host && out && all_commute, "rotate_single_dev: null argument");
    SG_REQUIRE(in->coeff || in->T == 0, "rotate_single_dev: operator has no coefficients");
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq, W = 2 * Wq;
    *out = nullptr;
    *all_commute = 1;
    if (T == 0) return SYMGPU_OK;
    SG_REQUIRE(T < ((i64)1 << 31), "rotate_single_dev: too many rows");
    if (!getenv("SYMGPU_ROTATE_GENERAL")) {
        // one persistent launch with the rows resident in LDS (rotate_resident.hip): duplicate-free operators that fit the chip's LDS
        int done = 0;
        SG_TRY(rotate_resident_try(in, q_row_host, cos_t, sin_t, clifford_k, thr, out, all_commute, &done));
        if (done) return SYMGPU_OK;
        *out = nullptr;
        *all_commute = 1;
    }
    Scratch q, anti, sel, apos, cpos, ph, totals, dmain, dprod;
    SG_TRY(q.alloc((size_t)W * 8));
    SG_TRY(anti.alloc((size_t)T * 4));
    SG_TRY(sel.alloc((size_t)T * 4));
    SG_TRY(apos.alloc((size_t)T * 4));
    SG_TRY(cpos.alloc((size_t)T * 4));
    SG_TRY(ph.alloc((size_t)T));
    SG_TRY(totals.alloc(16));
    SG_TRY(dmain.alloc((size_t)T * 4));
    SG_TRY(dprod.alloc((size_t)T * 4));
    // Q is uploaded by the first analyze launch of the call (in its kernel arguments when the row has <= 64 words), see analyze_rows
    bool q_pending = true;
    const bool clifford = clifford_k >= 0;
    const bool general_only = getenv("SYMGPU_ROTATE_GENERAL") != nullptr;
    const bool try_fast = !clifford && !general_only;
    // odd multiples of pi/2 multiply by Q through the reference's __mul__ (merge + threshold on sums): the per-row fast path is
    // only the same thing for an operator without duplicate rows
    const bool need_dup_check = clifford && (clifford_k & 1) && !in->dup_free && !general_only;
    bool analyzed = false, has_dup = false;
    if (try_fast) {
        int done = 0;
        SG_TRY(rotate_fast_nonclifford(in, q.as<u64>(), q_row_host, q_pending, anti.as<u32>(), ph.as<uint8_t>(), cos_t, sin_t, thr, out, all_commute, &done));
        if (done) return SYMGPU_OK;
        if (T < ((i64)1 << 22) - 1) q_pending = false;              // the join ran: Q is on the device
        *out = nullptr;                                            // duplicate rows, or too many rows for the join: general path;
        *all_commute = 1;                                          // the flags and phase exponents are in place if the join ran
        analyzed = T < ((i64)1 << 22) - 1;
    }
    if (!analyzed) {
        if (need_dup_check && T < ((i64)1 << 22) - 1) {
            // flags + the same join-table insert as the non-Clifford path: does the operator hold two equal rows?
            SG_TRY(ensure_hash_tables(ctx().hash_tab ? ctx().hash_seed : 1));
            JoinTable jt;
            SG_TRY(join_table_for(T, &jt));
            SG_TRY(analyze_rows(in, q.as<u64>(), anti.as<u32>(), ph.as<uint8_t>(), &jt, q_pending ? q_row_host : nullptr));
            q_pending = false;
            u32 hflag = 0;
            SG_TRY(read_back_words(jt.flags, 1, nullptr, 0, &hflag));
            has_dup = hflag == jt.gen;
            if (!has_dup) in->dup_free = 1;
        } else {
            has_dup = need_dup_check;                              // too large for the table: take the merging path
            SG_TRY(analyze_rows(in, q.as<u64>(), anti.as<u32>(), ph.as<uint8_t>(), nullptr, q_pending ? q_row_host : nullptr));
            q_pending = false;
        }
    }
    if (q_pending) HIP_TRY(hipMemcpyAsync(q.p, q_row_host, (size_t)W * 8, hipMemcpyHostToDevice, st));   // (not reached: every path analyzes first)
    if (clifford && !general_only && !has_dup) {
        int done = 0;
        SG_TRY(rotate_fast_clifford(in, q.as<u64>(), q_row_host, anti.as<u32>(), ph.as<uint8_t>(), clifford_k, thr, out, all_commute, &done));
        if (done) return SYMGPU_OK;
        *out = nullptr;
        *all_commute = 1;
    }
    // odd-k Clifford: every anticommuting row enters the product; the threshold applies to the merged sums below
    const int drop_small = 0;
    hipLaunchKernelGGL(k_rot_keepflags, dim3(grid_for(T)), dim3(256), 0, st, anti.as<u32>(), in->coeff, T, thr, drop_small, sel.as<u32>());
    KERNEL_CHECK();
    u32 *tot = totals.as<u32>();
    SG_TRY(exclusive_scan_u32(sel.as<u32>(), apos.as<u32>(), T, tot));            // positions among selected anticommuting rows
    hipLaunchKernelGGL(k_not_flags, dim3(grid_for(T)), dim3(256), 0, st, anti.as<u32>(), T, cpos.as<u32>());
    KERNEL_CHECK();
    SG_TRY(exclusive_scan_u32(cpos.as<u32>(), cpos.as<u32>(), T, tot + 1));        // positions among commuting rows
    u32 h[2] = {0, 0};
    SG_TRY(read_back_words(tot, 2, nullptr, 0, h));
    const i64 n_sel = h[0], n_comm = h[1];
    if (n_comm == T) return SYMGPU_OK;        // every term commutes: identity action (base.py:1131-1133)
    *all_commute = 0;
    const i64 n_stack = clifford ? (n_sel + n_comm) : (n_comm + 2 * n_sel);
    symgpu_op_t stack = nullptr;
    SG_TRY(symgpu_op_alloc(n_stack > 0 ? n_stack : 1, Wq, 1, &stack));
    stack->T = n_stack;
    if (clifford)
        hipLaunchKernelGGL(k_rot_coeff<1>, dim3(grid_for(T)), dim3(256), 0, st, in->coeff, anti.as<u32>(), sel.as<u32>(), apos.as<u32>(),
                           cpos.as<u32>(), ph.as<uint8_t>(), T, n_sel, n_comm, cos_t, sin_t, clifford_k, stack->coeff, dmain.as<u32>(), dprod.as<u32>());
    else
        hipLaunchKernelGGL(k_rot_coeff<0>, dim3(grid_for(T)), dim3(256), 0, st, in->coeff, anti.as<u32>(), sel.as<u32>(), apos.as<u32>(),
                           cpos.as<u32>(), ph.as<uint8_t>(), T, n_sel, n_comm, cos_t, sin_t, clifford_k, stack->coeff, dmain.as<u32>(), dprod.as<u32>());
    hipLaunchKernelGGL(k_rot_rows, dim3(grid_for(T * Wq)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(in->rows),
                       reinterpret_cast<const u32x4 *>(q.p), T, Wq, dmain.as<u32>(), dprod.as<u32>(), reinterpret_cast<u32x4 *>(stack->rows));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { symgpu_op_free(stack); return hip_fail(e, "rotate build", __FILE__, __LINE__); }
    if (clifford && !(clifford_k & 1)) {
        HIP_TRY(hipStreamSynchronize(st));
        stack->dup_free = in->dup_free;
        *out = stack;
        return SYMGPU_OK;
    }
    if (clifford) {
        // (anticom_self * Q) of base.py:1143 = cleanup of the rotated rows: duplicates merged in input order, |sum| > thr kept
        // (the exact factors -i / -1 commute with the IEEE sums); then the commuting rows, untouched (base.py:1151-1154)
        symgpu_op_t merged = nullptr;
        int rc = cleanup_core(stack->rows, stack->coeff, n_sel, W, nullptr, 0, nullptr, 0, thr, 1, &merged, Wq);
        if (rc != SYMGPU_OK) { symgpu_op_free(stack); return rc; }
        symgpu_op_t res = nullptr;
        rc = symgpu_op_alloc(merged->T + n_comm > 0 ? merged->T + n_comm : 1, Wq, 1, &res);
        hipError_t e = hipSuccess;
        if (rc == SYMGPU_OK) {
            if (merged->T > 0) {
                e = hipMemcpyAsync(res->rows, merged->rows, (size_t)merged->T * W * 8, hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(res->coeff, merged->coeff, (size_t)merged->T * 16, hipMemcpyDeviceToDevice, st);
            }
            if (e == hipSuccess && n_comm > 0) {
                e = hipMemcpyAsync(res->rows + (size_t)merged->T * W, stack->rows + (size_t)n_sel * W, (size_t)n_comm * W * 8, hipMemcpyDeviceToDevice, st);
                if (e == hipSuccess) e = hipMemcpyAsync(res->coeff + 2 * (size_t)merged->T, stack->coeff + 2 * (size_t)n_sel, (size_t)n_comm * 16, hipMemcpyDeviceToDevice, st);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            res->T = merged->T + n_comm;
        }
        symgpu_op_free(merged);
        symgpu_op_free(stack);
        if (rc != SYMGPU_OK) return rc;
        if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "rotate Clifford merge", __FILE__, __LINE__); }
        res->dup_free = in->dup_free;                              // merged rotated rows + the untouched commuting rows
        *out = res;
        return SYMGPU_OK;
    }
    symgpu_op_t res = nullptr;
    int rc = cleanup_core(stack->rows, stack->coeff, n_stack, W, nullptr, 0, nullptr, 0, thr, 1, &res, Wq);
    symgpu_op_free(stack);
    if (rc != SYMGPU_OK) return rc;
    *out = res;
    return SYMGPU_OK;
}

// The operator 0 * I (one identity row, coefficient 0): what the reference's cleanup() makes of an operator without terms (base.py:631-632)
static int zero_identity_op(int Wq, symgpu_op_t *out) {
    symgpu_op_t z = nullptr;
    SG_TRY(symgpu_op_alloc(1, Wq, 1, &z));
    hipStream_t st = ctx().stream;
    hipError_t e = hipMemsetAsync(z->rows, 0, (size_t)2 * Wq * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(z->coeff, 0, 16, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { symgpu_op_free(z); return hip_fail(e, "zero_identity_op", __FILE__, __LINE__); }
    z->T = 1;
    *out = z;
    return SYMGPU_OK;
}

// does any row of `op` commute with the row q (host)?  Only asked when a non-Clifford rotation has left no term at all.
static int any_row_commutes(symgpu_op_t op, const u64 *q_host, bool *any) {
    symgpu_op_t q = nullptr;
    SG_TRY(symgpu_op_upload(q_host, nullptr, 1, op->Wq, &q));
    Scratch flags;
    int rc = flags.alloc((size_t)op->T);
    if (rc == SYMGPU_OK) rc = symgpu_commutes_dev(op, 0, op->T, q, flags.as<uint8_t>());
    uint64_t sum = 0;
    if (rc == SYMGPU_OK) rc = symgpu_dev_checksum_u8(flags.as<uint8_t>(), op->T, &sum);
    symgpu_op_free(q);
    *any = sum != 0;
    return rc;
}

int symgpu_perform_rotations_dev(symgpu_op_t in, const uint64_t *q_rows_host, const double *cos_t, const double *sin_t, const int *ks_host, int64_t K,
                                 double thr, int clean, symgpu_op_t *out, uint8_t *acted, int64_t *n_done, int *clean_out) {
    SG_ENTER(in);
    SG_REQUIRE(in && out && n_done && clean_out && K >= 0 && (K == 0 || (q_rows_host && cos_t && sin_t && ks_host)), "perform_rotations_dev: null argument");
    const int W = 2 * in->Wq;
    symgpu_op_t cur = in;                                            // borrowed while cur == in, owned otherwise
    auto replace = [&](symgpu_op_t next) { if (cur != in) symgpu_op_free(cur); cur = next; };
    *out = nullptr;
    i64 step = 0;
    int rc = SYMGPU_OK;
    while (step < K && rc == SYMGPU_OK) {
        if (cur->T == 0) {
            // every rotation is the identity on an operator without terms (np.all of an empty mask, base.py:1130-1133) and the
            // cleanup() that follows it returns 0 * I (base.py:631-632); the next step's cleanup() drops that term again
            symgpu_op_t z = nullptr;
            rc = zero_identity_op(in->Wq, &z);
            if (rc != SYMGPU_OK) break;
            replace(z);
            clean = 0;
            ++step;
            continue;
        }
        if (clean && cur->T <= ((i64)1 << 22) && ks_host[step] >= 0) {
            // a run of Clifford rotations of a clean operator: no step drops or merges anything, so the reference's per-step cleanup()
            // is the identity and the whole run goes to the chain entry point
            i64 e = step;
            while (e < K && ks_host[e] >= 0) ++e;
            symgpu_op_t res = nullptr;
            rc = symgpu_rotate_clifford_chain_dev(cur, q_rows_host + step * W, ks_host + step, e - step, &res);
            if (rc != SYMGPU_OK) break;
            replace(res);
            step = e;
            continue;
        }
        symgpu_op_t res = nullptr;
        int allc = 1;
        rc = symgpu_rotate_single_dev(cur, q_rows_host + step * W, cos_t[step], sin_t[step], ks_host[step], thr, &res, &allc);
        if (rc != SYMGPU_OK) break;
        if (!allc && res && res->T == 0) {
            // The rotation itself has left no term.  Clifford: the rows are vstack([cleaned product, commuting rows]) = none, and the
            // cleanup() after it gives 0 * I.  Non-Clifford: the result is `commute_self + anticom_part` (base.py:1159-1161), an
            // append + cleanup that yields 0 * I when BOTH parts hold no row (then the loop's cleanup() drops its term: no terms) and
            // an operator without terms when commuting rows cancelled each other (then the loop's cleanup() gives 0 * I).
            bool zero_identity = true;
            if (ks_host[step] < 0) {
                bool any = false;
                rc = any_row_commutes(cur, q_rows_host + step * W, &any);
                if (rc != SYMGPU_OK) { symgpu_op_free(res); break; }
                zero_identity = any;
            }
            if (acted) acted[step] = 1;
            if (zero_identity) {
                symgpu_op_free(res);
                res = nullptr;
                rc = zero_identity_op(in->Wq, &res);
                if (rc != SYMGPU_OK) break;
                clean = 0;
            } else {
                clean = 1;                                          // no terms, and the cleanup() of this step has been applied
            }
            replace(res);
            ++step;
            continue;
        }
        if (!allc) { if (acted) acted[step] = 1; replace(res); }
        else if (res) symgpu_op_free(res);
        ++step;
        if (!clean) {
            symgpu_op_t cleaned = nullptr;
            rc = symgpu_cleanup_dev(cur, thr, 1, &cleaned);          // may leave no term (0 * I, or X + (-X) after an even multiple of pi/2)
            if (rc != SYMGPU_OK) break;
            replace(cleaned);
            clean = 1;
        }
    }
    if (rc != SYMGPU_OK) { if (cur != in) symgpu_op_free(cur); return rc; }
    *n_done = step;
    *clean_out = clean;
    *out = cur != in ? cur : nullptr;                                // nullptr: the operator is unchanged, keep using `in`
    return SYMGPU_OK;
}

int symgpu_debug_rotation_trace(uint64_t *out, int max_workgroups, int *n_workgroups) {
    SG_TRY(require_ctx());
    SG_REQUIRE(out && n_workgroups && max_workgroups >= 0, "debug_rotation_trace");
    return rotate_resident_trace(out, max_workgroups, n_workgroups);
}

int symgpu_rotate_single(const uint64_t *rows, const double *coeff, int64_t N, int Wq, const uint64_t *q_row, double cos_t, double sin_t,
                         int clifford_k, double thr, uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out,
                         int *all_commute) {
    SG_TRY(require_ctx());
    SG_REQUIRE(N >= 0 && Wq >= 1 && q_row && all_commute && n_out, "rotate_single: arguments");
    SG_REQUIRE(N == 0 || (rows && coeff), "rotate_single: null input");
    symgpu_op_t in = nullptr, res = nullptr;
    SG_TRY(symgpu_op_upload(rows, coeff, N, Wq, &in));
    int rc = symgpu_rotate_single_dev(in, q_row, cos_t, sin_t, clifford_k, thr, &res, all_commute);
    symgpu_op_free(in);
    if (rc != SYMGPU_OK) return rc;
    if (*all_commute || !res) { *n_out = N; if (res) symgpu_op_free(res); return SYMGPU_OK; }
    *n_out = res->T;
    if (res->T > capacity) {
        set_error("rotate_single: capacity %lld < %lld rows", (long long)capacity, (long long)res->T);
        rc = SYMGPU_E_CAPACITY;
    } else {
        rc = symgpu_op_download(res, out_rows, out_coeff, capacity);
    }
    symgpu_op_free(res);
    return rc;
}

int symgpu_rotate_clifford_chain_dev(symgpu_op_t in, const uint64_t *q_rows_host, const int *ks_host, int64_t K, symgpu_op_t *out) {
    SG_ENTER(in);
    SG_REQUIRE(in && out && K >= 0 && (K == 0 || (q_rows_host && ks_host)), "rotate_clifford_chain_dev: null argument");
    SG_REQUIRE(in->coeff || in->T == 0, "rotate_clifford_chain_dev: operator has no coefficients");
    SG_REQUIRE(in->dup_free, "rotate_clifford_chain_dev: the operator must come from a cleanup (no duplicate rows, |c| > threshold)");
    SG_REQUIRE(in->T <= ((i64)1 << 22), "rotate_clifford_chain_dev: more than 2^22 rows (rotate one by one)");
    for (i64 r = 0; r < K; ++r) SG_REQUIRE(ks_host[r] >= 0 && ks_host[r] <= 3, "rotate_clifford_chain_dev: k must be 0..3 (see rotation_args)");
    hipStream_t st = ctx().stream;
    const i64 T = in->T;
    const int Wq = in->Wq, W = 2 * Wq;
    *out = nullptr;
    symgpu_op_t a = nullptr, b = nullptr;
    SG_TRY(symgpu_op_alloc(T > 0 ? T : 1, Wq, 1, &a));
    int rc = symgpu_op_alloc(T > 0 ? T : 1, Wq, 1, &b);
    if (rc != SYMGPU_OK) { symgpu_op_free(a); return rc; }
    Scratch qs, ks, which;
    hipError_t e = hipSuccess;
    int in_b = 0;
    if (T > 0) {
        e = hipMemcpyAsync(a->rows, in->rows, (size_t)T * W * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(a->coeff, in->coeff, (size_t)T * 16, hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess && T > 0 && K > 0) {
        rc = qs.alloc((size_t)K * W * 8);
        if (rc == SYMGPU_OK) rc = ks.alloc((size_t)K * 4);
        if (rc == SYMGPU_OK) rc = which.alloc(16);
        if (rc != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rc; }
        e = hipMemcpyAsync(qs.p, q_rows_host, (size_t)K * W * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(ks.p, ks_host, (size_t)K * 4, hipMemcpyHostToDevice, st);
        int G = 1;
        while (G < Wq && G < 64) G <<= 1;
        i64 local_t = CHAIN_LOCAL_T;
        if (const char *env = getenv("SYMGPU_CHAIN_LOCAL_T")) {      // tests: the single-workgroup kernel up to its own limit
            local_t = atoll(env);
            if (local_t > CHAIN_TMAX) local_t = CHAIN_TMAX;
        }
        const size_t lds_rows = (size_t)2 * T * W * 8;
        const bool lds_on = [] { const char *e3 = SG_TUNE("SYMGPU_CHAIN_LDS"); return !(e3 && e3[0] == '0'); }();
        const bool lds_attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_clifford_chain_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                                 128 * 1024) == hipSuccess);
        const bool reg_chain = clifford_chain_registers_applicable(T, Wq) && !getenv("SYMGPU_CHAIN_LOCAL_T");
        // the register chain; if its one-launch sort timed out at a barrier the rows in `a`/`b` are garbage: restore `a` from the
        // untouched input and run once more (the sort form is off by then, so the second run takes the multi-launch radix sort)
        auto run_reg_chain = [&]() -> int {
            int r = clifford_chain_registers(a, b, T, qs.as<u64>(), ks_host, K, &in_b);
            if (r != CHAIN_RETRY) return r;
            hipError_t e2 = hipMemcpyAsync(a->rows, in->rows, (size_t)T * W * 8, hipMemcpyDeviceToDevice, st);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(a->coeff, in->coeff, (size_t)T * 16, hipMemcpyDeviceToDevice, st);
            if (e2 != hipSuccess) return hip_fail(e2, "rotate_clifford_chain_dev: restore", __FILE__, __LINE__);
            r = clifford_chain_registers(a, b, T, qs.as<u64>(), ks_host, K, &in_b);
            if (r == CHAIN_RETRY) { set_error("rotate_clifford_chain: the one-launch sort timed out twice"); return SYMGPU_E_HIP; }
            return r;
        };
        if (e == hipSuccess && reg_chain) {
            // the whole run with the rows in registers and ONE sort of the accumulated partition bits per 40 rotations (rotate_chain.hip):
            // faster than every other form at every size (1 term: 0.5 us per rotation against 1.6 of the LDS-resident kernel, 128 terms:
            // 1.2 against 3.7, 10^5 terms: 5.3 against 22.9 of the two-launch form)
            rc = run_reg_chain();
            if (rc != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rc; }
        } else if (e == hipSuccess && lds_on && lds_attr && T <= local_t && T <= CHAIN_LDS_T && W <= 128 && lds_rows <= 128 * 1024) {
            // small operator, resident in LDS for the whole run
            hipLaunchKernelGGL(k_clifford_chain_lds, dim3(1), dim3(1024), lds_rows, st, a->rows, a->coeff, (int)T, Wq, G, qs.as<u64>(), ks.as<int>(), (int)K);
            e = hipGetLastError();
            in_b = 0;
        } else if (e == hipSuccess && T <= local_t) {
            // small operator: the whole run in one single-workgroup launch
            hipLaunchKernelGGL(k_clifford_chain, dim3(1), dim3(1024), 0, st, a->rows, a->coeff, b->rows, b->coeff, (int)T, Wq, G, qs.as<u64>(),
                               ks.as<int>(), (int)K, which.as<int>());
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(&in_b, which.p, 4, hipMemcpyDeviceToHost, st);
        } else if (e == hipSuccess && clifford_chain_registers_applicable(T, Wq)) {
            // (SYMGPU_CHAIN_LOCAL_T set and T above it: the tests' way to the register chain next to the single-workgroup kernels)
            rc = run_reg_chain();
            if (rc != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rc; }
        } else if (e == hipSuccess) {
            // large operator: the per-rotation kernels of the Clifford fast path, enqueued back to back — T is constant for a
            // clean operator (nothing is dropped: thr = -1), so no count has to come back to the host between the rotations
            Scratch anti, ph, selfc, prodc, cls, pself, pnew, cnt, blk;
            const int n_blk = (int)((T + 1023) / 1024);
            rc = anti.alloc((size_t)T * 4);
            if (rc == SYMGPU_OK) rc = ph.alloc((size_t)T);
            if (rc == SYMGPU_OK) rc = selfc.alloc((size_t)T * 16);
            if (rc == SYMGPU_OK) rc = prodc.alloc((size_t)T * 16);
            if (rc == SYMGPU_OK) rc = cls.alloc((size_t)T);
            if (rc == SYMGPU_OK) rc = pself.alloc((size_t)T * 4);
            if (rc == SYMGPU_OK) rc = pnew.alloc((size_t)T * 4);
            if (rc == SYMGPU_OK) rc = cnt.alloc(sizeof(RotCounts));
            if (rc == SYMGPU_OK) rc = blk.alloc((size_t)n_blk * 16);
            if (rc != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rc; }
            symgpu_op_t cur = a, nxt = b;
            const bool two_on = [] { const char *e2 = SG_TUNE("SYMGPU_CHAIN_TWO"); return !(e2 && e2[0] == '0'); }();
            if (two_on && Wq <= 64 && (Wq & (Wq - 1)) == 0 && T <= (SG_TUNE("SYMGPU_CHAIN_TWO_T") ? atoll(SG_TUNE("SYMGPU_CHAIN_TWO_T")) : CHAIN_TWO_T)) {
                // two launches per rotation (k_cchain_flags / k_cchain_move)
                Scratch af, nc, cnts;
                const int n_cnt = (int)((T + 1023) / 1024);
                rc = af.alloc((size_t)T + 1024);
                if (rc == SYMGPU_OK) rc = nc.alloc((size_t)T * 16);
                if (rc == SYMGPU_OK) rc = cnts.alloc(2 * 256 * sizeof(u32));
                if (rc != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rc; }
                e = hipMemsetAsync(cnts.p, 0, 2 * 256 * sizeof(u32), st);
                // rows per block: 256 / WQ (one pass: lowest latency, 6.7-7.1 us per rotation up to 1,000 rows) or 64 (the block's fixed
                // costs — Q, group counts, flag prefix — paid once per four passes: 10.4 / 13.4 / 20.8 / 26.7 us at 8,000 / 16,384 /
                // 65,536 / 131,072 rows against 12.9 / 20.7 / 40.2 / 59.3 us)
                const bool big = T > 4096;
                for (i64 r = 0; r < K && e == hipSuccess; ++r) {
                    const u64 *q = qs.as<u64>() + r * W;
                    u32 *c_now = cnts.as<u32>() + (r & 1) * 256, *c_next = cnts.as<u32>() + ((r + 1) & 1) * 256;
                    const u32x4 *src = reinterpret_cast<const u32x4 *>(cur->rows);
                    u32x4 *dstr = reinterpret_cast<u32x4 *>(nxt->rows);
#define LAUNCH_CC2(WQV, ROWSV) do { \
                        const i64 gb = (T + (ROWSV) - 1) / (ROWSV); \
                        hipLaunchKernelGGL((k_cchain_flags<WQV, ROWSV>), dim3((unsigned)gb), dim3(256), 0, st, src, cur->coeff, T, q, ks_host[r], af.as<uint8_t>(), nc.as<double>(), c_now); \
                        hipLaunchKernelGGL((k_cchain_move<WQV, ROWSV>), dim3((unsigned)gb), dim3(256), 0, st, src, T, q, ks_host[r], af.as<uint8_t>(), nc.as<double>(), c_now, n_cnt, \
                                           c_next, dstr, nxt->coeff); } while (0)
#define LAUNCH_CC(WQV) do { if (big && 256 / (WQV) < 64) LAUNCH_CC2(WQV, 64); else LAUNCH_CC2(WQV, (256 / (WQV))); } while (0)
                    switch (Wq) {
                        case 1: LAUNCH_CC(1); break;
                        case 2: LAUNCH_CC(2); break;
                        case 4: LAUNCH_CC(4); break;
                        case 8: LAUNCH_CC(8); break;
                        case 16: LAUNCH_CC(16); break;
                        case 32: LAUNCH_CC(32); break;
                        default: LAUNCH_CC(64); break;
                    }
#undef LAUNCH_CC2
#undef LAUNCH_CC
                    symgpu_op_t t2 = cur; cur = nxt; nxt = t2;
                }
                if (e == hipSuccess) e = hipGetLastError();
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                in_b = (cur == b) ? 1 : 0;
            } else {
            for (i64 r = 0; r < K; ++r) {
                u64 *q = qs.as<u64>() + r * W;
                {   // flags + phase exponents (chunk-per-lane kernel where the row length allows it)
                    cur->T = T;
                    const int rca = analyze_rows(cur, q, anti.as<u32>(), ph.as<uint8_t>(), nullptr);
                    if (rca != SYMGPU_OK) { symgpu_op_free(a); symgpu_op_free(b); return rca; }
                }
                hipLaunchKernelGGL(k_rotc_classify, dim3(n_blk), dim3(1024), 0, st, anti.as<u32>(), ph.as<uint8_t>(), cur->coeff, T, ks_host[r], -1.0,
                                   cls.as<uint8_t>(), selfc.as<double>(), prodc.as<double>(), blk.as<u32>());
                hipLaunchKernelGGL(k_rotf_scan3, dim3(n_blk), dim3(1024), 0, st, cls.as<uint8_t>(), T, blk.as<u32>(), n_blk, pself.as<u32>(), pnew.as<u32>(),
                                   cnt.as<RotCounts>(), (const u32 *)nullptr, 0u);
                hipLaunchKernelGGL(k_rotf_write, dim3(grid_for((T * Wq + 3) / 4) + (unsigned)((T + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(cur->rows),
                                   reinterpret_cast<const u32x4 *>(q), T, Wq, cls.as<uint8_t>(), pself.as<u32>(), pnew.as<u32>(), cnt.as<RotCounts>(),
                                   selfc.as<double>(), prodc.as<double>(), reinterpret_cast<u32x4 *>(nxt->rows), nxt->coeff, 1, (const u64 *)nullptr, (u64)0,
                                   (u64 *)nullptr);
                symgpu_op_t t2 = cur; cur = nxt; nxt = t2;
            }
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(st);       // the scratch buffers go back to the allocator on return
            in_b = (cur == b) ? 1 : 0;
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { symgpu_op_free(a); symgpu_op_free(b); return hip_fail(e, "rotate_clifford_chain_dev", __FILE__, __LINE__); }
    symgpu_op_t res = in_b ? b : a;
    symgpu_op_free(in_b ? a : b);
    res->T = T;
    res->dup_free = 1;                                              // a permutation of distinct rows XORed with Q on a Q-anticommuting subset
    *out = res;
    return SYMGPU_OK;
}

}  // extern "C"
