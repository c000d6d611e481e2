// cleanup.hip — duplicate-term cleanup (reference: symplectic_cleanup, symmer/operators/utils.py:230-279;
// PauliwordOp.cleanup base.py:617-638) and the fused product+cleanup (base.py:764-794).
//
// The reference keys a hash map with the full row.  Here rows are grouped by sorting a 64-bit GF(2)-LINEAR
// row hash  h(r) = XOR over set bits of a random 64-bit vector per bit position, evaluated as 8 byte-table
// lookups per word (LDS), a per-word rotation and a per-64-word-block xorshift step.  Linearity gives
// h(a ^ b) = h(a) ^ h(b): the key of product row (i, o) is hI[i] ^ hO[o], so the fused path never
// materialises the N*M product rows — only the surviving unique rows are written.
// Exactness does not rest on the hash: after the stable sort every adjacent equal-key pair is compared
// word by word; any mismatch reseeds the tables and retries (SYMGPU_E_COLLISION if it survives 4 seeds).
//
// Pipeline: hash -> stable LSD radix sort (key, input index) -> head flags + verify -> scan (segment ids)
//           -> per-segment SEQUENTIAL coefficient sum in input order (== np.add.at, utils.py:273-274)
//           -> threshold |c| > thr (strict, utils.py:275-278) -> first-occurrence order via mark+scan over
//           input positions (qiskit `unordered_unique` order, utils.py:271) -> gather surviving rows.
#include "common.h"
#include <vector>

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static u64 host_splitmix64(u64 &s) {
    u64 z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static std::vector<u64> g_host_tab;   // host copy of the device tables (to hash single rows, e.g. a rotation's Q)

int ensure_hash_tables(u64 seed) {
    Context &c = ctx();
    if (c.hash_tab && c.hash_seed == seed) return SYMGPU_OK;
    if (!c.hash_tab) HIP_TRY(hipMalloc((void **)&c.hash_tab, 8 * 256 * 2 * sizeof(u64)));
    std::vector<u64> tab(8 * 256 * 2);
    u64 s = seed * 0x2545f4914f6cdd1dULL + 0x1234567ULL;
    u64 basis[2][64];
    for (int h = 0; h < 2; ++h)
        for (int b = 0; b < 64; ++b) basis[h][b] = host_splitmix64(s);
    for (int k = 0; k < 8; ++k)
        for (int v = 0; v < 256; ++v)
            for (int h = 0; h < 2; ++h) {
                u64 x = 0;
                for (int b = 0; b < 8; ++b)
                    if ((v >> b) & 1) x ^= basis[h][8 * k + b];
                tab[((size_t)k * 256 + v) * 2 + h] = x;
            }
    HIP_TRY(hipMemcpyAsync(c.hash_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));   // tab is a host temporary
    c.hash_seed = seed;
    g_host_tab = tab;
    return SYMGPU_OK;
}

// host evaluation of the same linear hash h1 as k_hash_rows (per-lane Horner over 64-word blocks, XOR over lanes)
u64 host_row_hash(const u64 *row, int W) {
    const int n_blk = (W + 63) / 64;
    u64 h = 0;
    for (int g = 0; g < 64; ++g) {
        u64 hg = 0;
        for (int b = 0; b < n_blk; ++b) {
            const int w = b * 64 + g;
            u64 a1 = 0;
            if (w < W) {
                const u64 x = row[w];
                for (int k = 0; k < 8; ++k) a1 ^= g_host_tab[((size_t)k * 256 + ((x >> (8 * k)) & 255)) * 2];
            }
            hg ^= hg << 13; hg ^= hg >> 7; hg ^= hg << 17;
            const int r = g & 63;
            hg ^= r ? ((a1 << r) | (a1 >> (64 - r))) : a1;
        }
        h ^= hg;
    }
    return h;
}

__device__ __forceinline__ u64 rotl64(u64 x, int r) { r &= 63; return r ? ((x << r) | (x >> (64 - r))) : x; }
__device__ __forceinline__ u64 xorshift_step(u64 h) { h ^= h << 13; h ^= h >> 7; h ^= h << 17; return h; }

// Row hash on row-major rows.  G (power of two, <= 64) lanes cooperate on one row; lane g handles words
// g, g+64, g+128, ... (only G == 64 has more than one).  h1 -> out1[t], h2 -> out2[t] (out2 may be null).
__global__ __launch_bounds__(256) void k_hash_rows(const u64 *__restrict__ rows, i64 T, int W, int G, const u64 *__restrict__ tab_g,
                                                    u64 *__restrict__ out1, u64 *__restrict__ out2) {
    __shared__ u64 tab[8 * 256 * 2];
    for (int k = threadIdx.x; k < 8 * 256 * 2; k += 256) tab[k] = tab_g[k];
    __syncthreads();
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    const int n_blk = (W + 63) / 64;
    for (i64 t0 = (i64)blockIdx.x * rows_per_block; t0 < T; t0 += (i64)gridDim.x * rows_per_block) {
        const i64 t = t0 + rsub;
        u64 h1 = 0, h2 = 0;
        if (t < T) {
            for (int b = 0; b < n_blk; ++b) {
                const int w = b * 64 + g;
                u64 a1 = 0, a2 = 0;
                if (w < W && g < 64) {
                    const u64 x = rows[t * W + w];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int v = (int)((x >> (8 * k)) & 255);
                        a1 ^= tab[(k * 256 + v) * 2];
                        a2 ^= tab[(k * 256 + v) * 2 + 1];
                    }
                }
                h1 = xorshift_step(h1) ^ rotl64(a1, g);
                h2 = xorshift_step(h2) ^ rotl64(a2, g * 29 + 7);
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) {
            h1 ^= __shfl_xor(h1, off);
            h2 ^= __shfl_xor(h2, off);
        }
        if (g == 0 && t < T) {
            out1[t] = h1;
            if (out2) out2[t] = h2;
        }
    }
}

__global__ void k_iota_keys_plain(u32 *__restrict__ idx, i64 T) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) idx[t] = (u32)t;
}

__global__ void k_pair_keys(const u64 *__restrict__ hI, i64 Ni, const u64 *__restrict__ hO, i64 T, u64 *__restrict__ keys, u32 *__restrict__ idx) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        const i64 o = t / Ni, i = t - o * Ni;
        keys[t] = hI[i] ^ hO[o];
        idx[t] = (u32)t;
    }
}

// head flags + exact verification of equal-key neighbours + coefficient gather into sorted order.
// PAIR: row(t) = inner[t % Ni] ^ outer[t / Ni].  One wavefront owns 64 consecutive sorted positions; the positions whose
// key equals their predecessor's (ballot) are verified COOPERATIVELY: G = pow2 >= W lanes (<= 64) read the words of the
// two rows with coalesced loads, 64/G comparisons in flight per step.  A mismatch (two different rows with one 64-bit
// hash) only raises the collision flag: the caller reseeds the hash and redoes the pass, so exactness never rests on the
// hash.  cg[s] = coeff[idx[s]] turns the segment sums into sequential reads.
template <bool PAIR>
__global__ __launch_bounds__(256) void k_heads(const u64 *__restrict__ keys, const u32 *__restrict__ idx, i64 T, const u64 *__restrict__ rows, int W,
                                                const u64 *__restrict__ inner, i64 Ni, const u64 *__restrict__ outer, int G,
                                                const double *__restrict__ coeff, double *__restrict__ cg,
                                                u32 *__restrict__ heads, u32 *__restrict__ collision) {
    const int lane = threadIdx.x & 63;
    const int per = 64 / G, gi = lane / G, gl = lane % G;
    const i64 n_chunks = (T + 63) / 64;
    for (i64 chunk = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks; chunk += (i64)gridDim.x * 4) {
        const i64 s = chunk * 64 + lane;
        const bool valid = s < T;
        const u64 k = valid ? keys[s] : 0ULL;
        const u64 kp = (valid && s > 0) ? keys[s - 1] : ~k;
        const u32 t1 = valid ? idx[s] : 0u;
        const u32 t0 = (valid && s > 0) ? idx[s - 1] : 0u;
        const bool eq = valid && s > 0 && k == kp;
        if (valid) {
            heads[s] = eq ? 0u : 1u;
            reinterpret_cast<double2 *>(cg)[s] = reinterpret_cast<const double2 *>(coeff)[t1];
        }
        u64 m = __ballot(eq);
        bool mism = false;
        while (m) {                                                  // wave-uniform
            u64 mm = m;
            for (int q = 0; q < gi; ++q) mm &= mm - 1;               // this group's candidate: the gi-th lowest set bit
            const bool act = mm != 0;
            const int p = act ? __builtin_ctzll(mm) : 0;
            const i64 a1 = __shfl(t1, p), a0 = __shfl(t0, p);
            if (act) {
                if (PAIR) {
                    const i64 o1 = a1 / Ni, i1 = a1 - o1 * Ni, o0 = a0 / Ni, i0 = a0 - o0 * Ni;
                    const u64 *r1 = inner + i1 * W, *q1 = outer + o1 * W, *r0 = inner + i0 * W, *q0 = outer + o0 * W;
                    for (int w = gl; w < W; w += G) mism |= ((r1[w] ^ q1[w]) != (r0[w] ^ q0[w]));
                } else {
                    const u64 *r1 = rows + a1 * W, *r0 = rows + a0 * W;
                    for (int w = gl; w < W; w += G) mism |= (r1[w] != r0[w]);
                }
            }
            for (int q = 0; q < per && m; ++q) m &= m - 1;           // retire the `per` candidates just handled
        }
        if (__ballot(mism) && lane == 0) atomicOr(collision, 1u);
    }
}

// Truncated sort fix-up.  The radix sort only orders the top `nb` key bits (random hash bits: ~log2(T)+5..12 of them already
// separate almost all distinct keys).  Inside a run of equal prefixes the elements are still in input order; if such a
// run holds more than one distinct key it is re-ordered here by (full key, input order) with a stable insertion sort.
// Runs longer than FIX_MAX that are not uniform raise `fallback`: the caller then redoes a full 64-bit sort.
constexpr int FIX_MAX = 48;
// phase 1 (read-only): a position whose key differs from its predecessor's INSIDE a prefix run marks the run's start
__global__ void k_fixup_mark(const u64 *__restrict__ keys, i64 T, int shift, u32 *__restrict__ need) {
    for (i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x; s < T; s += (i64)gridDim.x * blockDim.x) {
        if (s == 0) continue;
        const u64 k = keys[s], kp = keys[s - 1];
        if ((k >> shift) != (kp >> shift) || k == kp) continue;
        i64 b = s - 1;
        while (b > 0 && (keys[b - 1] >> shift) == (kp >> shift)) --b;
        need[b] = 1u;
    }
}
// phase 2: the marked run starts (one thread per mixed run) sort their run
__global__ void k_fixup_sort(u64 *__restrict__ keys, u32 *__restrict__ idx, i64 T, int shift, const u32 *__restrict__ need, u32 *__restrict__ fallback) {
    for (i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x; b < T; b += (i64)gridDim.x * blockDim.x) {
        if (!need[b]) continue;
        const u64 pfx = keys[b] >> shift;
        i64 e = b + 1;
        while (e < T && (keys[e] >> shift) == pfx) ++e;
        if (e - b > FIX_MAX) { atomicOr(fallback, 1u); continue; }
        for (i64 a = b + 1; a < e; ++a) {                                     // stable insertion sort by full key
            const u64 ka = keys[a];
            const u32 ia = idx[a];
            i64 c = a - 1;
            while (c >= b && keys[c] > ka) { keys[c + 1] = keys[c]; idx[c + 1] = idx[c]; --c; }
            keys[c + 1] = ka;
            idx[c + 1] = ia;
        }
    }
}

// Segment sums without segment ids: the thread at a head position walks its segment (the following non-head positions),
// summing SEQUENTIALLY in ascending input order (the sort is stable) — exactly np.add.at's order (utils.py:273-274).
// The sum replaces cg[s] at the head; heads[s] becomes 2 if the term survives the strict |c| > thr test (1 otherwise) and
// the first-occurrence index of a surviving term is marked for the output-order scan.
__global__ void k_segsum_heads(u32 *__restrict__ heads, const u32 *__restrict__ idx, i64 T, double *__restrict__ cg, double thr, int use_thr,
                               u32 *__restrict__ mark) {
    for (i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x; s < T; s += (i64)gridDim.x * blockDim.x) {
        if (heads[s] == 0u) continue;
        double2 c = reinterpret_cast<const double2 *>(cg)[s];
        double re = __dadd_rn(0.0, c.x), im = __dadd_rn(0.0, c.y);
        for (i64 e = s + 1; e < T && heads[e] == 0u; ++e) {
            c = reinterpret_cast<const double2 *>(cg)[e];
            re = __dadd_rn(re, c.x);
            im = __dadd_rn(im, c.y);
        }
        const bool keep = use_thr ? (hypot(re, im) > thr) : true;
        if (keep) {
            double2 o; o.x = re; o.y = im;
            reinterpret_cast<double2 *>(cg)[s] = o;
            heads[s] = 2u;
            mark[idx[s]] = 1u;
        }
    }
}

// out position of a kept segment = exclusive scan of mark at its first index (= idx at the head: the sort is stable)
__global__ void k_emit_heads(const u32 *__restrict__ heads, const u32 *__restrict__ idx, i64 T, const double *__restrict__ cg,
                             const u32 *__restrict__ outpos, double *__restrict__ out_coeff, u32 *__restrict__ out_src) {
    for (i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x; s < T; s += (i64)gridDim.x * blockDim.x) {
        if (heads[s] != 2u) continue;
        const u32 first = idx[s];
        const u32 p = outpos[first];
        reinterpret_cast<double2 *>(out_coeff)[p] = reinterpret_cast<const double2 *>(cg)[s];
        out_src[p] = first;
    }
}

// gather surviving rows as 16-byte chunks: out[p][c] = row(out_src[p])[c]
template <bool PAIR>
__global__ void k_gather_rows(const u32 *__restrict__ out_src, i64 n_out, int Wq, const u32x4 *__restrict__ rows,
                              const u32x4 *__restrict__ inner, i64 Ni, const u32x4 *__restrict__ outer, u32x4 *__restrict__ out) {
    const i64 total = n_out * Wq;
    for (i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (i64)gridDim.x * blockDim.x) {
        const i64 p = k / Wq;
        const int c = (int)(k - p * Wq);
        const i64 t = out_src[p];
        u32x4 v;
        if (PAIR) {
            const i64 o = t / Ni, i = t - o * Ni;
            v = inner[i * Wq + c] ^ outer[o * Wq + c];
        } else {
            v = rows[t * Wq + c];
        }
        out[k] = v;
    }
}

static int grid_for(i64 n, int block = 256, int cap = 8192) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static int pow2_group(int W) {
    int g = 1;
    while (g < W && g < 64) g <<= 1;
    return g;
}

int hash_rows(const u64 *rows, i64 T, int W, u64 *out1) {
    if (T == 0) return SYMGPU_OK;
    const int G = pow2_group(W);
    const int rpb = 256 / G;
    i64 g = (T + rpb - 1) / rpb;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_hash_rows, dim3((unsigned)g), dim3(256), 0, ctx().stream, rows, T, W, G, ctx().hash_tab, out1, (u64 *)nullptr);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

int cleanup_finish(u32 *heads, const u32 *is, i64 T, double *cg, double thr, int use_thr, u32 *mark, bool pair, const u64 *rows, int W,
                   const u64 *inner, i64 Ni, const u64 *outer, symgpu_op_t *out, int Wq_out) {
    hipStream_t st = ctx().stream;
    hipLaunchKernelGGL(k_segsum_heads, dim3(grid_for(T)), dim3(256), 0, st, heads, is, T, cg, thr, use_thr, mark);
    KERNEL_CHECK();
    Scratch total;
    SG_TRY(total.alloc(16));
    SG_TRY(exclusive_scan_u32(mark, mark, T, total.as<u32>()));
    u32 n_out32 = 0;
    HIP_TRY(hipMemcpyAsync(&n_out32, total.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const i64 n_out = n_out32;
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(n_out > 0 ? n_out : 1, Wq_out, 1, &res));
    res->T = n_out;
    if (n_out > 0) {
        Scratch src;
        int rc = src.alloc((size_t)n_out * 4);
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        hipLaunchKernelGGL(k_emit_heads, dim3(grid_for(T)), dim3(256), 0, st, heads, is, T, cg, mark, res->coeff, src.as<u32>());
        const int Wq = W / 2;
        if (pair)
            hipLaunchKernelGGL(k_gather_rows<true>, dim3(grid_for(n_out * Wq)), dim3(256), 0, st, src.as<u32>(), n_out, Wq,
                               (const u32x4 *)nullptr, reinterpret_cast<const u32x4 *>(inner), Ni, reinterpret_cast<const u32x4 *>(outer),
                               reinterpret_cast<u32x4 *>(res->rows));
        else
            hipLaunchKernelGGL(k_gather_rows<false>, dim3(grid_for(n_out * Wq)), dim3(256), 0, st, src.as<u32>(), n_out, Wq,
                               reinterpret_cast<const u32x4 *>(rows), (const u32x4 *)nullptr, (i64)1, (const u32x4 *)nullptr,
                               reinterpret_cast<u32x4 *>(res->rows));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // src is freed on return; keep ordering simple
        if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "cleanup emit/gather", __FILE__, __LINE__); }
    }
    *out = res;
    return SYMGPU_OK;
}

// plain mode: rows/coeff of T terms.  pair mode (inner != null): T = Ni*No, coeff holds the pair coefficients
// in index order t = o*Ni + i.  *out is a fresh operator with the cleaned result.
int cleanup_core(const u64 *rows, const double *coeff, i64 T, int W, const u64 *inner, i64 Ni, const u64 *outer, i64 No,
                 double thr, int use_thr, symgpu_op_t *out, int Wq_out) {
    hipStream_t st = ctx().stream;
    const bool pair = inner != nullptr;
    if (T >= ((i64)1 << 32) - 1) {
        set_error("cleanup: %lld terms exceed the 2^32-2 limit of the 32-bit index sort", (long long)T);
        return SYMGPU_E_INVALID;
    }
    symgpu_op_t res = nullptr;
    if (T == 0) {
        SG_TRY(symgpu_op_alloc(1, Wq_out, 1, &res));
        res->T = 0;
        *out = res;
        return SYMGPU_OK;
    }
    Scratch keys, keys2, idx, idx2, heads, collision, cg;
    SG_TRY(keys.alloc((size_t)T * 8));
    SG_TRY(keys2.alloc((size_t)T * 8));
    SG_TRY(idx.alloc((size_t)T * 4));
    SG_TRY(idx2.alloc((size_t)T * 4));
    SG_TRY(heads.alloc((size_t)T * 4));
    SG_TRY(collision.alloc(16));
    SG_TRY(cg.alloc((size_t)T * 16));
    u64 *ks = nullptr;
    u32 *is = nullptr;
    u64 seed = ctx().hash_tab ? ctx().hash_seed : 1;
    bool ok = false;
    // number of (top) key bits the radix sort orders; the rest is handled by k_fixup_mark / k_fixup_sort
    int nb = 64;
    {
        int lg = 0;
        while (((i64)1 << lg) < T) ++lg;
        const int want = (lg + 5 + 7) / 8 * 8;   // ~1-3 % of the keys then share a prefix with another key: cheap local fix-up
        if (want < 64) nb = want;
    }
    for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
        SG_TRY(ensure_hash_tables(seed));
        if (pair) {
            Scratch hI, hO;
            SG_TRY(hI.alloc((size_t)Ni * 8));
            SG_TRY(hO.alloc((size_t)No * 8));
            SG_TRY(hash_rows(inner, Ni, W, hI.as<u64>()));
            SG_TRY(hash_rows(outer, No, W, hO.as<u64>()));
            hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(T)), dim3(256), 0, st, hI.as<u64>(), Ni, hO.as<u64>(), T, keys.as<u64>(), idx.as<u32>());
            KERNEL_CHECK();
        } else {
            SG_TRY(hash_rows(rows, T, W, keys.as<u64>()));
            hipLaunchKernelGGL(k_iota_keys_plain, dim3(grid_for(T)), dim3(256), 0, st, idx.as<u32>(), T);
            KERNEL_CHECK();
        }
        bool in_tmp = false;
        SG_TRY(radix_sort_pairs_u64_u32(keys.as<u64>(), idx.as<u32>(), keys2.as<u64>(), idx2.as<u32>(), T, 64 - nb, 64, &in_tmp));
        ks = in_tmp ? keys2.as<u64>() : keys.as<u64>();
        is = in_tmp ? idx2.as<u32>() : idx.as<u32>();
        HIP_TRY(hipMemsetAsync(collision.p, 0, 16, st));
        if (nb < 64) {
            // `heads` doubles as the run-start marker array here (it is overwritten by k_heads afterwards)
            HIP_TRY(hipMemsetAsync(heads.p, 0, (size_t)T * 4, st));
            hipLaunchKernelGGL(k_fixup_mark, dim3(grid_for(T)), dim3(256), 0, st, ks, T, 64 - nb, heads.as<u32>());
            hipLaunchKernelGGL(k_fixup_sort, dim3(grid_for(T)), dim3(256), 0, st, ks, is, T, 64 - nb, heads.as<u32>(), collision.as<u32>() + 1);
            KERNEL_CHECK();
        }
        {
            int G = 8;
            while (G < W && G < 64) G <<= 1;
            i64 gh = ((T + 63) / 64 + 3) / 4;
            if (gh > 16384) gh = 16384;
            if (pair)
                hipLaunchKernelGGL(k_heads<true>, dim3((unsigned)gh), dim3(256), 0, st, ks, is, T, (const u64 *)nullptr, W, inner, Ni, outer, G,
                                   coeff, cg.as<double>(), heads.as<u32>(), collision.as<u32>());
            else
                hipLaunchKernelGGL(k_heads<false>, dim3((unsigned)gh), dim3(256), 0, st, ks, is, T, rows, W, (const u64 *)nullptr, (i64)1,
                                   (const u64 *)nullptr, G, coeff, cg.as<double>(), heads.as<u32>(), collision.as<u32>());
        }
        KERNEL_CHECK();
        u32 hflags[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(hflags, collision.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (hflags[1]) { nb = 64; continue; }       // a long mixed prefix run: redo with a full 64-bit sort, same seed
        ok = (hflags[0] == 0);
        if (!ok) ++seed;                            // genuine 64-bit hash collision: reseed and retry
    }
    if (!ok) {
        set_error("cleanup: 64-bit row-hash collision survived 4 reseeds");
        return SYMGPU_E_COLLISION;
    }
    Scratch mark;
    SG_TRY(mark.alloc((size_t)T * 4));
    HIP_TRY(hipMemsetAsync(mark.p, 0, (size_t)T * 4, st));
    return cleanup_finish(heads.as<u32>(), is, T, cg.as<double>(), thr, use_thr, mark.as<u32>(), pair, rows, W, inner, Ni, outer, out, Wq_out);
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_cleanup_dev(symgpu_op_t in, double thr, int use_thr, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(in && out, "cleanup_dev: null handle");
    SG_REQUIRE(in->coeff || in->T == 0, "cleanup_dev: operator has no coefficients");
    return cleanup_core(in->rows, in->coeff, in->T, 2 * in->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, in->Wq);
}

int symgpu_mul_cleanup_dev(symgpu_op_t inner, symgpu_op_t outer, int inner_is_left, double thr, int use_thr, symgpu_op_t *out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(inner && outer && out, "mul_cleanup_dev: null handle");
    SG_REQUIRE(inner->Wq == outer->Wq, "mul_cleanup_dev: operands must share Wq");
    const i64 Ni = inner->T, No = outer->T;
    const i64 T = Ni * No;
    if (T == 0) return cleanup_core(nullptr, nullptr, 0, 2 * inner->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, inner->Wq);
    SG_REQUIRE(inner->coeff && outer->coeff, "mul_cleanup_dev: operands have no coefficients");
    SG_REQUIRE(No == 0 || Ni < ((i64)1 << 32) / No, "mul_cleanup_dev: Ni*No must stay below 2^32 (tile the outer operand)");
    Scratch coeff;
    SG_TRY(coeff.alloc((size_t)T * 16));
    SG_TRY(mul_coeff_dev(inner->rows, inner->coeff, Ni, outer->rows, outer->coeff, 0, No, inner->Wq, inner_is_left, coeff.as<double>()));
    return cleanup_core(nullptr, coeff.as<double>(), T, 2 * inner->Wq, inner->rows, Ni, outer->rows, No, thr, use_thr, out, inner->Wq);
}

static int finish_to_host(symgpu_op_t res, uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out) {
    if (n_out) *n_out = res->T;
    int rc = SYMGPU_OK;
    if (res->T > capacity) {
        set_error("output capacity %lld < %lld rows", (long long)capacity, (long long)res->T);
        rc = SYMGPU_E_CAPACITY;
    } else if (res->T > 0) {
        rc = symgpu_op_download(res, out_rows, out_coeff, capacity);
    }
    symgpu_op_free(res);
    return rc;
}

int symgpu_cleanup(const uint64_t *rows, const double *coeff, int64_t T, int W, double thr, int use_thr, uint64_t *out_rows,
                   double *out_coeff, int64_t capacity, int64_t *n_out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(T >= 0 && W >= 2 && (W % 2) == 0 && capacity >= 0, "cleanup: sizes (W must be 2*Wq)");
    SG_REQUIRE(T == 0 || (rows && coeff), "cleanup: null input");
    symgpu_op_t in = nullptr, res = nullptr;
    SG_TRY(symgpu_op_upload(rows, coeff, T, W / 2, &in));
    if (T == 0 && !in->coeff) { /* empty upload has no coeff buffer; cleanup_core handles T == 0 first */ }
    int rc = cleanup_core(in->rows, in->coeff, T, W, nullptr, 0, nullptr, 0, thr, use_thr, &res, W / 2);
    symgpu_op_free(in);
    if (rc != SYMGPU_OK) return rc;
    return finish_to_host(res, out_rows, out_coeff, capacity, n_out);
}

int symgpu_mul_cleanup(const uint64_t *inner, const double *ci, int64_t Ni, const uint64_t *outer, const double *co, int64_t No,
                       int Wq, int inner_is_left, double thr, int use_thr, uint64_t *out_rows, double *out_coeff, int64_t capacity,
                       int64_t *n_out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(Ni >= 0 && No >= 0 && Wq >= 1 && capacity >= 0, "mul_cleanup: sizes");
    if (Ni == 0 || No == 0) { if (n_out) *n_out = 0; return SYMGPU_OK; }
    SG_REQUIRE(inner && outer && ci && co, "mul_cleanup: null input");
    symgpu_op_t a = nullptr, b = nullptr, res = nullptr;
    int rc = symgpu_op_upload(inner, ci, Ni, Wq, &a);
    if (rc == SYMGPU_OK) rc = symgpu_op_upload(outer, co, No, Wq, &b);
    if (rc == SYMGPU_OK) rc = symgpu_mul_cleanup_dev(a, b, inner_is_left, thr, use_thr, &res);
    symgpu_op_free(a); symgpu_op_free(b);
    if (rc != SYMGPU_OK) return rc;
    return finish_to_host(res, out_rows, out_coeff, capacity, n_out);
}

}  // extern "C"
